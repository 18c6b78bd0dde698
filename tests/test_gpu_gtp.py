"""GPU: the GTP front-end (elf_amd/gtp.py over elfsp_play / elfsp_restart / the search loop) -- the reference console's command
set and replies (scripts/elfgames/go/console_lib.py:207-372), board state checked against the CPU oracle."""
import numpy as np
import pytest

from pyoracle import Port

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def elf(built):
    import elf_amd
    return elf_amd


def make_actor(n, seed=0):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)

    def actor(batch):
        b = batch["s"].shape[0]
        pi = torch.softmax(2.0 * torch.randn((b, n * n + 1), device="cuda", generator=g), dim=1)
        v = torch.round(torch.tanh(torch.randn((b,), device="cuda", generator=g)) * 64) / 64
        return dict(pi=pi, V=v)
    return actor


def test_gtp_session(elf):
    from elf_amd.gtp import GtpEngine, move2xy, xy2move
    n = 9
    eng = GtpEngine(make_actor(n), board_size=n, mcts_rollout_per_thread=64, nodes_per_game=2048, keep_records=4)
    port = Port(n)
    st = port.new()
    S = n + 2

    def coord(mv):
        x, y = move2xy(mv)
        return 0 if x < 0 else (y + 1) * S + (x + 1)

    assert eng.command("protocol_version") == "= 2\n\n"
    assert eng.command("name") == "= DF2\n\n"
    assert eng.command("boardsize 9") == "= \n\n"
    assert eng.command("boardsize 19").startswith("? We only support 9x9")
    assert eng.command("komi 7.5") == "= \n\n" and eng.command("komi 6.5").startswith("? We only support")
    assert "genmove" in eng.command("list_commands") and eng.command("foo").startswith("?")
    assert eng.command("clear_board") == "= \n\n"
    assert eng.command("play b D4") == "= \n\n"
    assert port.forward(st, coord("D4")) == 1
    assert eng.command("play b E5").startswith("? Specified next player b is not the same as the next player W")
    assert eng.command("play w D4") == "? illegal move\n\n"          # occupied: the game is untouched
    assert eng.command("play w Z9") == "? illegal move\n\n"
    r = eng.command("genmove w")
    assert r.startswith("= ")
    mv = r[2:].strip()
    assert port.forward(st, coord(mv)) == 1, mv
    assert eng.command("genmove w").startswith("? Specified next player")
    assert eng.command("play b pass") == "= \n\n"
    assert port.forward(st, 0) == 1
    for _ in range(3):                       # engine and human alternate a few more moves, J-column letters included
        r = eng.command("genmove " + eng.next_player().lower())
        assert r.startswith("= ")
        assert port.forward(st, coord(r[2:].strip())) == 1
        legal = port.legal_mask(st)
        a = int(np.nonzero(legal[: n * n])[0][-1])          # the last legal point (high x: exercises the skipped letter I)
        mvs = xy2move(a // n, a % n)
        assert eng.command("play %s %s" % (eng.next_player().lower(), mvs)) == "= \n\n"
        assert port.forward(st, coord(mvs)) == 1
    info = eng.boards.info_host(n=1)
    assert int(info["hash"][0]) == port.hash(st) and int(info["ply"][0]) == int(port.info(st)[0])
    sb = eng.command("showboard")
    assert sb.startswith("= \n") and " X" in sb and " O" in sb and "Next:" in sb
    score = port.evaluate(st, 7.5)
    want = ("B+%.1f" % score) if score > 0 else ("W+%.1f" % -score)
    assert eng.command("final_score") == "= %s\n\n" % want
    assert eng.command("clear_board") == "= \n\n"
    assert int(eng.boards.info_host(n=1)["ply"][0]) == 1
    assert eng.command("final_score") == "= %s\n\n" % want          # getLastScore of the game just cleared
    recs = eng.sp.pop_records()
    assert len(recs) == 1                                            # the first clear_board found a game that had not started (:306-311)
    import json
    j = json.loads(recs[0])
    assert j["result"]["num_move"] == int(port.info(st)[0]) - 1 and abs(j["result"]["reward"] - score) < 1e-6
    assert eng.command("quit") == "= \n\n" and eng.exit
    eng.close()


def test_genmove_reports_a_resignation(elf):
    """ResignCheck (game_utils.h:14-54) through the console: a value head that sees Black losing makes the engine resign for
    Black once ply >= 50 (ply 51, Black to move); genmove must answer 'resign', not a move inferred from the restarted board."""
    import torch
    from elf_amd.gtp import GtpEngine
    n = 9
    g = torch.Generator(device="cuda").manual_seed(3)

    def actor(batch):
        s = batch["s"]
        b = s.shape[0]
        pi = torch.softmax(2.0 * torch.randn((b, n * n + 1), device="cuda", generator=g), dim=1)
        v = torch.full((b,), -0.96875, device="cuda")        # values are Black-positive: Black is always seen losing
        return dict(pi=pi, V=v)

    eng = GtpEngine(actor, board_size=n, mcts_rollout_per_thread=32, nodes_per_game=2048, resign_thres=0.1, ply_pass_enabled=200)
    replies = []
    for _ in range(80):
        r = eng.command("genmove " + eng.next_player().lower())
        assert r.startswith("= ")
        replies.append(r[2:].strip())
        if replies[-1] == "resign":
            break
    assert replies[-1] == "resign" and len(replies) == 51 and "resign" not in replies[:-1]
    assert int(eng.boards.info_host(n=1)["ply"][0]) == 1          # finish_game(FR_RESIGN) restarted the board
    eng.close()


def test_gtp_genmove_latency_19x19(elf, record_property):
    """Single-game latency through the GTP front-end (what README.rst:147 is about: one game, bs 16): `genmove` on 19x19 with 1024 and
    4096 rollouts per move and a net that costs nothing (random replies on the GPU), i.e. the search kernels + the per-step Python
    callback of this front-end.  A step is 7 kernel launches for ONE game: launch-bound, ~0.3-0.5 ms; the bound below is loose (a
    regression guard, not a benchmark: `bench.py`'s sub-result `single_game` times the same loop with the real net)."""
    import time
    import torch
    from elf_amd.gtp import GtpEngine
    n = 19
    out = {}
    for rollouts in (1024, 4096):
        eng = GtpEngine(make_actor(n), board_size=n, mcts_rollout_per_thread=rollouts, nodes_per_game=4 * rollouts + 1024)
        assert eng.command("genmove b").startswith("= ")        # first move: also pays the one-off set-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = eng.command("genmove w")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert r.startswith("= ") and dt < 0.01 * rollouts / 16 + 2.0, (rollouts, dt)     # < 10 ms per step of 16 rollouts
        out[rollouts] = dt
        record_property("genmove_seconds_%d_rollouts" % rollouts, dt)
    print("genmove latency 19x19 (no net cost): %s" % {k: "%.3f s = %.2f ms per 16-rollout step" % (v, v / (k / 16) * 1e3) for k, v in out.items()})
