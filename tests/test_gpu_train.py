"""GPU: training-side replay loader (k_replay_extract, elftrain_*) and self-play record emission (elfsp_pop_record) through the
C ABI, against the golden rows / records dumped by the REAL reference (oracle/gen_golden_train.py).  Bar: bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from sp_drive import drive_stub, sp_from_fixture_cfg
from pyoracle import Port, port_train_sample, sgfstr2coords, stub_net

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def elf(built):
    import elf_amd
    return elf_amd


def load_rows(n):
    g = np.load(os.path.join(GOLDEN, "train_%s.npz" % n))
    return g, [str(t) for t in g["records"]]


# keep: the reference's own procedure (reset + forward x move_to; the replayed state stays in the engine) / the trainer's mode
# (from the record's checkpoint, at most 31 forwards).  "9_superko": a record that continues past a positional repetition
@pytest.mark.parametrize("n", [9, 19, "9_superko"])
@pytest.mark.parametrize("fmt", ["f32_nchw", "f16_nhwc"])
@pytest.mark.parametrize("keep", [False, True])
def test_train_batch_matches_reference_rows(elf, n, fmt, keep):
    import torch
    g, recs = load_rows(n)
    n = int(g["board_size"])
    for nfa in (1, 3):
        sel = np.nonzero(g["nfa"] == nfa)[0]
        ld = elf.ReplayLoader(board_size=n, capacity=len(recs) + 3, batchsize=2 * len(sel), num_future_actions=nfa, feature_format=fmt, keep_states=keep)
        for i, t in enumerate(recs):
            ld.put(i + 2, t)                       # slots need not start at 0
        assert len(ld) == len(recs)
        b = ld.extract(g["rec"][sel] + 2, g["move_to"][sel], g["d4"][sel])
        torch.cuda.synchronize()
        s = b["s"].float().cpu().numpy()
        want_s = np.stack([np.unpackbits(g["s"][i])[: 18 * n * n].reshape(18, n, n) for i in sel]).astype(np.float32)
        assert np.array_equal(s, want_s)
        assert np.array_equal(b["offline_a"].cpu().numpy(), g["offline_a"][sel][:, :nfa])
        assert np.array_equal(b["winner"].cpu().numpy(), g["winner"][sel])
        assert np.array_equal(b["predicted_value"].cpu().numpy().view(np.uint32), g["predicted_value"][sel].view(np.uint32))
        assert np.array_equal(b["move_idx"].cpu().numpy(), g["move_idx"][sel])
        assert np.array_equal(b["num_move"].cpu().numpy(), g["num_move"][sel])
        assert np.array_equal(b["aug_code"].cpu().numpy(), g["aug_code"][sel])
        assert np.array_equal(b["selfplay_ver"].cpu().numpy(), g["selfplay_ver"][sel])
        ms, want = b["mcts_scores"].cpu().numpy(), g["mcts_scores"][sel]
        np.testing.assert_array_equal(ms, want)    # NaN rows (all-zero recorded policy: 0/0 in the reference too) compare equal
        fin = np.isfinite(want)
        assert np.array_equal(ms[fin].view(np.uint32), want[fin].view(np.uint32))
        if keep:
            # the replayed GoState of sample i sits in board slot i of the loader's engine
            info = ld.engine.info(n=len(sel)).cpu().numpy()
            assert np.array_equal(info[:, 0] - 1, g["move_idx"][sel])
        # a record put again (other slot, then the same slot): its checkpoints are rewritten, the rows stay
        if not keep and nfa == 1:
            ld.put(0, recs[0])
            ld.put(2, recs[0])
            first = np.nonzero(g["rec"][sel] == 0)[0]
            b2 = ld.extract(np.concatenate([np.zeros(len(first), np.int32), np.full(len(first), 2, np.int32)]),
                            np.tile(g["move_to"][sel][first], 2), np.tile(g["d4"][sel][first], 2))
            s2 = b2["s"].float().cpu().numpy()
            assert np.array_equal(s2[: len(first)], want_s[first]) and np.array_equal(s2[len(first):], want_s[first])
        ld.close()


@pytest.mark.parametrize("name", ["train_act_9", "train_act_9_evict", "train_act_19"])
def test_replay_buffer_batches_equal_the_reference_trainer_path(elf, name):
    """The trainer's input path end to end against the REAL one: the fixture is what one real GoGameTrain game thread
    (train/game_train.cc:23-58) sent to the "train" batch group from a real ReaderQueuesT<Record> filled with InsertWithParity
    (oracle/ref_selfplay.cc reftrain_act).  ReplayBuffer given the same records in the same order, the same queue shape and
    seeds: same rows in the same order -- record and move drawn, D4 code, the 18 planes, offline_a, winner, mcts_scores, move_idx,
    num_move, selfplay_ver -- including evictions from a full queue and records too short for num_future_actions."""
    import torch
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [int(v) for v in g["cfg_vals"]]))
    n, recs = int(g["board_size"]), [str(t) for t in g["records"]]
    rows = 64 * cfg["num_acts"]
    for fmt in ("f32_nchw", "f16_nhwc"):
        rb = elf.ReplayBuffer(board_size=n, num_reader=cfg["num_reader"], queue_min_size=cfg["q_min_size"], queue_max_size=cfg["q_max_size"],
                              batchsize=rows, insert_seed=cfg["insert_seed"], num_threads=1, seed=cfg["game_seed"],
                              num_future_actions=cfg["num_future_actions"], feature_format=fmt)
        for t in recs:
            rb.insert(t)
        b = rb.sample(cfg["num_acts"])
        torch.cuda.synchronize()
        want_s = np.unpackbits(g["s"], axis=1)[:, : 18 * n * n].reshape(rows, 18, n, n).astype(np.float32)
        assert np.array_equal(b["s"].float().cpu().numpy(), want_s)
        for k in ("offline_a", "winner", "move_idx", "num_move", "aug_code", "selfplay_ver"):
            assert np.array_equal(b[k].cpu().numpy(), g[k]), (name, k)
        ms, want = b["mcts_scores"].cpu().numpy(), g["mcts_scores"]
        np.testing.assert_array_equal(ms, want)
        fin = np.isfinite(want)
        assert np.array_equal(ms[fin].view(np.uint32), want[fin].view(np.uint32))
        rb.close()


def test_client_message_of_a_selfplay_context_is_read_by_the_reference_server(elf):
    """The loop of a self-play client (train/distri_client.h): games play, their thread states and finished games collect in a
    ClientRecords (GuardedRecords), and the message it dumps is parsed by the REAL Records::createFromJsonString -- the call
    TrainCtrl::OnReceive makes on the training server -- into the same number of records and states."""
    import torch
    from pyoracle import RefSelfPlay
    n, G = 9, 4
    sp = elf.SelfPlay(board_size=n, num_games=G, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, seed=17, move_cutoff=12,
                      keep_records=16, nodes_per_game=1024, model_ver=6, game_idx_base=100)
    c = elf.ClientRecords("mi355x-0")
    finished = 0
    while finished < 6:
        rows = sp.begin_step()
        pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), 4, 0)
        sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device), torch.full((rows,), 6, dtype=torch.int64, device=sp.device))
        before = len(c)
        c.update_from(sp)
        finished += len(c) - before
    plies = sp.board_engine().info_host()["ply"][:G]
    text = c.dump_and_clear()
    j = json.loads(text)
    assert j["identity"] == "mi355x-0" and len(j["records"]) == finished and len(j["states"]) == G
    by_id = {t["thread_id"]: t for t in j["states"]}
    assert sorted(by_id) == [100, 101, 102, 103]
    for g in range(G):
        t = by_id[100 + g]
        assert t["move_idx"] == int(plies[g]) - 1 and t["black"] == 6 and t["white"] == -1 and t["seq"] >= 2
    assert all(r["request"]["vers"]["black_ver"] == 6 and r["thread_id"] in by_id for r in j["records"])
    if RefSelfPlay.available(n):
        got = RefSelfPlay(n).records_parse(text)
        assert got == (finished, G, sum(t["move_idx"] for t in j["states"]), "mi355x-0")
    # the server's reply: a MsgRequestSeq text with a new model and other search options (more rollouts per step than the context
    # was created with: the row buffer grows) -> parsed -> sent to the games -> they restart under it
    import ctypes as C
    from elf_amd.client import TsOptions
    from elf_amd.selfplay import SpRequest
    reply = elf.request_seq_to_json(SpRequest(7, -1, 0.05, 0.05, 0.1, -1, 0, 0, 1),
                                    TsOptions(0, 2, 24, 12, 0, 0, 1, 0, 0, 0.0, 0.0, 2, 1, 0, 0, 0.75, b""), 41)
    if RefSelfPlay.available(n):
        assert RefSelfPlay(n).request_seq_roundtrip(reply) == reply          # the reference reads and rewrites it unchanged
    q, seq, ts = elf.parse_request_seq(reply)
    assert seq == 41 and q.black_ver == 7
    L = elf.lib()
    bv, wv = C.c_int64(0), C.c_int64(0)
    L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv))
    sp.send_request(q, ts)
    assert sp.max_rows >= G * 2 * 12
    ver, guard = 6, 0
    while sp.progress()["searches"] < 200 and guard < 400:
        rows = sp.begin_step()
        if L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv)):
            ver = bv.value
        pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), 4, 0)
        sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device), torch.full((rows,), ver, dtype=torch.int64, device=sp.device))
        c.update_from(sp)
        guard += 1
        if ver == 7 and len(c) >= 2:
            break
    assert ver == 7 and rows == G * 2 * 12                                    # every game plays under the new options: T x K rows each
    j2 = json.loads(c.dump_and_clear())
    new = [r for r in j2["records"] if r["request"]["vers"]["black_ver"] == 7]
    assert new and all(r["request"]["vers"]["mcts_opt"]["num_rollouts_per_thread"] == 24 and r["request"]["vers"]["mcts_opt"]["num_threads"] == 2
                       and r["request"]["client_ctrl"]["black_resign_thres"] == float(np.float32(0.05)) for r in new)
    assert all(t["black"] == 7 for t in j2["states"])
    sp.close()
    c.close()


def test_replayed_positions_and_draws(elf):
    """sample(): draws within range and reproducible for a seed; replayed boards equal the oracle's replay (hash, legal mask)."""
    import torch
    n = 19
    g, recs = load_rows(n)
    parsed = [json.loads(t) for t in recs]
    port = Port(n)
    nfa = 2
    outs = []
    for rep in range(2):
        # rep 0: the reference's procedure with the states kept for inspection; rep 1: from the checkpoints -- same draws, same batch
        ld = elf.ReplayLoader(board_size=n, capacity=len(recs), batchsize=96, num_future_actions=nfa, seed=42, keep_states=(rep == 0))
        for i, t in enumerate(recs):
            ld.put(i, t)
        b = ld.sample(96)
        torch.cuda.synchronize()
        d = ld._draw.cpu().numpy()
        outs.append((d.copy(), {k: v.clone() for k, v in b.items()}))
        if rep == 0:
            info = ld.engine.info(n=96).cpu().numpy()
            mask = ld.engine.legal_mask(n=96).cpu().numpy()
            for i in range(96):
                r, mt, d4 = (int(d[k, i]) for k in range(3))
                nm = parsed[r]["result"]["num_move"]
                assert 0 <= mt <= nm - nfa and 0 <= d4 < 8
                o = port_train_sample(port, parsed[r], mt, d4, nfa)
                assert np.array_equal(b["s"][i].cpu().numpy(), o["s"]), i
                assert np.array_equal(b["offline_a"][i].cpu().numpy(), o["offline_a"]), i
                st = port.new()
                for c in sgfstr2coords(n, parsed[r]["result"]["content"])[:mt]:
                    port.forward(st, int(c))
                h = (int(info[i, 13]) & 0xFFFFFFFF) | ((int(info[i, 14]) & 0xFFFFFFFF) << 32)
                assert h == port.hash(st) and np.array_equal(mask[i], port.legal_mask(st)), i
                port.free(st)
        ld.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    assert len(set(outs[0][0][0].tolist())) > 1 and len(set(outs[0][0][2].tolist())) == 8


def test_loader_argument_errors(elf):
    import ctypes as C
    L = elf.lib()
    ld = elf.ReplayLoader(board_size=9, capacity=2, batchsize=4, with_policies=False)
    mv = np.array([12, 13], np.uint16)
    pol = np.zeros((1, 121), np.uint8)
    assert L.elftrain_put(ld._h, 5, mv.ctypes.data, 2, C.c_float(1), 0, None, 0, None, 0) == -1      # slot out of range
    assert L.elftrain_put(ld._h, 0, mv.ctypes.data, 9999, C.c_float(1), 0, None, 0, None, 0) == -1   # too many moves
    assert L.elftrain_put(ld._h, 0, mv.ctypes.data, 2, C.c_float(1), 0, pol.ctypes.data, 1, None, 0) == -1   # store has no policies
    assert L.elftrain_put(ld._h, 0, mv.ctypes.data, 2, C.c_float(1), 0, None, 0, None, 0) == 0
    with pytest.raises(ValueError):
        ld.extract([0] * 5, [0] * 5, [0] * 5)
    import torch
    d = torch.zeros(3, 4, dtype=torch.int32, device="cuda")
    assert L.elftrain_draw(ld._h, 4, 3, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()), C.c_void_p(d[2].data_ptr()), None) == -1  # no record long enough
    ld.close()


@pytest.mark.parametrize("name", ["records_9_cutoff", "records_9_resign", "records_9_twopass", "records_9_neverresign", "records_9_preload", "records_19_resign", "records_19_cutoff", "records_19_sgf_preload",
                                  "records_9_eval", "records_9_eval_swap_resign",
                                  "records_9_cheat_selfplay", "records_9_cheat_eval", "records_9_cheat_eval_swap", "records_9_sgf_policy_only"])
def test_selfplay_records_equal_reference_dump(elf, name):
    """GPU self-play under the fixture's configuration leaves the same Record JSON text as the reference's
    GoStateExt::dumpRecord for every finished game (content, quantised policies, predicted values, reward, seq), timestamp aside."""
    import torch
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    want = [json.loads(str(t)) for t in g["records"]]
    sp = sp_from_fixture_cfg(elf, n, cfg, keep_records=8, nodes_per_game=4096, log_searches=int(g["searches"]))
    if "preload_moves" in g.files:
        sp.preload(g["preload_moves"], int(g["preload_move_to"]))
    got = []

    def collect(sp, rows_total):
        got.extend(sp.pop_records())

    drive_stub(sp, n, cfg, lambda sp: len(got) >= len(want), collect)
    _check_run_against_fixture(sp, g, got, want, name)
    sp.close()


def _check_run_against_fixture(sp, g, got, want, name):
    assert len(got) >= len(want)
    # every search of the run, across game ends and restarts: root edges in iteration order, visits, priors, rewards, move played
    rec, coord, visits, prior, reward = sp.search_log()
    m = min(len(rec), int(g["searches"]))
    assert m > 20
    for i in range(m):
        ne = int(g["n_edges"][i])
        ctx = "%s search %d" % (name, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], g["coord"][i, :ne].astype(np.int32)), ctx
        assert np.array_equal(visits[i, :ne], g["visits"][i, :ne]), ctx
        assert np.array_equal(prior[i, :ne].view(np.uint32), g["prior"][i, :ne].view(np.uint32)), ctx
        assert np.array_equal(reward[i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32)), ctx
        assert rec[i].move_played == int(g["move_played"][i]) and rec[i].best_action == int(g["best_action"][i]), ctx
        assert np.float32(rec[i].root_value) == g["root_value"][i], ctx
    for t, w in zip(got, want):
        j = json.loads(t)
        assert j["timestamp"] > 0
        j["timestamp"] = w["timestamp"]
        for k in w["result"]:
            a, b = j["result"].get(k), w["result"][k]
            if isinstance(b, list) and isinstance(a, list) and a != b:
                d = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
                raise AssertionError((name, w["seq"], k, len(a), len(b), d[:5], [(a[i], b[i]) for i in d[:3]] if k != "policies" else "..."))
            assert a == b, (name, w["seq"], k, a, b)
        assert j == w
        assert json.dumps(j, separators=(",", ":"), sort_keys=True) == json.dumps(w, separators=(",", ":"), sort_keys=True)
        t2 = t.replace('"timestamp":%d' % json.loads(t)["timestamp"], '"timestamp":%d' % w["timestamp"])
        assert t2 == str(g["records"][got.index(t)])   # text-identical to the reference's json::dump()


@pytest.mark.parametrize("name", ["records_9_req2_restart", "records_9_req2_async", "records_9_req2_ts", "records_9_req2_eval"])
def test_second_request_while_a_game_plays_equals_reference(elf, name):
    """The reference run of the fixture got a second request (black_ver 3 -> 4) into the game's mailbox during its eighth search
    (oracle/ref_selfplay.cc req2_*).  GoGameSelfPlay::act reads the mailbox at every fifth act only (game_selfplay.cc:273-289), so
    searches 8 and 9 still run under the old request; at the eleventh act OnReceive (:222-270) either restarts the game from the
    empty board with new AIs, nothing recorded, seq advanced (other versions) or lets it go on (async; the record then names both
    models).  Same search log, same record texts, same "game_start" batches as the reference.
    *_ts: the second request has the same model but other search options (ModelPair.mcts_opt: the reference's server dictates them);
    *_eval: it is an evaluation request as EvalSubCtrl writes it (train/ctrl_eval.h:227-237: a second AI, Dirichlet noise and the
    q_zero flags off) -- the games restart and search with the request's options."""
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    want = [json.loads(str(t)) for t in g["records"]]
    sp = sp_from_fixture_cfg(elf, n, cfg, keep_records=8, nodes_per_game=4096, log_searches=int(g["searches"]))
    L = elf.lib()
    got, starts, state = [], [], dict(sent=False, ver=int(cfg["black_ver"]))

    state["wver"] = int(cfg["white_ver"])

    def model_version():
        # a request restarts the games at the top of an act, i.e. inside begin_step: the "game_start" it makes due comes before the
        # rows of the new games are answered (its callback loads the models the batch names, selfplay.py)
        bv, wv = C.c_int64(-9), C.c_int64(-9)
        for _ in range(L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv))):
            starts.append(bv.value)
            state["ver"], state["wver"] = bv.value, wv.value
        return state["ver"], state["wver"]

    def on_step(sp, rows_total):
        got.extend(sp.pop_records())
        if not state["sent"] and sp.progress()["searches"] >= int(cfg["req2_after_searches"]):
            ts = None
            if int(cfg.get("req2_ts", 0)):
                from elf_amd.client import TsOptions
                f32 = lambda k: float(np.float32(cfg[k]))
                ts = TsOptions(0, int(cfg["mcts_threads"]), int(cfg["req2_rollouts_per_thread"]), int(cfg["req2_rollouts_per_batch"]), 0, 0,
                               int(cfg["persistent_tree"]), int(cfg["pick_method"]), 0, f32("req2_root_epsilon"), f32("req2_root_alpha"),
                               int(cfg["virtual_loss"]), int(cfg["use_prior"]), int(cfg["req2_unexplored_q_zero"]),
                               int(cfg["req2_root_unexplored_q_zero"]), f32("req2_c_puct"), b"")
            sp.set_request(int(cfg["req2_black_ver"]), int(cfg.get("req2_white_ver", -1)), float(np.float32(cfg["resign_thres"])),
                           float(np.float32(cfg["never_resign_prob"])), async_=bool(cfg["req2_async"]), num_game_thread_used=1, mcts_opt=ts)
            state["sent"] = True

    drive_stub(sp, n, cfg, lambda sp: len(got) >= len(want) and sp.progress()["searches"] >= int(g["searches"]), on_step,
               black_ver=model_version)
    assert starts == g["start_versions"].tolist() and len(starts) == int(g["game_starts"])
    _check_run_against_fixture(sp, g, got, want, name)
    sp.close()


def test_selfplay_soak_records_replay_on_the_oracle(elf):
    """Many concurrent games through hundreds of game ends and restarts (move limit, two passes, resignation) with a random
    net: every record the engine emits must replay move by move on the CPU oracle (each move legal, same ply), non-resigned
    games must carry the oracle's Tromp-Taylor result, and the node pools must not leak across restarts."""
    import ctypes as C
    import torch
    n, G = 9, 96
    sp = elf.SelfPlay(board_size=n, num_games=G, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, mcts_puct=1.5,
                      mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=20,
                      policy_distri_cutoff=8, resign_thres=0.35, never_resign_prob=0.2, seed=4321, nodes_per_game=1024,
                      keep_records=100000)
    gen = torch.Generator(device="cuda").manual_seed(11)
    recs = []
    for step in range(6000):
        rows = sp.begin_step()
        pi = torch.softmax(3.0 * torch.randn((sp.max_rows, n * n + 1), device="cuda", generator=gen), dim=1)
        v = torch.tanh(0.8 * torch.randn((sp.max_rows,), device="cuda", generator=gen))
        sp.end_step(pi, v)
        if step % 64 == 0:
            recs += sp.pop_records()
            if len(recs) >= 400:
                break
    recs += sp.pop_records()
    st = sp.stats()
    assert len(recs) >= 400 and st["games"] == len(recs)
    port = Port(n)
    ends = {"resign": 0, "limit": 0, "twopass": 0}
    for t in recs:
        j = json.loads(t)
        res = j["result"]
        mv = sgfstr2coords(n, res["content"])
        assert len(mv) == res["num_move"] and len(res["values"]) in (len(mv), len(mv) + 1)
        s = port.new()
        for c in mv:
            assert port.forward(s, int(c)) == 1
        if len(res["values"]) == len(mv) + 1:          # the last search ended in a resignation: no move, reward +-1
            assert res["reward"] in (1.0, -1.0) and len(mv) + 1 >= 50
            assert res["reward"] == (1.0 if len(mv) % 2 == 1 else -1.0)     # the side to move resigned
            ends["resign"] += 1
        else:
            assert port.terminated(s) and res["reward"] == port.evaluate(s, 7.5)
            ends["twopass" if (len(mv) >= 2 and mv[-1] == 0 and mv[-2] == 0) else "limit"] += 1
        assert len(res.get("policies", [])) == min(8, len(res["values"]))
        port.free(s)
    assert ends["resign"] > 0 and ends["limit"] + ends["twopass"] > 0, ends
    # node pools: after the last move every game holds at most the kept subtree; nothing leaked over ~400 restarts
    L = elf.lib()
    info = torch.zeros((G, 8), dtype=torch.int32, device="cuda")
    assert L.elfmcts_root(L.elfsp_mcts(sp._h), C.c_void_p(info.data_ptr()), None, None, None, None, None, None) == 0
    torch.cuda.synchronize()
    live = info[:, 7].cpu().numpy()                       # RootInfo word 7: node ids the game's tree holds
    assert (live <= 32 * 4 + 64).all() and (live >= 1).all(), (live.min(), live.max())
    p = sp.pool_info()                                    # the context's shared pool: every id is in a tree or free
    assert p["live"] == int(live.sum()) == int(sp.count_live().sum()) and p["live"] + p["small_free"] + p["big_free"] == p["small_total"] + p["big_total"], p
    assert sp.validate_trees()[0] == 0
    sp.close()


def test_prefetched_batches_equal_single_batch_launches(elf):
    """ReplayLoader(batches_per_launch = k): one k_replay_extract launch draws and extracts k train batches (the trainer
    prefetches; the replay kernel is a latency chain per sample and wants many samples in flight).  Same samples, in the same order,
    bit for bit, as k single-batch launches of a loader with the same seed."""
    import torch
    n, B, k = 9, 64, 3
    g, recs = load_rows(n)
    outs = []
    for kk in (1, k):
        ld = elf.ReplayLoader(board_size=n, capacity=len(recs), batchsize=B, device=0, num_future_actions=1, seed=77, batches_per_launch=kk)
        for i, t in enumerate(recs):
            ld.put(i, t)
        if kk == 1:
            got = [{key: v.clone() for key, v in ld.sample().items()} for _ in range(k)]
        else:
            got = [{key: v.clone() for key, v in b.items()} for b in ld.sample_batches()]
        torch.cuda.synchronize()
        outs.append(got)
        ld.close()
    for a, b in zip(*outs):
        assert set(a) == set(b)
        for key in a:
            # bit patterns: a record whose recorded policy sums to zero yields NaN rows (0 / 0, as in the reference), and NaN != NaN
            va = a[key].contiguous().view(torch.int32 if a[key].element_size() == 4 else torch.int64 if a[key].element_size() == 8 else torch.int16)
            vb = b[key].contiguous().view(va.dtype)
            assert a[key].shape[0] == B and torch.equal(va, vb), key


def test_evaluation_games_soak_with_requests(elf):
    """Evaluation games under churn: 48 games with two AIs (different step counts, so the games drift out of step), Black's AI
    playing policy-only, resignations and cutoffs, a player_swap request and a thread-count request arriving mid-run.  Every
    record must replay on the CPU oracle, carry the request it was played under, the tree records must keep their invariants in
    both pools, and no node may leak."""
    import ctypes as C
    import torch
    n, G = 9, 48
    sp = elf.SelfPlay(board_size=n, num_games=G, mcts_rollout_per_thread=48, mcts_rollout_per_batch=16, mcts_puct=1.5, mcts_virtual_loss=1,
                      mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=20, policy_distri_cutoff=8,
                      seed=77, nodes_per_game=1024, keep_records=100000, white_mcts_rollout_per_thread=32, white_mcts_rollout_per_batch=8,
                      white_puct=1.0, black_use_policy_network_only=True, move_cutoff=60)
    sp.set_request(4, 5, resign_thres=0.35, never_resign_prob=0.2)
    gen = torch.Generator(device="cuda").manual_seed(3)
    L = elf.lib()
    recs, mixed = [], 0
    for step in range(20000):
        rb, rw = sp.begin_step2()
        rep = [None, None]
        for a, rows in enumerate((rb, rw)):
            if rows:
                pi = torch.softmax(3.0 * torch.randn((rows, n * n + 1), device="cuda", generator=gen), dim=1)
                v = torch.tanh(0.8 * torch.randn((rows,), device="cuda", generator=gen))
                rep[a] = (pi, v, None)
        sp.end_step2(rep)
        mixed += rb > 0 and rw > 0          # both AIs had leaves in this step: games are out of step with each other
        if step == 300:
            sp.set_request(4, 5, resign_thres=0.35, never_resign_prob=0.2, player_swap=True)     # every game restarts, AIs swap colours
        if step == 900:
            sp.set_request(4, 5, resign_thres=0.35, never_resign_prob=0.2, player_swap=True, num_game_thread_used=G - 8)   # 8 games go idle
        if step % 97 == 0:
            for a in range(2):
                out = np.zeros(5, np.int32)
                assert L.elfmcts_validate(L.elfsp_mcts_actor(sp._h, a), out.ctypes.data) == 0 and out[0] == 0, (step, a, out)
            recs += sp.pop_records()
            if len(recs) >= 150 and step > 1200:
                break
    recs += sp.pop_records()
    assert len(recs) >= 150 and mixed > 0
    assert sp.progress()["waiting"] == 8
    port = Port(n)
    swapped = 0
    for t in recs:
        j = json.loads(t)
        res, ctrl = j["result"], j["request"]["client_ctrl"]
        assert j["request"]["vers"]["black_ver"] == 4 and j["request"]["vers"]["white_ver"] == 5 and res["using_models"] == [4, 5]
        swapped += bool(ctrl["player_swap"])
        mv = sgfstr2coords(n, res["content"])
        assert len(mv) == res["num_move"]
        s = port.new()
        for c in mv:
            assert port.forward(s, int(c)) == 1
        if len(res["values"]) == len(mv) + 1:
            assert res["reward"] == (1.0 if len(mv) % 2 == 1 else -1.0) and len(mv) + 1 >= 50
        else:
            assert res["reward"] == port.evaluate(s, 7.5)
        port.free(s)
    assert 0 < swapped < len(recs)
    # node pools of both AIs: nothing leaked over the restarts
    for a in range(2):
        info = torch.zeros((G, 8), dtype=torch.int32, device="cuda")
        assert L.elfmcts_root(L.elfsp_mcts_actor(sp._h, a), C.c_void_p(info.data_ptr()), None, None, None, None, None, None) == 0
        torch.cuda.synchronize()
        live = info[:, 7].cpu().numpy()
        assert (live <= 48 * 4 + 64).all() and (live >= 1).all(), (a, live.min(), live.max())
        p = sp.pool_info(actor=a)
        assert p["live"] == int(live.sum()) == int(sp.count_live(actor=a).sum()), (a, p)
        assert p["live"] + p["small_free"] + p["big_free"] == p["small_total"] + p["big_total"], (a, p)
    sp.close()
