"""GPU: device-resident MCTS + self-play host loop vs golden fixtures produced by the REAL reference stack
(oracle/gen_golden_mcts.py: elf::Context + GoGameSelfPlay + MCTSGoAI + tree_search/*.h), same stub net.
Bar: bit-exact -- root edge ORDER (unordered_map iteration order), priors (after Dirichlet), visit counts,
accumulated rewards, most-visited action, sampled move, root value, for every search of the fixture."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import stub_net
from sp_drive import drive_stub, sp_from_fixture_cfg

pytestmark = pytest.mark.gpu


def run_case(elf_amd, name, max_searches=None):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    m = len(g["move_played"]) if max_searches is None else min(max_searches, len(g["move_played"]))
    sp = sp_from_fixture_cfg(elf_amd, n, cfg, log_searches=m)
    if "fixed_time" in g.files:       # uniform_random: the value time(NULL) gave the reference's pick generator
        sp.set_pick_seed(int(g["fixed_time"]))
    if "preload_moves" in g.files:    # GameOptions.preload_sgf / preload_sgf_move_to (game_selfplay.cc:202-219,392-405)
        sp.preload(g["preload_moves"], int(g["preload_move_to"]))

    def check_trees(sp, rows_total):
        if sum(rows_total) < 4096:   # the node records' own invariants (scoring order, child back links, visit sums) after every step
            assert sp.validate_trees()[0] == 0, "%s: node record invariants %s" % (name, sp.validate_trees())

    rows_total = drive_stub(sp, n, cfg, lambda sp: sp.stats()["logged"] >= m, check_trees)
    rec, coord, visits, prior, reward = sp.search_log()
    na = n * n + 1
    for i in range(m):
        ne = int(g["n_edges"][i])
        ctx = "%s search %d" % (name, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], g["coord"][i, :ne].astype(np.int32)), ctx + ": edge iteration order"
        assert np.array_equal(prior[i, :ne].view(np.uint32), g["prior"][i, :ne].view(np.uint32)), ctx + ": priors"
        assert np.array_equal(visits[i, :ne], g["visits"][i, :ne]), ctx + ": visit counts"
        assert np.array_equal(reward[i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32)), ctx + ": rewards"
        assert rec[i].best_action == int(g["best_action"][i]), ctx
        assert rec[i].total_visits == int(g["total_visits"][i]), ctx
        assert np.float32(rec[i].root_value) == g["root_value"][i], ctx
        assert rec[i].move_played == int(g["move_played"][i]), ctx + ": move played"
    assert na <= coord.shape[1]
    if max_searches is None and "white_rows" in g.files and int(cfg["white_ver"]) >= 0:
        assert rows_total[1] > 0          # the second AI's rows went to their own batch group
    sp.close()
    return sum(rows_total)


@pytest.fixture(scope="module")
def elf(built):
    import elf_amd
    return elf_amd


@pytest.mark.parametrize("name", ["mcts_19_r128_fresh", "mcts_19_r256_dir", "mcts_19_r256_ties", "mcts_19_r512_client",
                                  "mcts_9_r512", "mcts_9_r64_ties", "mcts_19_r128_vl0", "mcts_19_r128_noprior", "mcts_9_r128_rootq0",
                                  "mcts_9_r96_bs4", "mcts_9_r128_bs64"])
def test_search_matches_reference(elf, name):
    run_case(elf, name)


@pytest.mark.parametrize("name", ["mcts_9_eval_two_ai", "mcts_19_eval_swap", "mcts_9_policy_only_eval"])
def test_two_ai_games_match_reference(elf, name):
    """A request with white_ver >= 0: a second MCTSGoAI plays White (game_selfplay.cc:165-185, 364-366) with its own tree,
    model ("actor_white" rows, own version in rv), rng stream and the white_* overrides; player_swap exchanges the two; both
    trees follow every move.  Fixtures from the real reference stack with two different stub nets."""
    run_case(elf, name)


@pytest.mark.parametrize("name", ["mcts_9_pick_prior", "mcts_9_policy_only_white"])
def test_pick_method_and_policy_only_match_reference(elf, name):
    """TSOptions.pick_method = strongest_prior (tree_search.h:506-509) and GameOptions.white_use_policy_network_only
    (MCTSAI_T::actPolicyOnly, mcts.h:83-90 / runPolicyOnly tree_search.h:385-407)."""
    run_case(elf, name)


def test_pick_method_uniform_random_matches_reference(elf):
    """TSOptions.pick_method = uniform_random (tree_search.h:514-517): random_idx = rng() % edges from MCTSResultT::addActions'
    process-wide generator seeded with time(NULL) (tree_search_base.h:238).  The fixture is a reference process whose time() was
    held at a known value (oracle/ref_selfplay.cc refsp_set_time); the context seeded with it picks the same edges: 40 searches,
    best action, uniform policy, sampled opening moves, game restarts."""
    run_case(elf, "mcts_9_pick_uniform")


@pytest.mark.parametrize("name", ["mcts_9_r256_bs128", "mcts_19_r512_bs256", "mcts_9_r1024_bs512"])
def test_more_than_64_rollouts_per_batch(elf, name):
    """num_rollouts_per_batch 128 / 256 / 512 (tree_search_options.h:81 has no bound): the leaf table of a step is sized by the
    launch (up to 1024 leaves = num_threads x num_rollouts_per_batch)."""
    run_case(elf, name)


@pytest.mark.parametrize("name", ["mcts_9_T2_r128", "mcts_9_T4_r256", "mcts_19_T2_r512", "mcts_19_T8_client", "mcts_9_T3_eval_two_ai"])
def test_search_threads_match_the_turnstile_reference(elf, name):
    """TSOptions.num_threads = 2, 3, 4, 8 against the REAL reference.  The reference's search threads race on the shared tree
    (tree_search.h:345-368), so these fixtures come from its turnstile build (oracle/Makefile, libelfsp*_ts.so: four elf_ts_hook()
    calls inserted into a build-time copy of TreeSearchSingleThreadT::batch_rollouts, nothing else changed), which makes the threads of
    a search take turns in one fixed order per round: descents of thread 0 .. T-1, evaluation, setEvaluation + backup of thread
    0 .. T-1 -- the interleaving k_mcts_select / k_mcts_expand / k_mcts_backup implement.  Every thread's MCTSActor draws its D4 codes
    from its own generator (all seeded alike, game_selfplay.cc:45-47,77): one D4 window per thread on the device.  Covers a whole
    9x9 game (terminal leaves, passes), the client configuration (8 threads x 1 rollout per batch, virtual loss 5: revisits of leaves
    another thread has requested in the same round) and an evaluation game with two 3-thread AIs."""
    run_case(elf, name)


@pytest.mark.parametrize("name", ["mcts_19_sgf_p60", "mcts_19_sgf_p120", "mcts_19_sgf_p180", "mcts_19_sgf_p195", "mcts_19_sgf_b_p150",
                                  "mcts_19_sgf_c_p100", "mcts_19_sgf_T2_p140", "mcts_19_sgf_r8192_p160"])
def test_search_from_dense_sgf_positions_matches_reference(elf, name):
    """north_star: "visit counts under a fixed RNG seed ... on the same SGF positions".  The REAL reference stack preloaded with
    ladder-suite games (GameOptions.preload_sgf, game_selfplay.cc:202-219: 406844.sgf to plies 60 / 120 / 140 / 160 / 180 / 195, and the
    two longest other games of the suite to plies 150 / 100) searches mid- and late-game 19x19 positions: 183..303 legal moves, dozens of
    groups, captures and kos inside the tree, ply_pass_enabled below the preload ply so that the pass edge, Tromp-Taylor leaves
    (go/mcts/mcts.h:232-242) and remove_pass_if_dangerous (:185-207) occur; every searched move is replaced by the SGF's next move
    (:392-405), so the persistent tree advances along the game.  One case with 2 search threads (turnstile build), one with 8192
    rollouts per move, one with the client's virtual loss 5 / puct 0.85 / 8 rollouts per batch, one with fresh trees."""
    run_case(elf, name)


def test_config3_8192_rollouts(elf):
    """BASELINE config 3 search settings: bs 16, 8192 rollouts/move, puct 1.5, vloss 1, eps 0.25 / alpha 0.03."""
    run_case(elf, "mcts_19_r8192")


def test_exhausted_node_pool_is_a_status_code(elf):
    """A tree that outgrows nodes_per_game surfaces as ELFGO_E_MCTS_BASE - ELFMCTS_E_POOL (an exception in Python), not as
    corrupted statistics or a hang."""
    import torch
    sp = elf.SelfPlay(board_size=9, num_games=2, mcts_rollout_per_thread=512, mcts_rollout_per_batch=16, nodes_per_game=64, seed=3)
    g = torch.Generator(device="cuda").manual_seed(0)
    with pytest.raises(elf.ElfGoError) as e:
        for _ in range(64):
            rows = sp.begin_step()
            pi = torch.softmax(torch.randn((sp.max_rows, 82), device="cuda", generator=g), dim=1)
            v = torch.zeros(sp.max_rows, device="cuda")
            sp.end_step(pi, v)
    assert "-101" in str(e.value)
    sp.close()


def test_one_game_may_hold_several_times_its_share_of_the_pool(elf):
    """The node memory of a context is ONE pool of num_games x nodes_per_game ids (the reference takes its nodes from the heap,
    tree_search_node.h:439-467: a game whose kept subtree is large simply holds more).  Eight games, nodes_per_game = 64 for searches
    of 256 rollouts per move on a persistent tree -- a fixed per-game pool of 64 ids cannot hold one search -- and only game 0 plays
    (num_game_thread_used = 1): it must reproduce the reference fixture search for search while its tree holds more than five times
    the per-game share, and every id is accounted for afterwards."""
    name = "mcts_19_r256_dir"
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    m = 8
    cfg["num_games"], cfg["thread_used"] = 8, 1           # DispatcherCallback::OnFirstSend: games 1..7 wait, game 0 is the fixture's game
    sp = sp_from_fixture_cfg(elf, 19, cfg, log_searches=m, nodes_per_game=64)
    if "white_ver" not in cfg:                            # a fixture older than the request keys: the helper sent no request
        sp.set_request(0, -1, float(np.float32(cfg["resign_thres"])), float(np.float32(cfg["never_resign_prob"])), num_game_thread_used=1)
    p0 = sp.pool_info()
    assert p0["small_total"] == 8 * 64 and p0["live"] == 8 and p0["small_free"] == p0["small_total"] - 8

    def accounted(sp, rows_total):
        p, live = sp.pool_info(), sp.count_live()
        assert live.sum() == p["live"] and live[0] == p["live_max_game"] and (live[1:] == 1).all(), (p, live)
        assert p["live"] + p["small_free"] + p["big_free"] == p["small_total"] + p["big_total"], p
    drive_stub(sp, 19, cfg, lambda sp: sp.stats()["logged"] >= m, accounted)
    rec, coord, visits, prior, reward = sp.search_log()
    for i in range(m):
        ne = int(g["n_edges"][i])
        assert rec[i].game == 0 and rec[i].n_edges == ne
        assert np.array_equal(coord[i, :ne], g["coord"][i, :ne].astype(np.int32))
        assert np.array_equal(visits[i, :ne], g["visits"][i, :ne])
        assert np.array_equal(reward[i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32))
        assert rec[i].move_played == int(g["move_played"][i])
    p = sp.pool_info()
    assert p["peak_max_game"] > 5 * 64, p                       # one tree held more than three per-game shares
    assert p["live"] + p["small_free"] + p["big_free"] == p["small_total"] + p["big_total"], p     # every id is in a tree or free
    assert sp.validate_trees()[0] == 0
    sp.close()


def test_wide_steps_on_a_small_pool_do_not_starve_each_other(elf):
    """A game's stash is topped up to several steps' worth of node ids with one atomic -- but never beyond a quarter of its nominal
    share of the pool: 8 games whose step is 256 rollouts wide (4 search threads x 64 per batch) on 512 ids per game would otherwise
    let the first two games' stashes (8 x 256 ids each) empty the pool before the others have popped anything."""
    import torch
    n, G = 9, 8
    sp = elf.SelfPlay(board_size=n, num_games=G, mcts_rollout_per_thread=64, mcts_rollout_per_batch=64, mcts_threads=4, nodes_per_game=512,
                      seed=11, move_cutoff=12, policy_distri_cutoff=4)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(40):                       # 40 moves of every game, across game ends and restarts
        rows = sp.begin_step()
        pi = torch.softmax(2.0 * torch.randn((sp.max_rows, n * n + 1), device="cuda", generator=gen), dim=1)
        v = torch.tanh(torch.randn(sp.max_rows, device="cuda", generator=gen))
        sp.end_step(pi[:max(rows, 1)], v[:max(rows, 1)]) if rows else sp.end_step(None, None)
    p = sp.pool_info()
    assert p["live"] + p["small_free"] + p["big_free"] == p["small_total"] + p["big_total"], p
    assert sp.validate_trees()[0] == 0
    sp.close()


LIVE = [
    (9, dict(rollouts_per_thread=96, max_searches=40, seed=4242, net_salt=77, policy_distri_cutoff=9, virtual_loss=2, c_puct=1.1,
             root_epsilon=0.3, root_alpha=0.2, ply_pass_enabled=12, komi=6.5)),
    (9, dict(rollouts_per_thread=48, rollouts_per_batch=12, batchsize=12, max_searches=60, seed=31337, net_salt=78, net_tie_levels=6,
             persistent_tree=0, unexplored_q_zero=1, move_cutoff=25)),
    (19, dict(rollouts_per_thread=80, max_searches=7, seed=2718, net_salt=79, policy_distri_cutoff=3, virtual_loss=4, c_puct=2.0)),
]


@pytest.mark.parametrize("case", range(len(LIVE)))
def test_live_differential_against_the_reference_stack(elf, case):
    """Not a committed fixture: the REAL reference self-play stack (oracle/_ref/libelfsp*.so, prebuilt, travels with the repo)
    is run here on the host cores with a configuration no fixture uses, and the GPU engine must reproduce every search of it.
    Where the prebuilt reference is absent (a fresh clone), its CPU restatement (oracle/mcts_oracle.cc, pinned on the reference's
    fixtures) takes its place, so the test never skips."""
    import torch
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    n, kw = LIVE[case]
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(kw)
    if not RefSelfPlay.available(n):       # not silent: the report says which checker ran
        import warnings
        warnings.warn("oracle/_ref/libelfsp%d.so is absent: this run compares with the CPU restatement (oracle/mcts_oracle.cc), not with the "
                      "compiled reference" % n)
    ref = (RefSelfPlay(n) if RefSelfPlay.available(n) else PortSelfPlay(n)).run(**cfg)
    S = ref["search"]
    m = len(S)
    assert m == cfg["max_searches"]
    sp = elf.SelfPlay(
        board_size=n, num_games=1, device=0, mcts_rollout_per_thread=cfg["rollouts_per_thread"], mcts_rollout_per_batch=cfg["rollouts_per_batch"],
        mcts_puct=cfg["c_puct"], mcts_virtual_loss=cfg["virtual_loss"], mcts_use_prior=bool(cfg["use_prior"]),
        mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=cfg["root_epsilon"], mcts_alpha=cfg["root_alpha"],
        mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]), mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]),
        komi=cfg["komi"], ply_pass_enabled=cfg["ply_pass_enabled"], policy_distri_cutoff=cfg["policy_distri_cutoff"],
        move_cutoff=cfg["move_cutoff"], resign_thres=cfg["resign_thres"], never_resign_prob=cfg["never_resign_prob"], seed=cfg["seed"],
        log_searches=m, nodes_per_game=4096)
    while sp.stats()["logged"] < m:
        rows = sp.begin_step()
        if rows:
            pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), cfg["net_salt"], cfg["net_tie_levels"])
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)
    rec, coord, visits, prior, reward = sp.search_log()
    for i in range(m):
        ne = S[i].n_edges
        ctx = "live case %d search %d" % (case, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], ref["coord"][i, :ne]), ctx
        assert np.array_equal(visits[i, :ne], ref["visits"][i, :ne]), ctx
        assert np.array_equal(prior[i, :ne].view(np.uint32), ref["prior"][i, :ne].view(np.uint32)), ctx
        assert np.array_equal(reward[i, :ne].view(np.uint32), ref["reward"][i, :ne].view(np.uint32)), ctx
        assert rec[i].move_played == S[i].move_played and rec[i].best_action == S[i].best_action, ctx
        assert np.float32(rec[i].root_value) == np.float32(S[i].root_value), ctx
    sp.close()


@pytest.mark.parametrize("n", [19, 9])
def test_adversarial_policy_rows_equal_the_reference(elf, n):
    """k_mcts_expand's exact std::sort replay (generation-parallel introsort, then one lane per short segment, then the stable final
    sort) on the rows std::sort is sensitive to: median-of-3 killers (the depth limit and __partial_sort fallback), ramps, organ pipes,
    2 / 5 distinct values, all equal (tests/adapters.py adversarial_net) -- the REAL reference stack (oracle/_ref; its CPU restatement
    where that is absent) and the engine search with the same net; every root statistic of every search must be equal."""
    import torch
    from adapters import adversarial_net
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(rollouts_per_thread=96 if n == 19 else 128, max_searches=6 if n == 19 else 16, seed=515, ply_pass_enabled=2, policy_distri_cutoff=4)
    net = adversarial_net(n)
    ref = (RefSelfPlay(n) if RefSelfPlay.available(n) else PortSelfPlay(n)).run(net=net, **cfg)
    S = ref["search"]
    m = len(S)
    assert m == cfg["max_searches"]
    sp = _sp_from_cfg(elf, n, cfg, log_searches=m)
    while sp.stats()["logged"] < m:
        rows = sp.begin_step()
        if rows:
            pi, v = net(sp.s[:rows].cpu().numpy())
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)
    rec, coord, visits, prior, reward = sp.search_log()
    for i in range(m):
        ne = S[i].n_edges
        ctx = "adversarial rows, %dx%d, search %d" % (n, n, i)
        assert rec[i].n_edges == ne, ctx
        assert np.array_equal(coord[i, :ne], ref["coord"][i, :ne]), ctx + ": edge order"
        assert np.array_equal(visits[i, :ne], ref["visits"][i, :ne]), ctx
        assert np.array_equal(prior[i, :ne].view(np.uint32), ref["prior"][i, :ne].view(np.uint32)), ctx
        assert np.array_equal(reward[i, :ne].view(np.uint32), ref["reward"][i, :ne].view(np.uint32)), ctx
        assert rec[i].move_played == S[i].move_played and rec[i].best_action == S[i].best_action, ctx
    sp.close()


# ---- the benchmarked configuration, pinned (VERDICT r1 "next round" item 1) --------------------------------------------------
def _sp_from_cfg(elf, n, cfg, **over):
    kw = dict(board_size=n, device=0, mcts_rollout_per_thread=cfg["rollouts_per_thread"], mcts_rollout_per_batch=cfg["rollouts_per_batch"],
              mcts_puct=cfg["c_puct"], mcts_virtual_loss=cfg["virtual_loss"], mcts_use_prior=bool(cfg["use_prior"]),
              mcts_persistent_tree=bool(cfg["persistent_tree"]), mcts_epsilon=cfg["root_epsilon"], mcts_alpha=cfg["root_alpha"],
              mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]), mcts_root_unexplored_q_zero=bool(cfg["root_unexplored_q_zero"]),
              komi=cfg["komi"], ply_pass_enabled=cfg["ply_pass_enabled"], policy_distri_cutoff=cfg["policy_distri_cutoff"],
              move_cutoff=cfg["move_cutoff"], resign_thres=cfg["resign_thres"], never_resign_prob=cfg["never_resign_prob"],
              seed=cfg["seed"], mcts_threads=cfg["mcts_threads"], num_games=cfg["num_games"], nodes_per_game=4096)
    kw.update(over)
    return elf.SelfPlay(**kw)


def _drive_stub(sp, n, cfg, searches, wait_rows=True):
    import torch
    while sp.stats()["logged"] < searches:
        rows = sp.begin_step(wait_rows=wait_rows)
        k = rows if wait_rows else sp.max_rows          # without the wait every row is evaluated; stale ones are ignored
        if k:
            pi, v = stub_net(n, sp.s[:k].cpu().numpy(), cfg["net_salt"], cfg["net_tie_levels"])
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)


def _per_game(rec, coord, visits, prior, reward, games):
    out = {g: [] for g in range(games)}
    for i, r in enumerate(rec):
        ne = r.n_edges
        out[r.game].append((ne, r.move_played, r.best_action, r.total_visits, np.float32(r.root_value).tobytes(),
                            coord[i, :ne].tobytes(), visits[i, :ne].tobytes(), prior[i, :ne].tobytes(), reward[i, :ne].tobytes()))
    return out


@pytest.mark.parametrize("n,G,roll,per_game", [(19, 8, 64, 4), (9, 16, 48, 10)])
def test_multi_game_context_equals_the_reference_game_by_game(elf, n, G, roll, per_game):
    """A G-game context (the benchmarked shape) against the reference stack running G game threads: game g of the context must
    reproduce, search for search, the reference's game g, which the reference harness seeds with seed + g
    (oracle/ref_selfplay.cc; the per-game seed rule of include/elf_amd.h).  Where the prebuilt reference is absent its CPU
    restatement plays the G games one by one with the same seeds."""
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=G, rollouts_per_thread=roll, seed=9001, net_salt=311, policy_distri_cutoff=6, move_cutoff=30 if n == 9 else -1)
    want = {}
    if RefSelfPlay.available(n):
        # the G reference game threads finish searches at their own pace: run long enough for `per_game` searches of every game
        r = RefSelfPlay(n).run(max_searches=G * per_game * 3, **{k: v for k, v in cfg.items() if k != "max_searches"})
        want = _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)
    else:
        for g in range(G):
            c1 = dict(cfg)
            c1.update(num_games=1, seed=cfg["seed"] + g, max_searches=per_game)
            r = PortSelfPlay(n).run(**c1)
            for x in r["search"]:
                x.game = g
            want.update({g: _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)[g]})
    assert all(len(want[g]) >= per_game for g in range(G)), [len(want[g]) for g in range(G)]
    sp = _sp_from_cfg(elf, n, cfg, log_searches=G * per_game)
    _drive_stub(sp, n, cfg, G * per_game)
    rec, coord, visits, prior, reward = sp.search_log()
    na = n * n + 1
    got = _per_game(rec, coord[:, :na], visits[:, :na], prior[:, :na], reward[:, :na], G)
    for g in range(G):
        assert len(got[g]) == per_game
        for k in range(per_game):
            assert got[g][k] == want[g][k], "game %d search %d" % (g, k)
    sp.close()


def test_step_without_host_wait_equals_step_with_it(elf):
    """elfsp_begin_step(n_rows = NULL): the row count stays on the device and the expansion kernel reads it there.  Same search
    logs as the path that waits for the count."""
    from pyoracle import MCTS_DEFAULTS
    n, G, m = 19, 6, 3
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=G, rollouts_per_thread=96, seed=515, net_salt=99, policy_distri_cutoff=2)
    logs = []
    for wait in (True, False):
        sp = _sp_from_cfg(elf, n, cfg, log_searches=G * m)
        _drive_stub(sp, n, cfg, G * m, wait_rows=wait)
        rec, coord, visits, prior, reward = sp.search_log()
        logs.append(_per_game(rec, coord, visits, prior, reward, G))
        st = sp.stats()
        assert st["rows"] > 0 and st["moves"] == G * m
        sp.close()
    assert logs[0] == logs[1]


@pytest.mark.parametrize("n,T,K,roll", [(9, 2, 8, 64), (19, 2, 16, 96), (9, 4, 4, 32), (9, 2, 64, 256), (19, 4, 32, 128)])
def test_search_threads_equal_their_restatement(elf, n, T, K, roll):
    """TSOptions.num_threads = T > 1: T x num_rollouts_per_thread rollouts per move (tree_search.h:472-476), the T batch_rollouts
    of a round run back to back on the shared tree.  The reference's racing threads are not deterministic; the
    bar is the sequential interleaving restated in oracle/mcts_oracle.cc, bit for bit, plus the visit budget."""
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=1, mcts_threads=T, rollouts_per_thread=roll, rollouts_per_batch=K, batchsize=K, seed=77 + T, net_salt=5,
               max_searches=6, policy_distri_cutoff=3)
    ref = PortSelfPlay(n).run(**cfg)
    m = len(ref["search"])
    assert m == 6
    sp = _sp_from_cfg(elf, n, cfg, log_searches=m)
    assert sp.max_rows == T * K
    _drive_stub(sp, n, cfg, m)
    rec, coord, visits, prior, reward = sp.search_log()
    for i in range(m):
        ne = ref["search"][i].n_edges
        assert rec[i].n_edges == ne
        assert np.array_equal(coord[i, :ne], ref["coord"][i, :ne]) and np.array_equal(visits[i, :ne], ref["visits"][i, :ne]), i
        assert np.array_equal(reward[i, :ne].view(np.uint32), ref["reward"][i, :ne].view(np.uint32)), i
        assert rec[i].move_played == ref["search"][i].move_played
    # budget: every step adds at most T*K visits below the root; the first search loses its first round (the root is the leaf)
    assert 0.5 * T * roll < rec[0].total_visits <= T * roll - T * K
    assert sp.stats()["rollouts"] == m * T * ((roll + K - 1) // K) * K
    sp.close()
    with pytest.raises(Exception):     # the leaf tables hold T x K <= 1024 leaves per step, rejected loudly beyond
        elf.SelfPlay(board_size=9, num_games=1, mcts_rollout_per_batch=64, mcts_threads=32)


def test_backup_order_is_first_occurrence_with_unquantised_values(elf):
    """fp32 reward sums depend on the order in which the leaves of a batch are backed up.  With a value head that is
    NOT quantised (stub salt bit 31) the HIP search must still equal, bit for bit, a CPU search that uses the same documented
    order (first occurrence: oracle/mcts_oracle.cc); the reference itself iterates heap addresses and may differ in the last
    ulps of edges that received two backups from one batch."""
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay
    n = 19
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=1, rollouts_per_thread=512, seed=4, net_salt=0x80000000 | 41, max_searches=4)
    ref = PortSelfPlay(n).run(**cfg)
    sp = _sp_from_cfg(elf, n, cfg, log_searches=4)
    _drive_stub(sp, n, cfg, 4)
    rec, coord, visits, prior, reward = sp.search_log()
    inexact = 0
    for i in range(4):
        ne = ref["search"][i].n_edges
        assert np.array_equal(visits[i, :ne], ref["visits"][i, :ne]), i
        assert np.array_equal(reward[i, :ne].view(np.uint32), ref["reward"][i, :ne].view(np.uint32)), i
        inexact += int(np.sum((reward[i, :ne] * 256.0) != np.round(reward[i, :ne] * 256.0)))
    assert inexact > 0      # the sums really are off the 1/256 grid: the order was exercised
    sp.close()


def test_pipelined_graph_fp16_groups_equal_the_serial_fp32_loop(elf):
    """The headline bench configuration's machinery -- PipelinedSelfPlay with two groups on their own streams, no host wait
    inside a move, the net call replayed as a HIP graph, fp16 channels_last feature rows -- against the plain serial loop with
    fp32 NCHW rows and an eager net: identical search logs, group by group and game by game."""
    import torch
    from adapters import HashNet
    from elf_amd.pipeline import PipelinedSelfPlay
    n, Gg, m, roll = 19, 4, 3, 128
    kw = dict(board_size=n, num_games=Gg, mcts_rollout_per_thread=roll, mcts_rollout_per_batch=16, mcts_puct=1.5, mcts_virtual_loss=1,
              mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, policy_distri_cutoff=30, nodes_per_game=2048,
              log_searches=Gg * m)
    net = HashNet(n, torch.device("cuda", 0))
    steps = m * (roll // 16)
    # serial reference loops, one per group, seeded as the pipeline seeds its groups
    serial = []
    for i in range(2):
        sp = elf.SelfPlay(seed=1234, game_idx_base=i * Gg, feature_format="f32_nchw", **kw)
        for _ in range(steps):
            rows = sp.begin_step()
            pi, v = net(sp.s[:rows])
            sp.end_step(pi, v)
        rec, coord, visits, prior, reward = sp.search_log()
        serial.append(_per_game(rec, coord, visits, prior, reward, Gg))
        assert len(rec) == Gg * m
        sp.close()
    pipe = PipelinedSelfPlay(groups=2, seed=1234, wait_rows=False, feature_format="f16_nhwc", **kw)
    graphs, outs = {}, {}
    side = torch.cuda.Stream()
    for g in pipe.groups:                      # one captured graph per group, reading that group's own "s" tensor
        with torch.cuda.stream(side):
            for _ in range(2):
                net(g.s)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            outs[g.s.data_ptr()] = net(g.s)
        graphs[g.s.data_ptr()] = gr

    def net_fn(s, rows):
        graphs[s.data_ptr()].replay()
        return outs[s.data_ptr()]

    for _ in range(steps):
        pipe.step(net_fn)
    pipe.synchronize()
    for i, g in enumerate(pipe.groups):
        rec, coord, visits, prior, reward = g.search_log()
        assert len(rec) == Gg * m
        assert _per_game(rec, coord, visits, prior, reward, Gg) == serial[i], "group %d" % i
    pipe.close()


# ---- SURVEY.md 8(d) config 3 as the survey defines it: the real net, the real reference, visit counts ------
def test_config3_real_net_against_the_real_reference_stack(elf):
    """Model_PolicyValue 20 x 256, torch.manual_seed(0), fp32, eval -- on this GPU -- drives BOTH the real reference stack
    (oracle/_ref: batcher + GoGameSelfPlay + MCTSGoAI + tree_search/*.h, through its batch interface) and the HIP engine;
    num_games = 1, mcts_threads = 1, seed 1234, bs 16, puct 1.5, vloss 1, Dirichlet 0.25 / 0.03, persistent tree: moves 1..8 at 512
    rollouts and one 8192-rollout search.
    (a) Against the reference built with the CANONICAL backup order (oracle/Makefile, libelfsp19_h2.so: three lines of a build-time
        copy of tree_search.h make batch_rollouts walk the unique leaves of a batch in first-occurrence order instead of the
        iteration order of a map keyed by heap addresses -- SURVEY.md hazard H2): EVERY statistic bit for bit, reward sums included.
        0 ulps: the engine's order is that definition, not a coincidence.
    (b) Against the STOCK reference (heap-address order, :216,245): root edge order, priors, visit counts, most-visited action, move
        played and root value bit for bit; the reward sum of an edge that received two or more leaves of one batch may differ in its
        last bits (<= 2 ulps admitted; 1 ulp is the most ever seen: 4 of 208 searches, profiles/r03a_, r03m_, r04w_).
    The net is made a pure function of the feature row (fixed evaluation batches, memoised by the row's digest:
    tests/real_net_parity.py) so that both engines see identical (pi, V) for identical positions."""
    import real_net_parity as rp
    from pyoracle import RefSelfPlay
    if not RefSelfPlay.available(19):
        pytest.skip("oracle/_ref/libelfsp19.so (the reference compiled in place) is not present: make -C oracle ref")
    memo = rp.make_memo_net(19, 20, 256)
    for rollouts, moves, seed in ((512, 8, 1234), (8192, 1, 1234)):
        cfg = rp.search_cfg(rollouts_per_thread=rollouts, seed=seed)
        have_canon = RefSelfPlay.available(19, canonical_backup=True)
        canon = rp.run_reference(memo, 19, cfg, 1, moves, canonical_backup=True) if have_canon else None
        ref = rp.run_reference(memo, 19, cfg, 1, moves)
        got, engine_only_rows = rp.run_engine(memo, 19, cfg, 1, moves)
        assert engine_only_rows == 0          # the engine asked the net for positions the reference asked for, nothing else
        if have_canon:
            res = rp.compare(canon, got, 1, moves)
            assert res["searches_compared"] == moves and res["bit_equal"] == moves, res      # 0 ulps, no tolerance
            assert res["max_reward_ulps"] == 0 and res["reward_ulps_only"] == 0 and res["decision_diverged"] == 0, res
        res = rp.compare(ref, got, 1, moves)
        assert res["searches_compared"] == moves
        assert res["decision_diverged"] == 0, res
        assert res["bit_equal"] + res["reward_ulps_only"] == moves, res
        assert res["max_reward_ulps"] <= 2, res   # hazard H2 against the stock build: a report, bounded
        for k in range(moves):
            assert ref[0][k]["total_visits"] == got[0][k]["total_visits"]
        # 8192 rollouts at bs 16: the all-at-root first batch adds no visit
        if rollouts == 8192:
            assert ref[0][0]["total_visits"] == 8176


def test_real_net_from_a_dense_sgf_position_against_the_real_reference_stack(elf):
    """The same 20 x 256 fp32 net, the search started from a DENSE position: ladder-suite game 406844.sgf preloaded to ply 150
    (GameOptions.preload_sgf, game_selfplay.cc:202-219; ~220 legal moves, 50+ groups), ply_pass_enabled = 100 so that the pass edge,
    Tromp-Taylor leaves and remove_pass_if_dangerous (go/mcts/mcts.h:185-207,232-242) occur with un-quantised values; 4 searches at 512
    rollouts (the played moves are the SGF's, :392-405).  Against the canonical-backup build: 0 ulps on every statistic; against the
    stock build: the bounded report."""
    import real_net_parity as rp
    from pyoracle import RefSelfPlay
    if not RefSelfPlay.available(19):
        pytest.skip("oracle/_ref/libelfsp19.so (the reference compiled in place) is not present: make -C oracle ref")
    memo = rp.make_memo_net(19, 20, 256)
    pre = (np.load(os.path.join(GOLDEN, "sgf_406844.npz"))["moves"].astype(np.uint16), 150)
    moves = 4
    cfg = rp.search_cfg(rollouts_per_thread=512, seed=4321, ply_pass_enabled=100)
    got, engine_only_rows = rp.run_engine(memo, 19, cfg, 1, moves, preload=pre)
    assert got[0][0]["n_edges"] < 240
    if RefSelfPlay.available(19, canonical_backup=True):
        canon = rp.run_reference(memo, 19, cfg, 1, moves, canonical_backup=True, preload=pre)
        res = rp.compare(canon, got, 1, moves)
        assert res["searches_compared"] == moves and res["bit_equal"] == moves and res["max_reward_ulps"] == 0, res
    ref = rp.run_reference(memo, 19, cfg, 1, moves, preload=pre)
    res = rp.compare(ref, got, 1, moves)
    assert res["searches_compared"] == moves and res["decision_diverged"] == 0 and res["max_reward_ulps"] <= 2, res


def test_search_threads_with_the_real_net_against_the_turnstile_reference(elf):
    """mcts_threads = 2 with the benchmark's own 20 x 256 fp32 net: the real reference under the turnstile schedule AND the canonical
    backup order (oracle/Makefile, libelfsp19_tsh2.so: both build-time patches) against the engine -- every statistic of 4 searches at
    2 x 128 rollouts bit for bit, reward sums included (un-quantised values: the order of every fp32 sum matters here).
    Measured on more searches: profiles/history/r05x_config3_real_net_parity_canon_T2.json / _T4.json (24 + 8 searches, all bit-equal)."""
    import real_net_parity as rp
    from pyoracle import RefSelfPlay
    if not RefSelfPlay.available(19, turnstile=True, canonical_backup=True):
        pytest.skip("oracle/_ref/libelfsp19_tsh2.so is not present: make -C oracle ref")
    memo = rp.make_memo_net(19, 20, 256)
    moves = 4
    cfg = rp.search_cfg(rollouts_per_thread=128, seed=99, mcts_threads=2)
    ref = rp.run_reference(memo, 19, cfg, 1, moves, canonical_backup=True, turnstile=True)
    got, engine_only_rows = rp.run_engine(memo, 19, cfg, 1, moves)
    res = rp.compare(ref, got, 1, moves)
    assert engine_only_rows == 0
    assert res["searches_compared"] == moves and res["bit_equal"] == moves and res["max_reward_ulps"] == 0, res


# ---- round 3: requests, idle games, evaluation games out of step, uniform_random ------------------------------------------------
def _live_per_game(n, cfg, G, per_game):
    """the reference stack running G game threads (game g seeded seed + g) -> {g: [search tuples]}"""
    from pyoracle import RefSelfPlay
    r = RefSelfPlay(n).run(**dict(cfg, max_searches=G * per_game * 3))
    return _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G), r


def test_evaluation_games_out_of_step_equal_the_reference_game_by_game(elf):
    """Four evaluation games in one context, the two AIs with different rollout budgets and batch sizes (Black 48 rollouts at
    bs 16 = 3 steps per move, White 40 rollouts at bs 8 = 5 steps), games ended by resignation at different plies (52..58) or by
    the cutoff: from the first game end on the games' searches are out of step -- different AIs searching in the same step,
    moves and restarts at different steps -- and every game must still reproduce the reference's game thread (live, oracle/_ref)."""
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    n, G, per_game = 9, 4, 72
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=G, rollouts_per_thread=48, seed=7001, net_salt=61, white_net_salt=62, black_ver=3, white_ver=4,
               white_rollouts_per_thread=40, white_rollouts_per_batch=8, white_puct=1.1, policy_distri_cutoff=5, move_cutoff=64,
               resign_thres=0.9)
    if RefSelfPlay.available(n):
        r = RefSelfPlay(n).run(**dict(cfg, max_searches=G * per_game * 2))
        want = _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)
    else:       # a fresh clone: the CPU restatement (pinned on the reference's evaluation-game fixtures) plays the games one by one
        want = {}
        for g in range(G):
            r = PortSelfPlay(n).run(**dict(cfg, num_games=1, seed=cfg["seed"] + g, max_searches=per_game))
            for x in r["search"]:
                x.game = g
            want[g] = _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)[g]
    assert all(len(want[g]) >= per_game for g in range(G)), [len(want[g]) for g in range(G)]
    sp = sp_from_fixture_cfg(elf, n, cfg, log_searches=G * per_game * 2, nodes_per_game=2048)
    seen = {}
    L = elf.lib()

    def out_of_step(sp, rows_total):
        a = tuple(L.elfsp_game_actor(sp._h, g) for g in range(G))
        seen[a] = seen.get(a, 0) + 1

    def enough(sp):
        if sp.progress()["searches"] < G * per_game:
            return False
        rec = sp.search_log()[0]
        return all(sum(1 for x in rec if x.game == g) >= per_game for g in range(G))

    drive_stub(sp, n, cfg, enough, out_of_step)
    rec, coord, visits, prior, reward = sp.search_log()
    na = n * n + 1
    got = _per_game(rec, coord[:, :na], visits[:, :na], prior[:, :na], reward[:, :na], G)
    for g in range(G):
        for k in range(per_game):
            assert got[g][k] == want[g][k], "game %d search %d" % (g, k)
    assert any(len(set(a)) > 1 for a in seen), seen      # at some step the games were searching with different AIs
    assert sp.games_finished() >= G                      # every game ended at least once (both trees reset, next game started)
    sp.close()


def test_idle_game_threads_equal_the_reference(elf):
    """num_game_thread_used = 2 of 3 games (DispatcherCallback::OnFirstSend): game 2 waits, games 0 and 1 equal the reference's."""
    from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay
    n, G, per_game = 9, 3, 6
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=G, rollouts_per_thread=48, seed=8100, net_salt=63, thread_used=2, policy_distri_cutoff=3)
    # games 0 and 1 of the reference do not depend on the idle thread: they are the reference's two-game context with the same
    # seeds.  (The reference itself cannot be run to completion with an idle game thread: that thread blocks in waitMail for
    # ever and Context::stop never joins it.)
    if RefSelfPlay.available(n):
        r = RefSelfPlay(n).run(**dict(cfg, num_games=2, thread_used=0, max_searches=2 * per_game * 2))
        want = _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)
    else:
        want = {2: []}
        for g in range(2):
            r = PortSelfPlay(n).run(**dict(cfg, num_games=1, thread_used=0, seed=cfg["seed"] + g, max_searches=per_game))
            for x in r["search"]:
                x.game = g
            want[g] = _per_game(r["search"], r["coord"], r["visits"], r["prior"], r["reward"], G)[g]
    assert len(want[2]) == 0 and min(len(want[0]), len(want[1])) >= per_game
    sp = sp_from_fixture_cfg(elf, n, cfg, log_searches=64, nodes_per_game=2048)
    drive_stub(sp, n, cfg, lambda sp: sp.stats()["logged"] >= 2 * per_game)
    rec, coord, visits, prior, reward = sp.search_log()
    na = n * n + 1
    got = _per_game(rec, coord[:, :na], visits[:, :na], prior[:, :na], reward[:, :na], G)
    assert len(got[2]) == 0 and sp.progress()["waiting"] == 1
    for g in range(2):
        for k in range(per_game):
            assert got[g][k] == want[g][k], "game %d search %d" % (g, k)
    # a request that uses all three threads starts the waiting game from the empty board (is_prev_waiting -> restart)
    sp.set_request(0, -1, num_game_thread_used=3)
    drive_stub(sp, n, cfg, lambda sp: sum(1 for x in sp.search_log()[0] if x.game == 2) >= 2)
    assert sp.progress()["waiting"] == 0
    first = [x for x in sp.search_log()[0] if x.game == 2][0]
    assert first.total_visits == 48 - 16           # a first search of a fresh game: the all-at-root first batch adds no visit
    sp.close()


def test_pick_method_uniform_random(elf):
    """TSOptions.pick_method = uniform_random (tree_search.h:514-517, addActions :243-279): the move is the random_idx-th root edge
    in iteration order, random_idx = rng() % edges, every edge scores 1 -- so max_score is 1 and, with the opening temperature on,
    the sampled move comes from the uniform policy.  Several games here (in the reference its game threads race for the one
    process-wide generator, so only the one-game fixture above can be equalled): the structure, and determinism under a fixed
    GameOptions.seed."""
    import torch
    runs = []
    for _ in range(2):
        sp = elf.SelfPlay(board_size=9, num_games=4, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, seed=99, log_searches=40,
                          mcts_pick_method="uniform_random", policy_distri_cutoff=0, nodes_per_game=1024, move_cutoff=20)
        while sp.stats()["logged"] < 40:
            rows = sp.begin_step()
            pi, v = stub_net(9, sp.s[:rows].cpu().numpy(), 5, 0)
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        rec, coord, visits, prior, reward = sp.search_log()
        for i, r in enumerate(rec):
            assert r.max_score == 1.0 and r.best_action in coord[i, :r.n_edges] and r.move_played == r.best_action
        # not the most visited edge every time (that would be most_visited in disguise)
        assert sum(1 for i, r in enumerate(rec) if r.best_action != coord[i, int(np.argmax(visits[i, :r.n_edges]))]) > 10
        runs.append([(r.game, r.move_played) for r in rec])
        sp.close()
    assert runs[0] == runs[1]
    with pytest.raises(ValueError):
        elf.SelfPlay(board_size=9, num_games=1, mcts_pick_method="softmax")


def test_async_request_changes_the_model_without_restarting_the_games(elf):
    """ClientCtrl.async (GoGameSelfPlay::OnReceive :251-262 -> setAsync :150-156): a request with new versions does NOT restart the
    games; the AIs stop checking reply versions, the game's Record lists every model it was played with (GoStateExt::using_models_),
    and a game_start is due (UPDATE_MODEL_ASYNC).  Engine-side semantics (the reference harness sends one request per run)."""
    import json
    import torch
    n = 9
    sp = elf.SelfPlay(board_size=n, num_games=2, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, seed=5, move_cutoff=14,
                      keep_records=4, nodes_per_game=1024, model_ver=3)

    def step(ver):
        rows = sp.begin_step()
        pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), 9, 0)
        sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device),
                    torch.full((rows,), ver, dtype=torch.int64, device=sp.device))

    L = elf.lib()
    import ctypes as C
    bv, wv = C.c_int64(-5), C.c_int64(-5)
    while sp.progress()["searches"] < 2 * 3:
        step(3)
    assert L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv)) == 1 and (bv.value, wv.value) == (3, -1)
    sp.set_request(8, -1, async_=True)
    plies_before = sp.board_engine().info_host()["ply"].copy()
    while sp.progress()["searches"] < 2 * 6:      # received at the sixth act of each game; until then version 3 is still required
        step(3)
    step(12345)                                    # now any version is accepted ...
    assert L.elfsp_take_game_starts(sp._h, C.byref(bv), C.byref(wv)) == 1 and (bv.value, wv.value) == (8, -1)
    assert (sp.board_engine().info_host()["ply"] >= plies_before).all()    # ... and the games went on, not back to the empty board
    recs = []
    while len(recs) < 2:
        step(777)
        recs += sp.pop_records()
    j = json.loads(recs[0])
    assert j["result"]["using_models"] == [3, 8] and j["request"]["vers"]["black_ver"] == 8 and j["request"]["client_ctrl"]["async"] is True
    assert j["result"]["num_move"] == 13           # one uninterrupted game to the cutoff
    sp.close()


def test_request_options_that_cannot_be_applied_yet_are_counted_not_dropped_silently(elf):
    """The tree pools belong to the whole context and are rebuilt for a request's TSOptions only when no game is mid-play.  A request
    that RESTARTS some games (here: the one that was waiting, num_game_thread_used 1 -> 2) with other search options while another
    game plays ON (an async request does not restart a playing game, setAsync :150-156) cannot be honoured for the restarted game:
    it searches with the context's options.  That case is counted (elfsp_ts_requests_deferred, elfsp_ts_games_deferred) and logged, not
    silent; the games keep running and their records carry the search options that were actually used."""
    import torch
    from elf_amd.client import TsOptions
    n = 9
    sp = elf.SelfPlay(board_size=n, num_games=2, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16, seed=5, move_cutoff=10,
                      keep_records=4, nodes_per_game=1024, model_ver=3)
    L = elf.lib()
    assert L.elfsp_ts_requests_deferred(sp._h) == 0

    def step():
        rows = sp.begin_step()
        if rows:
            pi, v = stub_net(n, sp.s[:rows].cpu().numpy(), 9, 0)
            sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
        else:
            sp.end_step(None, None)

    sp.set_request(3, -1, num_game_thread_used=1)          # both games restart at the barrier; then game 1 waits, game 0 plays
    while sp.progress()["searches"] < 7 or sp.progress()["waiting"] != 1:
        step()
    assert L.elfsp_ts_requests_deferred(sp._h) == 0
    ts = TsOptions(0, 1, 48, 16, 0, 0, 1, 0, 0, 0.25, 0.03, 1, 1, 0, 0, 1.5, b"")     # 48 rollouts per move instead of 32
    sp.set_request(4, -1, async_=True, num_game_thread_used=2, mcts_opt=ts)
    before = sp.progress()["searches"]
    while sp.progress()["waiting"] != 0 or sp.progress()["searches"] < before + 12:
        step()
    assert L.elfsp_ts_requests_deferred(sp._h) == 1          # game 1 restarted under the request while game 0 played on
    assert sp.stats()["steps_per_move"] == 2                 # the context's options still rule: 32 rollouts = 2 steps of 16
    assert L.elfsp_ts_games_deferred(sp._h) == 2             # both games now hold a request whose options the pools do not have
    # the records say what was USED (32 rollouts per thread), not what the request asked for (48)
    import json
    late = []
    for _ in range(400):
        late += [r for r in map(json.loads, sp.pop_records()) if r["request"]["vers"]["black_ver"] == 4]
        if late:
            break
        step()
    assert late and all(r["request"]["vers"]["mcts_opt"]["num_rollouts_per_thread"] == 32 for r in late), late[:1]
    sp.close()


def test_pipelined_groups_with_requests_and_evaluation_games(elf):
    """PipelinedSelfPlay.step2: game groups pipelined against the net while requests arrive -- self-play, then an evaluation request
    with its own search options (second AI, noise off), then self-play with the next model.  Every group restarts at its own
    barrier; replies carry the versions of the group's last "game_start"; the records name the models they were played with."""
    import json
    import torch
    from elf_amd.client import TsOptions
    from elf_amd.selfplay import SpRequest
    n = 9
    from elf_amd.pipeline import PipelinedSelfPlay
    pl = PipelinedSelfPlay(groups=2, seed=31, board_size=n, num_games=3, mcts_rollout_per_thread=32, mcts_rollout_per_batch=16,
                           move_cutoff=10, keep_records=32, nodes_per_game=1024, model_ver=1)
    seen = set()

    def make(salt_of):
        def fn(s, rows, ver):
            seen.add(ver)
            pi, v = stub_net(n, s[:rows].cpu().numpy(), salt_of(ver), 0)
            return torch.from_numpy(pi).to(pl.device), torch.from_numpy(v).to(pl.device)
        return fn

    fns = (make(lambda ver: 100 + ver), make(lambda ver: 200 + ver))
    out = elf.ClientRecords("pipelined")
    ts = TsOptions(0, 1, 32, 16, 0, 0, 1, 0, 0, 0.25, 0.03, 1, 1, 0, 0, 1.5, b"")
    ts_eval = TsOptions.from_buffer_copy(ts)
    ts_eval.root_epsilon = ts_eval.root_alpha = 0.0
    ts_eval.num_rollouts_per_thread = 48                      # other search options than the context was created with
    plan = {60: (SpRequest(2, 1, 0.0, 0.0, 0.0, -1, 0, 0, 2), ts_eval), 160: (SpRequest(2, -1, 0.0, 0.0, 0.0, -1, 0, 0, 1), ts)}
    recs = []
    for step in range(260):
        if step in plan:
            pl.send_request(*plan[step])
        assert pl.step2(fns) >= 0
        out.update_from(pl)
        if step % 50 == 49:
            recs += json.loads(out.dump_and_clear()).get("records", [])
    pl.synchronize()
    kinds = {(r["request"]["vers"]["black_ver"], r["request"]["vers"]["white_ver"]) for r in recs}
    assert kinds == {(1, -1), (2, 1), (2, -1)}, kinds
    ev = [r for r in recs if r["request"]["vers"]["white_ver"] == 1]
    assert all(r["request"]["vers"]["mcts_opt"]["num_rollouts_per_thread"] == 48 and r["request"]["vers"]["mcts_opt"]["root_epsilon"] == 0.0
               and r["result"]["using_models"] == [1, 2] and r["request"]["client_ctrl"]["client_type"] == 2 for r in ev)
    assert seen == {1, 2} and {r["thread_id"] for r in recs} == set(range(6))      # both groups' games (job-wide thread ids)
    assert pl.versions == [(2, -1), (2, -1)]
    pl.close()
    out.close()
