"""CPU: the MCTS oracle side.  (a) stub net properties the parity argument rests on; (b) the committed MCTS
golden fixtures are internally consistent with the reference's own invariants (tree_search.h:200-262:
visits sum, one visit per unique leaf); (c) where oracle/_ref is present, the REAL reference self-play stack
reproduces the committed fixture (the generating script is oracle/gen_golden_mcts.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import MCTS_DEFAULTS, PortSelfPlay, RefSelfPlay, stub_net

CASES = ["mcts_19_r8192", "mcts_19_r256_dir", "mcts_19_r256_ties", "mcts_19_r512_client", "mcts_19_r128_fresh", "mcts_9_r512",
         "mcts_9_r64_ties", "mcts_19_r128_vl0", "mcts_19_r128_noprior", "mcts_9_r128_rootq0", "mcts_9_r96_bs4", "mcts_9_r128_bs64"]
# round 3: evaluation games (two AIs), strongest_prior, policy-only play, more than 64 rollouts per batch
CASES_R3 = ["mcts_9_eval_two_ai", "mcts_19_eval_swap", "mcts_9_pick_prior", "mcts_9_policy_only_white", "mcts_9_policy_only_eval",
            "mcts_9_r256_bs128", "mcts_19_r512_bs256", "mcts_9_pick_uniform", "mcts_9_r1024_bs512"]


# round 5: mcts_threads > 1, fixtures from the turnstile build of the REAL reference (oracle/Makefile: libelfsp*_ts.so; four
# elf_ts_hook() calls inserted into a build-time copy of batch_rollouts force one thread order per round)
CASES_T = ["mcts_9_T2_r128", "mcts_9_T4_r256", "mcts_19_T2_r512", "mcts_19_T8_client", "mcts_9_T3_eval_two_ai"]
# round 6: searches from dense 19x19 positions (GameOptions.preload_sgf follows ladder-suite games to plies 60..195: n_edges 183..303,
# pass edges, Tromp-Taylor leaves inside the tree; one turnstile T = 2 case, one 8192-rollout case)
CASES_SGF = ["mcts_19_sgf_p60", "mcts_19_sgf_p120", "mcts_19_sgf_p180", "mcts_19_sgf_p195", "mcts_19_sgf_b_p150", "mcts_19_sgf_c_p100",
             "mcts_19_sgf_T2_p140", "mcts_19_sgf_r8192_p160"]


@pytest.mark.parametrize("n", [19, 9])
def test_stub_net_is_a_quantised_distribution(built, n):
    rng = np.random.default_rng(0)
    s = (rng.random((5, 18, n, n)) < 0.3).astype(np.float32)
    pi, v = stub_net(n, s)
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5) and (pi > 0).all()
    assert np.array_equal(v * 256, np.round(v * 256)) and (np.abs(v) <= 1).all()   # multiples of 1/256: exact fp32 sums
    pi2, v2 = stub_net(n, s)
    assert np.array_equal(pi, pi2) and np.array_equal(v, v2)
    pit, _ = stub_net(n, s, tie_levels=3)
    assert len(np.unique(pit[0])) <= 3   # forced equal priors


@pytest.mark.parametrize("name", CASES + CASES_T + CASES_SGF)
def test_fixture_invariants(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    for i in range(len(g["move_played"])):
        ne = int(g["n_edges"][i])
        c = g["coord"][i, :ne]
        assert len(set(c.tolist())) == ne and (g["coord"][i, ne:] == -1).all()
        assert g["visits"][i, :ne].sum() == g["total_visits"][i]
        assert abs(g["prior"][i, :ne].sum() - 1.0) < 1e-3
        # most_visited with strict '>' in iteration order (tree_search_base.h:276-281)
        assert g["best_action"][i] == c[int(np.argmax(g["visits"][i, :ne]))]
        if cfg["persistent_tree"] == 0:   # a fresh tree: the all-at-root first batch adds no visit
            per_batch = int(cfg["rollouts_per_batch"])
            assert g["total_visits"][i] <= (cfg["rollouts_per_thread"] - per_batch) * max(1, int(cfg["mcts_threads"]))
        assert ne <= n * n + 1


@pytest.mark.parametrize("name", ["mcts_19_r128_fresh", "mcts_9_r64_ties", "mcts_9_eval_two_ai", "mcts_9_policy_only_white"])
def test_reference_reproduces_fixture(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["board_size"])
    if not RefSelfPlay.available(n):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    kw = {k: (float(np.float32(v)) if isinstance(MCTS_DEFAULTS[k], float) else int(v)) for k, v in cfg.items()}
    r = RefSelfPlay(n).run(**kw)
    assert np.array_equal(r["coord"].astype(np.int16), g["coord"])
    assert np.array_equal(r["visits"], g["visits"])
    assert np.array_equal(r["prior"].view(np.uint32), g["prior"].view(np.uint32))
    assert np.array_equal(r["reward"].view(np.uint32), g["reward"].view(np.uint32))
    assert [s.move_played for s in r["search"]] == g["move_played"].tolist()


@pytest.mark.parametrize("name", ["mcts_9_T2_r128", "mcts_19_T8_client"])
def test_turnstile_reference_reproduces_fixture_and_the_stock_build_has_no_hooks(name):
    """The turnstile build of the reference is deterministic at mcts_threads > 1 and reproduces the committed fixture; the stock
    build (what every other fixture comes from) carries no hook."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["board_size"])
    if not RefSelfPlay.available(n, turnstile=True):
        pytest.skip("oracle/_ref/libelfsp*_ts.so not built (no /root/reference here)")
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    kw = {k: (float(np.float32(v)) if isinstance(MCTS_DEFAULTS[k], float) else int(v)) for k, v in cfg.items()}
    kw["max_searches"] = min(int(kw["max_searches"]), 12)
    m = kw["max_searches"]
    R = RefSelfPlay(n, turnstile=True)
    assert R.L.refsp_has_turnstile() == 1 and RefSelfPlay(n).L.refsp_has_turnstile() == 0
    r = R.run(**kw)
    assert np.array_equal(r["coord"].astype(np.int16), g["coord"][:m])
    assert np.array_equal(r["visits"], g["visits"][:m])
    assert np.array_equal(r["prior"].view(np.uint32), g["prior"][:m].view(np.uint32))
    assert np.array_equal(r["reward"].view(np.uint32), g["reward"][:m].view(np.uint32))
    assert [s.move_played for s in r["search"]] == g["move_played"][:m].tolist()


@pytest.mark.parametrize("name", ["mcts_19_r128_fresh", "mcts_9_r64_ties"])
def test_canonical_backup_build_reproduces_the_stock_fixtures(name):
    """The H2 build of the reference (first-occurrence backup order, oracle/Makefile) gives what the stock build gave wherever the
    order cannot matter: the stub net's values are multiples of 1/256, so the fp32 reward sums are exact in any order."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["board_size"])
    if not RefSelfPlay.available(n, canonical_backup=True):
        pytest.skip("oracle/_ref/libelfsp*_h2.so not built (no /root/reference here)")
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    kw = {k: (float(np.float32(v)) if isinstance(MCTS_DEFAULTS[k], float) else int(v)) for k, v in cfg.items()}
    R = RefSelfPlay(n, canonical_backup=True)
    assert R.L.refsp_has_canonical_backup() == 1 and RefSelfPlay(n).L.refsp_has_canonical_backup() == 0
    r = R.run(**kw)
    assert np.array_equal(r["visits"], g["visits"]) and np.array_equal(r["reward"].view(np.uint32), g["reward"].view(np.uint32))
    assert [s.move_played for s in r["search"]] == g["move_played"].tolist()


@pytest.mark.parametrize("seed", [11, 12, 14, 16])
def test_turnstile_reference_equals_restatement_on_random_configurations(built, seed):
    """Differential check of the mcts_threads > 1 semantics beyond the five committed fixtures: random search settings (threads,
    rollouts per batch / per thread, virtual loss, puct, Dirichlet on / off, persistent tree, pass rule, prior ties), the REAL
    reference under the turnstile schedule against the CPU restatement the engine is tested against live on the GPU
    (tests/test_gpu_mcts.py::test_search_threads_equal_their_restatement)."""
    n = 9
    if not RefSelfPlay.available(n, turnstile=True):
        pytest.skip("oracle/_ref/libelfsp9_ts.so not built (no /root/reference here)")
    rng = np.random.default_rng(seed)
    T = int(rng.choice([2, 3, 4, 8]))
    K = int(rng.choice([1, 2, 4, 8, 16]))
    kw = dict(MCTS_DEFAULTS)
    kw.update(num_games=1, mcts_threads=T, rollouts_per_batch=K, batchsize=max(K, 8), rollouts_per_thread=K * int(rng.integers(2, 9)),
              virtual_loss=int(rng.choice([0, 1, 5])), c_puct=float(np.float32(rng.choice([0.5, 0.85, 1.5]))),
              root_epsilon=float(np.float32(rng.choice([0.0, 0.25]))), root_alpha=float(np.float32(0.03)),
              persistent_tree=int(rng.integers(0, 2)), ply_pass_enabled=int(rng.choice([0, 3, 1000])), net_salt=int(rng.integers(1, 1000)),
              net_tie_levels=int(rng.choice([0, 0, 4])), policy_distri_cutoff=int(rng.choice([0, 6])), seed=int(rng.integers(1, 10000)),
              max_searches=10)
    r = RefSelfPlay(n, turnstile=True).run(**kw)
    p = PortSelfPlay(n).run(**kw)
    assert len(r["search"]) == len(p["search"]) == 10, kw
    for k in ("coord", "visits"):
        assert np.array_equal(r[k], p[k]), (k, kw)
    for k in ("prior", "reward"):
        assert np.array_equal(r[k].view(np.uint32), p[k].view(np.uint32)), (k, kw)
    assert [s.move_played for s in r["search"]] == [s.move_played for s in p["search"]], kw


RECORD_RUNS = ["records_9_cutoff", "records_9_resign", "records_9_twopass", "records_9_neverresign", "records_9_preload", "records_19_resign", "records_19_sgf_preload",
               "records_19_cutoff", "records_9_eval", "records_9_eval_swap_resign", "records_9_req2_restart", "records_9_req2_async",
               "records_9_cheat_selfplay", "records_9_cheat_eval", "records_9_cheat_eval_swap", "records_9_req2_ts", "records_9_req2_eval", "records_9_sgf", "records_9_sgf_policy_only"]


@pytest.mark.parametrize("name", CASES + CASES_R3 + CASES_T + CASES_SGF + RECORD_RUNS)
def test_restatement_matches_reference_fixture(built, name):
    """oracle/mcts_oracle.cc (the CPU restatement of MCTSActor, the tree search and the self-play loop over go_oracle.c) replays
    the fixture's configuration and must give what the REAL reference gave: every search's root edges in iteration order,
    priors, visit counts, accumulated rewards, most-visited action, move played and root value, bit for bit -- across Dirichlet
    noise, move sampling, prior ties, pass rules, resignation, never-resign draws, game ends and restarts."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    fl = ("c_puct", "root_epsilon", "root_alpha", "komi", "resign_thres", "never_resign_prob", "white_puct", "net_value", "req2_c_puct",
          "req2_root_epsilon", "req2_root_alpha")
    kw = {k: (float(np.float32(v)) if k in fl else int(v)) for k, v in cfg.items()}
    m = len(g["move_played"])
    if name in ("mcts_19_r8192", "mcts_19_sgf_r8192_p160"):
        m = 1                       # 8192 rollouts per search: one search keeps the CPU suite short
    kw["max_searches"] = m
    P = PortSelfPlay(n)
    if "fixed_time" in g.files:     # uniform_random: the value time(NULL) gave the reference's pick generator in the generating process
        P.set_time(int(g["fixed_time"]))
    if "preload_moves" in g.files:
        P.set_preload(g["preload_moves"], int(g["preload_move_to"]))
    try:
        r = P.run(**kw)
    finally:
        P.set_preload([], -1)
    assert len(r["search"]) == m
    for i in range(m):
        ne = int(g["n_edges"][i])
        S = r["search"][i]
        ctx = "%s search %d" % (name, i)
        assert S.n_edges == ne, ctx
        assert np.array_equal(r["coord"][i, :ne], g["coord"][i, :ne].astype(np.int32)), ctx
        assert np.array_equal(r["visits"][i, :ne], g["visits"][i, :ne]), ctx
        assert np.array_equal(r["prior"][i, :ne].view(np.uint32), g["prior"][i, :ne].view(np.uint32)), ctx
        assert np.array_equal(r["reward"][i, :ne].view(np.uint32), g["reward"][i, :ne].view(np.uint32)), ctx
        assert S.move_played == int(g["move_played"][i]) and S.best_action == int(g["best_action"][i]), ctx
        assert np.float32(S.root_value) == g["root_value"][i], ctx


def test_restatement_equals_reference_on_adversarial_policy_rows(built):
    """std::sort's order of equal priors (go/mcts/mcts.h:292-297) on the inputs it is sensitive to -- median-of-3 killers (the
    __partial_sort fallback of __introsort_loop), ramps, organ pipes, few distinct values, all equal (tests/adapters.py
    adversarial_net) -- through the whole search: the REAL reference stack against the CPU restatement, 19x19, 6 searches."""
    from adapters import adversarial_net
    n = 19
    if not RefSelfPlay.available(n):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    kw = dict(MCTS_DEFAULTS)
    kw.update(rollouts_per_thread=96, max_searches=6, seed=515, ply_pass_enabled=2, policy_distri_cutoff=4)
    net = adversarial_net(n)
    r = RefSelfPlay(n).run(net=net, **kw)
    p = PortSelfPlay(n).run(net=net, **kw)
    assert len(r["search"]) == len(p["search"]) == 6
    for k in ("coord", "visits"):
        assert np.array_equal(r[k], p[k]), k
    for k in ("prior", "reward"):
        assert np.array_equal(r[k].view(np.uint32), p[k].view(np.uint32)), k
    assert [s.move_played for s in r["search"]] == [s.move_played for s in p["search"]]
