"""Generates tests/golden/mcts_*.npz from the REAL reference self-play stack (oracle/_ref/libelfsp*.so =
elf::Context + GoGameSelfPlay + MCTSGoAI/MCTSActor + tree_search/*.h compiled in place from /root/reference),
fed by the deterministic stub net of oracle/stub_net.h.  Run in the build container only:
    python oracle/gen_golden_mcts.py
Per search (= per move) the fixture holds what GameNotifierBase::OnMCTSResult reports: the move played, the
most-visited action, root value, and the root edges IN THE REFERENCE'S ITERATION ORDER with prior, visit
count and accumulated reward.  One search thread (deterministic); V is a multiple of 1/256 so
the backup order cannot matter (H2).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pyoracle import MCTS_DEFAULTS, RefSelfPlay  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")

CASES = {
    # name: (board size, overrides)
    "mcts_19_r8192": (19, dict(rollouts_per_thread=8192, max_searches=3)),                      # BASELINE config 3 search settings
    "mcts_19_r256_dir": (19, dict(rollouts_per_thread=256, max_searches=24, policy_distri_cutoff=10)),
    "mcts_19_r256_ties": (19, dict(rollouts_per_thread=256, max_searches=12, net_tie_levels=5, root_epsilon=0.0)),
    "mcts_19_r512_client": (19, dict(rollouts_per_thread=512, rollouts_per_batch=8, virtual_loss=5, c_puct=0.85, max_searches=10,
                                     ply_pass_enabled=3, policy_distri_cutoff=30, net_salt=11)),   # start_client.sh flavour
    "mcts_19_r128_fresh": (19, dict(rollouts_per_thread=128, persistent_tree=0, max_searches=10, root_epsilon=0.0,
                                    unexplored_q_zero=1)),
    "mcts_9_r512": (9, dict(rollouts_per_thread=512, max_searches=70, policy_distri_cutoff=6, net_salt=3)),   # plays to game end
    # option space of TSOptions / SearchAlgoOptions (tree_search_options.h:23-229)
    "mcts_19_r128_vl0": (19, dict(rollouts_per_thread=128, max_searches=8, virtual_loss=0, net_salt=31)),
    "mcts_19_r128_noprior": (19, dict(rollouts_per_thread=128, max_searches=8, use_prior=0, virtual_loss=3, net_salt=32, root_epsilon=0.0)),
    "mcts_9_r128_rootq0": (9, dict(rollouts_per_thread=128, max_searches=30, root_unexplored_q_zero=1, virtual_loss=3, c_puct=0.5,
                                   net_salt=33, policy_distri_cutoff=5)),
    "mcts_9_r96_bs4": (9, dict(rollouts_per_thread=96, rollouts_per_batch=4, batchsize=4, max_searches=30, net_salt=34, net_tie_levels=4)),
    "mcts_9_r128_bs64": (9, dict(rollouts_per_thread=128, rollouts_per_batch=64, batchsize=64, max_searches=30, net_salt=35,
                                 ply_pass_enabled=4, komi=5.5)),
    "mcts_9_r64_ties": (9, dict(rollouts_per_thread=64, max_searches=60, net_tie_levels=3, root_epsilon=0.1, root_alpha=0.5)),
    # evaluation games: a second MCTSGoAI for White (game_selfplay.cc:165-185) with its own model ("actor_white" rows get another
    # stub net), puct / batch / rollout overrides (init_ai :51-70); games run to the cutoff and restart
    "mcts_9_eval_two_ai": (9, dict(rollouts_per_thread=64, rollouts_per_batch=8, batchsize=8, max_searches=60, black_ver=5, white_ver=6,
                                   net_salt=51, white_net_salt=52, white_puct=0.9, white_rollouts_per_batch=4,
                                   white_rollouts_per_thread=48, policy_distri_cutoff=6, move_cutoff=40)),
    "mcts_19_eval_swap": (19, dict(rollouts_per_thread=128, max_searches=10, black_ver=2, white_ver=3, player_swap=1, net_salt=53,
                                   white_net_salt=54, policy_distri_cutoff=4)),
    # TSOptions.pick_method = strongest_prior (tree_search.h:506-509; the MCTS policy is then the normalised priors)
    "mcts_9_pick_prior": (9, dict(rollouts_per_thread=64, max_searches=30, pick_method=1, net_salt=55, policy_distri_cutoff=5)),
    # GameOptions.white_use_policy_network_only: White's moves are MCTSAI_T::actPolicyOnly (mcts.h:83-90), one AI
    "mcts_9_policy_only_white": (9, dict(rollouts_per_thread=64, max_searches=40, white_policy_only=1, net_salt=56,
                                         policy_distri_cutoff=4, move_cutoff=30)),
    "mcts_9_policy_only_eval": (9, dict(rollouts_per_thread=64, max_searches=40, black_policy_only=1, black_ver=1, white_ver=2,
                                        net_salt=57, white_net_salt=58, move_cutoff=30)),
    # TSOptions without a bound on the batch: 128 rollouts per batch in one search thread (tree_search_options.h:81)
    "mcts_9_r256_bs128": (9, dict(rollouts_per_thread=256, rollouts_per_batch=128, batchsize=128, max_searches=20, net_salt=59,
                                  policy_distri_cutoff=4)),
    "mcts_19_r512_bs256": (19, dict(rollouts_per_thread=512, rollouts_per_batch=256, batchsize=256, max_searches=4, net_salt=60)),
    # round 4: the leaf table of a step is sized by the launch (up to 1024 leaves): 512 rollouts per batch, and 2 search threads x 384
    "mcts_9_r1024_bs512": (9, dict(rollouts_per_thread=1024, rollouts_per_batch=512, batchsize=512, max_searches=8, net_salt=64,
                                   policy_distri_cutoff=4)),
    # TSOptions.pick_method = uniform_random (tree_search.h:514-517): random_idx = rng() % edges from MCTSResultT::addActions'
    # process-wide `static std::mt19937 rng(time(NULL))` (tree_search_base.h:238).  fixed_time = what time() returns to the
    # reference in the generating process (oracle/ref_selfplay.cc refsp_set_time); one game, so the draws have one order; the
    # case runs in a process of its own because the generator is seeded once per process
    "mcts_9_pick_uniform": (9, dict(rollouts_per_thread=48, max_searches=40, net_salt=63, policy_distri_cutoff=6, num_games=1,
                                    pick_method=2, move_cutoff=25, fixed_time=1234567)),
    # round 5: mcts_threads > 1 on the REAL reference.  Its search threads race (tree_search.h:345-368), so these cases run the
    # turnstile build (oracle/Makefile: libelfsp*_ts.so = the same sources + four elf_ts_hook() calls inserted into a build-time copy
    # of batch_rollouts; oracle/ref_selfplay.cc): per round the threads descend in thread order, evaluate, set their evaluations and
    # back up in thread order.  Every thread's MCTSActor draws its D4 codes from its own generator (all seeded alike).
    "mcts_9_T2_r128": (9, dict(turnstile=1, mcts_threads=2, rollouts_per_thread=64, rollouts_per_batch=8, batchsize=8, max_searches=60,
                               net_salt=71, policy_distri_cutoff=6)),                                  # plays to the end of a game
    "mcts_9_T4_r256": (9, dict(turnstile=1, mcts_threads=4, rollouts_per_thread=64, rollouts_per_batch=16, batchsize=16, max_searches=30,
                               net_salt=72, policy_distri_cutoff=4, net_tie_levels=4)),
    "mcts_19_T2_r512": (19, dict(turnstile=1, mcts_threads=2, rollouts_per_thread=256, rollouts_per_batch=16, batchsize=16, max_searches=6,
                                 net_salt=73)),
    # the canonical client configuration (start_client.sh:11-30): 8 search threads x 1 rollout per batch, virtual loss 5, puct 0.85
    "mcts_19_T8_client": (19, dict(turnstile=1, mcts_threads=8, rollouts_per_thread=25, rollouts_per_batch=1, batchsize=8, virtual_loss=5,
                                   c_puct=0.85, max_searches=10, ply_pass_enabled=3, policy_distri_cutoff=30, net_salt=74)),
    "mcts_9_T3_eval_two_ai": (9, dict(turnstile=1, mcts_threads=3, rollouts_per_thread=32, rollouts_per_batch=4, batchsize=8, max_searches=40,
                                      black_ver=5, white_ver=6, net_salt=75, white_net_salt=76, white_puct=0.9, policy_distri_cutoff=6,
                                      move_cutoff=30)),
    # round 6: searches from DENSE 19x19 positions.  GameOptions.preload_sgf follows a ladder-suite game (game_selfplay.cc:202-219:
    # the first preload_sgf_move_to moves are forwarded before the first search; :392-405: every searched move is then replaced by the
    # SGF's next move, so the persistent tree advances along the game).  n_edges 160..300 instead of the 340..361 of every other 19x19
    # fixture; ply_pass_enabled below the preload ply, so the pass edge, Tromp-Taylor leaves (go/mcts/mcts.h:232-242) and
    # remove_pass_if_dangerous (:185-207) occur inside a 19x19 tree with dozens of groups, captures and kos.
    "mcts_19_sgf_p60": (19, dict(preload_sgf=("406844.sgf", 60), rollouts_per_thread=512, max_searches=5, ply_pass_enabled=50, net_salt=81)),
    "mcts_19_sgf_p120": (19, dict(preload_sgf=("406844.sgf", 120), rollouts_per_thread=512, max_searches=5, ply_pass_enabled=100, net_salt=82)),
    "mcts_19_sgf_p180": (19, dict(preload_sgf=("406844.sgf", 180), rollouts_per_thread=512, max_searches=5, ply_pass_enabled=100, net_salt=83)),
    "mcts_19_sgf_p195": (19, dict(preload_sgf=("406844.sgf", 195), rollouts_per_thread=512, max_searches=5, ply_pass_enabled=100, net_salt=84,
                                  net_tie_levels=5)),
    "mcts_19_sgf_b_p150": (19, dict(preload_sgf=("@longest", 150), rollouts_per_thread=512, max_searches=6, ply_pass_enabled=20, net_salt=85,
                                    virtual_loss=5, c_puct=0.85, rollouts_per_batch=8, batchsize=8)),
    "mcts_19_sgf_c_p100": (19, dict(preload_sgf=("@second", 100), rollouts_per_thread=512, max_searches=6, ply_pass_enabled=60, net_salt=86,
                                    persistent_tree=0, root_epsilon=0.0)),
    "mcts_19_sgf_T2_p140": (19, dict(turnstile=1, mcts_threads=2, preload_sgf=("406844.sgf", 140), rollouts_per_thread=256, max_searches=5,
                                     ply_pass_enabled=100, net_salt=87)),
    "mcts_19_sgf_r8192_p160": (19, dict(preload_sgf=("406844.sgf", 160), rollouts_per_thread=8192, max_searches=2, ply_pass_enabled=100,
                                        net_salt=88)),
}

LADDER = "/root/reference/ladder_suite/ladder"


def ladder_game(R, which):
    """(path, moves) of a ladder-suite game through the reference's own Sgf loader: a file name, or the longest / second-longest
    game of the suite whose every move is legal"""
    from pyoracle import Ref
    ref = Ref(19)
    if not which.startswith("@"):
        path = os.path.join(LADDER, which)
        return path, ref.sgf_moves(path)[0]
    g = np.load(os.path.join(OUT, "ladder_suite.npz"))
    lens = np.diff(g["offsets"])
    order = [i for i in np.argsort(-lens, kind="stable") if str(g["names"][i]) != "406844.sgf"]
    i = order[0 if which == "@longest" else 1]
    path = os.path.join(LADDER, str(g["names"][i]))
    return path, ref.sgf_moves(path)[0]


def run_case(name, path):
    n, kw = CASES[name]
    kw = dict(kw)
    R = RefSelfPlay(n, turnstile=bool(kw.pop("turnstile", 0)))
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(kw)
    fixed_time = cfg.pop("fixed_time", None)
    pre = cfg.pop("preload_sgf", None)
    extra0 = {}
    if pre is not None:
        sgf_path, mv = ladder_game(R, pre[0])
        R.set_preload(sgf_path, pre[1])
        extra0 = dict(preload_moves=np.asarray(mv, np.uint16), preload_move_to=np.int32(pre[1]), preload_name=np.array(os.path.basename(sgf_path)))
    if fixed_time is None:
        r = R.run(**cfg)
        r2 = R.run(**cfg)
        assert all(np.array_equal(r[k], r2[k]) for k in ("coord", "visits", "prior", "reward")), "reference not deterministic"
        extra = {}
    else:
        R.set_time(fixed_time)
        r = R.run(**cfg)
        extra = dict(fixed_time=np.int64(fixed_time))
    R.set_preload("", -1)
    extra.update(extra0)
    S = r["search"]
    np.savez_compressed(
        path,
        board_size=np.int32(n),
        cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()], np.float64),
        move_played=np.array([s.move_played for s in S], np.int32),
        best_action=np.array([s.best_action for s in S], np.int32),
        total_visits=np.array([s.total_visits for s in S], np.int32),
        n_edges=np.array([s.n_edges for s in S], np.int32),
        root_value=np.array([s.root_value for s in S], np.float32),
        coord=r["coord"].astype(np.int16), visits=r["visits"], prior=r["prior"], reward=r["reward"],
        rows=np.int64(r["rows"]), white_rows=np.int64(r["white_rows"]), **extra)
    print(name, "searches", len(S), "moves", [s.move_played for s in S][:12], "rows", r["rows"], "ref search s", r["usec"] / 1e6)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--one" in sys.argv:
        i = sys.argv.index("--one")
        run_case(sys.argv[i + 1], sys.argv[i + 2])
        return
    for name, (n, kw) in CASES.items():
        path = os.path.join(OUT, name + ".npz")
        if "--all" not in sys.argv and os.path.exists(path):
            continue
        if "fixed_time" not in kw:
            run_case(name, path)
            continue
        import subprocess
        tmp = ["/tmp/%s_%d.npz" % (name, i) for i in range(2)]
        for t in tmp:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, t], check=True)
        a, b = np.load(tmp[0]), np.load(tmp[1])
        assert all(np.array_equal(a[k], b[k]) for k in a.files), "reference not deterministic under a fixed clock"
        os.replace(tmp[0], path)


if __name__ == "__main__":
    main()
