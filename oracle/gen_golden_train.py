"""Generates tests/golden/records_*.npz and train_*.npz from the REAL reference (oracle/_ref/libelfsp*.so). Build container only:
    python oracle/gen_golden_train.py

records_<n>_<case>.npz   Record JSON of finished self-play games exactly as GoStateExt::dumpRecord + Record::setJsonFields +
                         json::dump() produce them (GameNotifierBase::OnGameEnd in oracle/ref_selfplay.cc), with the run's
                         configuration, so the GPU self-play can be asked to reproduce them.
train_<n>.npz            rows of the reference's "train" batch (real GoStateExtOffline::fromRecord / switchBeforeMove +
                         GoFeature extractors) for chosen (record, move_to, D4 code, num_future_actions), over (a) the self-play
                         records above and (b) long synthetic records: config-2 random games (captures, kos, passes) with random
                         quantised policies and values, serialised as Record JSON.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pyoracle import MCTS_DEFAULTS, Port, RefSelfPlay, playout_seeds  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")

RECORD_CASES = {
    "records_9_cutoff": (9, dict(rollouts_per_thread=64, max_searches=50, policy_distri_cutoff=6, net_salt=3, move_cutoff=24)),
    "records_9_resign": (9, dict(rollouts_per_thread=32, max_searches=120, policy_distri_cutoff=4, net_salt=5, resign_thres=0.9,
                                 move_cutoff=70)),
    # a game to the move limit (FR_MAX_STEP at ply 162) and one ended by two passes (FR_TWO_PASSES), pass enabled from ply 6
    "records_9_twopass": (9, dict(rollouts_per_thread=48, max_searches=340, policy_distri_cutoff=4, net_salt=13, ply_pass_enabled=6)),
    # never_resign_prob 0.5: two games draw "never resign" and run to the cutoff, one is resigned by Black, one by White
    "records_9_neverresign": (9, dict(rollouts_per_thread=32, max_searches=236, policy_distri_cutoff=3, net_salt=21, resign_thres=0.9,
                                      never_resign_prob=0.5, move_cutoff=64, seed=5)),
    "records_19_resign": (19, dict(rollouts_per_thread=32, max_searches=130, policy_distri_cutoff=12, net_salt=41, resign_thres=0.9,
                                   move_cutoff=80, seed=9)),                         # White resigns one game, Black the next
    # GameOptions.preload_sgf (game_selfplay.cc:202-219,392-405): 40 moves of a capture-rich random game forwarded, then 40
    # searches whose move is replaced by the SGF's, the game finished by the search that finds the SGF exhausted, and the
    # degenerate one-search games that follow; exercises treeAdvance along moves the search did not choose
    "records_9_preload": (9, dict(rollouts_per_thread=64, max_searches=44, policy_distri_cutoff=50, net_salt=17,
                                  preload=(909, 80, 40))),
    "records_19_cutoff": (19, dict(rollouts_per_thread=32, max_searches=64, policy_distri_cutoff=30, net_salt=9, move_cutoff=31)),
}


# round 3: records of evaluation games (Record.request carries white_ver / player_swap, using_models both versions); written as
# records_*.npz only -- the train_*.npz rows above stay as they were generated
SKIP, PASS, RESIGN, CLEAR = -100, -99, -98, -97   # SpecialActionType (common/game_feature.h:17) = GameContext.getParams()["ACTION_*"]
RECORD_CASES_R3 = {
    "records_9_eval": (9, dict(rollouts_per_thread=48, max_searches=70, policy_distri_cutoff=6, net_salt=71, white_net_salt=72,
                               black_ver=11, white_ver=12, white_rollouts_per_thread=32, white_puct=1.2, move_cutoff=24)),
    "records_9_eval_swap_resign": (9, dict(rollouts_per_thread=32, max_searches=160, policy_distri_cutoff=4, net_salt=73, white_net_salt=74,
                                           black_ver=21, white_ver=20, player_swap=1, resign_thres=0.9, move_cutoff=70)),
    # a second request while the (single) game plays, put into its mailbox during search 7 and read at the game's next fifth act
    # (game_selfplay.cc:272-290): other version, not async -> the running game is dropped and restarted under the new request
    # (restart() :159-220, seq advances, nothing recorded); async -> the game goes on, its record names both models
    "records_9_req2_restart": (9, dict(rollouts_per_thread=48, max_searches=44, policy_distri_cutoff=4, net_salt=75, num_games=1,
                                       black_ver=3, move_cutoff=14, req2_after_searches=7, req2_black_ver=4)),
    "records_9_req2_async": (9, dict(rollouts_per_thread=48, max_searches=44, policy_distri_cutoff=4, net_salt=75, num_games=1,
                                     black_ver=3, move_cutoff=14, req2_after_searches=7, req2_black_ver=4, req2_async=1)),
    # the second request carries its own ModelPair.mcts_opt, as every request of the reference's server does: same model, other
    # search options (restart because ModelPair::operator== compares mcts_opt); and an evaluation request as EvalSubCtrl writes it
    # (train/ctrl_eval.h:227-237: white_ver >= 0, Dirichlet noise and both q_zero flags off)
    "records_9_req2_ts": (9, dict(rollouts_per_thread=48, max_searches=44, policy_distri_cutoff=4, net_salt=79, num_games=1, black_ver=3,
                                  move_cutoff=14, req2_after_searches=7, req2_black_ver=3, req2_ts=1, req2_rollouts_per_thread=32,
                                  req2_rollouts_per_batch=8, req2_c_puct=0.9, req2_root_epsilon=0.0, req2_root_alpha=0.0)),
    "records_9_req2_eval": (9, dict(rollouts_per_thread=48, max_searches=44, policy_distri_cutoff=4, net_salt=80, white_net_salt=82, num_games=1,
                                    black_ver=3, move_cutoff=14, unexplored_q_zero=1, root_unexplored_q_zero=1, req2_after_searches=7,
                                    req2_black_ver=4, req2_white_ver=3, req2_ts=1, req2_rollouts_per_thread=48, req2_rollouts_per_batch=16,
                                    req2_c_puct=1.5, req2_root_epsilon=0.0, req2_root_alpha=0.0, req2_unexplored_q_zero=0,
                                    req2_root_unexplored_q_zero=0)),
    # GameOptions.cheat_* (finish_game, game_selfplay.cc:122-129 -> GoStateExt::setFinalValue go_state_ext.h:86-99): the result of
    # a self-play game is a draw of the game's generator (also for a resigned game; the draw shifts the stream the next game
    # samples its moves from); the result of an evaluation game is the parity of a hash of the two version strings, negated
    # under player_swap
    "records_9_cheat_selfplay": (9, dict(rollouts_per_thread=32, max_searches=150, policy_distri_cutoff=6, net_salt=76, num_games=1,
                                         resign_thres=0.9, move_cutoff=64, cheat_selfplay_random_result=1)),
    "records_9_cheat_eval": (9, dict(rollouts_per_thread=32, max_searches=60, policy_distri_cutoff=4, net_salt=77, white_net_salt=78,
                                     black_ver=11, white_ver=12, move_cutoff=18, cheat_eval_new_model_wins_half=1)),
    "records_9_cheat_eval_swap": (9, dict(rollouts_per_thread=32, max_searches=60, policy_distri_cutoff=4, net_salt=77, white_net_salt=78,
                                          black_ver=14, white_ver=12, player_swap=1, move_cutoff=18, cheat_eval_new_model_wins_half=1)),
    # GoStateExt::dumpSgf of every finished game (what dump_record_prefix writes; the `sgfs` array of fixtures generated from here
    # on): resigned and scored games, a policy-only White
    "records_9_sgf": (9, dict(rollouts_per_thread=32, max_searches=220, policy_distri_cutoff=4, net_salt=91, num_games=1,
                              resign_thres=0.95, move_cutoff=60, komi=6.5)),
    "records_9_sgf_policy_only": (9, dict(rollouts_per_thread=32, max_searches=60, policy_distri_cutoff=4, net_salt=92, num_games=1,
                                          move_cutoff=12, white_policy_only=1)),
    # GameOptions.mode = "online": the human_actor prompts of GoGameSelfPlay::act :290-330 answered from a script (a move, the same
    # move again = illegal and prompted again, SKIP = the AI searches and moves, PASS, CLEAR, RESIGN), every prompt's feature planes
    # kept; following_pass (mcts_update_info :104-111) with a stub net that is sure White wins / sure Black wins
    "online_9_script": (9, dict(rollouts_per_thread=32, rollouts_per_batch=8, batchsize=8, max_searches=100, net_salt=81, num_games=1, online=1,
                                human_script=[30, 30, SKIP, 50, SKIP, 22, SKIP, PASS, SKIP, CLEAR, 40, SKIP, RESIGN, 11, SKIP, 12])),
    "online_9_following_pass": (9, dict(rollouts_per_thread=32, rollouts_per_batch=8, batchsize=8, max_searches=100, net_salt=81, num_games=1,
                                        online=1, following_pass=1, net_value_on=1, net_value=-1.0,
                                        human_script=[PASS, SKIP, 33, SKIP, PASS, SKIP, 20])),
    "online_9_not_following": (9, dict(rollouts_per_thread=32, rollouts_per_batch=8, batchsize=8, max_searches=100, net_salt=81, num_games=1,
                                       online=1, following_pass=1, net_value_on=1, net_value=1.0,
                                       human_script=[PASS, SKIP, 33, SKIP, PASS, SKIP, 20, CLEAR, 5])),
    # round 6: Records of games that start from a DENSE 19x19 position -- ladder-suite game 406844.sgf preloaded to ply 170
    # (GameOptions.preload_sgf, game_selfplay.cc:202-219), the remaining 30 moves searched (64 rollouts each, pass enabled) and replaced
    # by the SGF's (:392-405), the game finished when the SGF is exhausted (FR_MAX_STEP :393-395), restarted and preloaded again: two
    # finished games, 200-move contents, quantised policies of late-game positions (~190 legal moves, pass edges)
    "records_19_sgf_preload": (19, dict(rollouts_per_thread=64, max_searches=64, policy_distri_cutoff=0, net_salt=91, ply_pass_enabled=100,
                                        preload_sgf=("406844.sgf", 170))),
}


def split_records(txt):
    return [json.dumps(j, separators=(",", ":")) for j in json.loads(txt)]


def synth_record(n, seed, rng, with_policies):
    """A long game by the config-2 policy as a Record dict (moves via the reference's own coords2sgfstr)."""
    port = Port(n)
    s = port.new()
    mv = port.playout_moves(s, int(seed))
    port.free(s)
    R = RefSelfPlay(n)
    P = (n + 2) ** 2
    k = len(mv)
    npol = int(rng.integers(0, k + 1)) if with_policies else 0
    pol = np.zeros((npol, P), np.uint8)
    for i in range(npol):
        idx = rng.integers(0, P, size=int(rng.integers(1, 40)))
        pol[i, idx] = rng.integers(1, 256, size=idx.size)
    vals = np.round(np.tanh(rng.standard_normal(k)) * 1e4) / 1e4
    return {"offline": False, "pri": 0.0,
            "request": {"client_ctrl": {"async": False, "black_resign_thres": 0.0, "client_type": 1, "never_resign_prob": 0.0,
                                        "num_game_thread_used": 1, "player_swap": False, "white_resign_thres": 0.0},
                        "vers": {"black_ver": int(rng.integers(0, 1000)), "white_ver": -1,
                                 "mcts_opt": {"alg_opt": {"c_puct": 1.5, "root_unexplored_q_zero": False, "unexplored_q_zero": False,
                                                          "use_prior": True},
                                              "log_prefix": "", "max_num_moves": 0, "num_rollouts_per_batch": 16,
                                              "num_rollouts_per_thread": 64, "num_threads": 1, "persistent_tree": True,
                                              "pick_method": "most_visited", "root_alpha": 0.03, "root_epsilon": 0.25, "seed": 0,
                                              "verbose": False, "verbose_time": False, "virtual_loss": 1}}},
            "result": {"black_never_resign": False, "white_never_resign": False, "content": R.coords2sgfstr(mv), "num_move": k,
                       "policies": pol.tolist(), "reward": float(rng.choice([-7.5, 12.5, -0.5, 0.5])), "using_models": [0],
                       "values": [float(np.float32(v)) for v in vals]},
            "seq": int(rng.integers(2, 50)), "thread_id": 0, "timestamp": 0}


def dump_case(name, n, kw):
    if "--missing" in sys.argv and os.path.exists(os.path.join(OUT, name + ".npz")):
        return []
    R = RefSelfPlay(n)
    cfg = dict(MCTS_DEFAULTS)
    kw = dict(kw)
    pre = kw.pop("preload", None)
    pre_sgf = kw.pop("preload_sgf", None)
    extra = {}
    if pre_sgf is not None:
        from pyoracle import Ref
        path = os.path.join("/root/reference/ladder_suite/ladder", pre_sgf[0])
        mv = Ref(n).sgf_moves(path)[0]
        R.set_preload(path, pre_sgf[1])
        extra = dict(preload_moves=np.array(mv, np.uint16), preload_move_to=np.int32(pre_sgf[1]), preload_name=np.array(pre_sgf[0]))
    if pre is not None:
        port = Port(n)
        st = port.new()
        mv = port.playout_moves(st, int(playout_seeds(1, base=pre[0])[0]))[: pre[1]]
        port.free(st)
        path = "/tmp/elf_preload_%s.sgf" % name
        with open(path, "w") as fh:
            fh.write("(;GM[1]FF[4]SZ[%d]KM[7.5]" % n + R.coords2sgfstr(mv)[1:])
        R.set_preload(path, pre[2])
        extra = dict(preload_moves=np.array(mv, np.uint16), preload_move_to=np.int32(pre[2]))
    script = kw.pop("human_script", None)
    cfg.update(kw)
    r = R.run(human_script=script, **cfg)
    R.set_preload("", -1)
    if script is not None:
        r2 = R.run(human_script=script, **cfg)
        strip = lambda t: [dict(j, timestamp=0) for j in json.loads(t)]
        assert np.array_equal(r["prompts"], r2["prompts"]) and strip(r["records"]) == strip(r2["records"]) and np.array_equal(r["visits"], r2["visits"]), name
        extra.update(human_script=np.array(script, np.int64), prompts=np.packbits(r["prompts"].reshape(len(r["prompts"]), -1), axis=1))
    recs = json.loads(r["records"])
    exact = [R.record_roundtrip(t) for t in split_records(r["records"])]
    assert len(recs) >= 1, name
    # the round trip through Record::createFromJson must reproduce the text the game thread dumped
    assert "[" + ",".join(exact) + "]" == r["records"], name
    np.savez_compressed(os.path.join(OUT, name + ".npz"), board_size=np.int32(n),
                        cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()], np.float64),
                        records=np.array(exact), searches=np.int32(len(r["search"])),
                        move_played=np.array([s.move_played for s in r["search"]], np.int32),
                        best_action=np.array([s.best_action for s in r["search"]], np.int32),
                        n_edges=np.array([s.n_edges for s in r["search"]], np.int32),
                        root_value=np.array([s.root_value for s in r["search"]], np.float32),
                        coord=r["coord"].astype(np.int16), visits=r["visits"], prior=r["prior"], reward=r["reward"],
                        game_starts=np.int32(r["game_starts"]), start_versions=np.array(r["start_versions"], np.int64),
                        sgfs=np.array(R.last_sgfs()), **extra)
    print(name, "records", len(exact), [(j["seq"], j["result"]["num_move"], j["result"]["reward"], len(j["result"].get("policies", [])))
                                        for j in recs])
    return exact


# the reference's trainer input path end to end (reftrain_act: ReaderQueuesT<Record> + one real GoGameTrain thread + the "train"
# batch group): (board size, replay-buffer shape, seeds, acts of 64 rows, num_future_actions)
TRAIN_ACT_CASES = {
    "train_act_9": (9, dict(num_reader=4, q_min_size=1, q_max_size=1000, insert_seed=3, game_seed=11, num_acts=3, num_future_actions=1)),
    "train_act_9_evict": (9, dict(num_reader=2, q_min_size=2, q_max_size=4, insert_seed=5, game_seed=12, num_acts=2, num_future_actions=3)),
    "train_act_19": (19, dict(num_reader=2, q_min_size=1, q_max_size=50, insert_seed=7, game_seed=13, num_acts=1, num_future_actions=1)),
}


def dump_train_act(name, n, kw):
    """records = those of train_<n>.npz, each given its own black_ver (1000 + index) so that a row's selfplay_ver names its record"""
    if "--missing" in sys.argv and os.path.exists(os.path.join(OUT, name + ".npz")):
        return
    src = np.load(os.path.join(OUT, "train_%d.npz" % n))
    recs = []
    for i, t in enumerate(src["records"]):
        j = json.loads(str(t))
        j["request"]["vers"]["black_ver"] = 1000 + i
        recs.append(json.dumps(j, separators=(",", ":")))
    R = RefSelfPlay(n)
    a, b = R.train_act(recs, **kw), R.train_act(recs, **kw)
    assert all(np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)) for k in a), name + ": reference not deterministic"
    np.savez_compressed(os.path.join(OUT, name + ".npz"), board_size=np.int32(n), records=np.array(recs),
                        cfg_keys=np.array(list(kw.keys())), cfg_vals=np.array([int(v) for v in kw.values()], np.int64),
                        rec=(a["selfplay_ver"] - 1000).astype(np.int32), s=np.packbits(a["s"].reshape(len(a["s"]), -1), axis=1),
                        **{k: a[k] for k in ("offline_a", "winner", "mcts_scores", "move_idx", "num_move", "aug_code", "selfplay_ver")})
    print(name, "rows", len(a["winner"]), "records used", sorted(set((a["selfplay_ver"] - 1000).tolist())))


# A 9x9 game that ends by positional superko at move 107 (found with the restated board engine: uniformly random legal moves, eye
# fills included, until GoState::terminated() by repetition), followed by 32 more recorded moves.  GoState::forward refuses every
# move after the repetition (go_state.cc:78-79), so a replay to any later move_to stays on the position of ply 108 -- which is what
# a loader that replays from stored checkpoints must reproduce (round 4: k_replay_checkpoint / k_replay_extract<KEEP = false>).
SUPERKO_GAME_9 = [60, 95, 48, 108, 83, 94, 18, 15, 41, 46, 96, 89, 71, 40, 67, 25, 79, 58, 86, 103, 17, 82, 107, 59, 106, 53, 63, 36, 69, 28, 34, 12,
                  102, 56, 51, 68, 105, 80, 20, 27, 81, 39, 70, 35, 29, 50, 92, 78, 91, 37, 75, 30, 42, 38, 45, 62, 47, 57, 61, 84, 100, 64, 85, 104,
                  13, 31, 14, 19, 93, 72, 23, 16, 103, 18, 101, 90, 73, 52, 82, 67, 97, 41, 104, 42, 62, 49, 94, 84, 12, 26, 74, 29, 108, 48, 95, 24,
                  80, 45, 14, 12, 23, 17, 72, 20, 13, 12, 14]


def dump_superko():
    n = 9
    R = RefSelfPlay(n)
    port = Port(n)
    st = port.new()
    for c in SUPERKO_GAME_9:
        assert port.forward(st, c) == 1
    assert port.terminated(st) and port.info(st)[0] == len(SUPERKO_GAME_9) + 1 < 2 * n * n      # ended by repetition, not by the move limit
    port.free(st)
    rng = np.random.default_rng(4109)
    rec = synth_record(n, playout_seeds(1, base=4109)[0], rng, with_policies=True)
    tail = [int(c) for c in R.sgfstr2coords(rec["result"]["content"])][:32]
    mv = SUPERKO_GAME_9 + tail
    k = len(mv)
    P = (n + 2) ** 2
    pol = np.zeros((k, P), np.uint8)
    for i in range(k):
        idx = rng.integers(0, P, size=int(rng.integers(1, 40)))
        pol[i, idx] = rng.integers(1, 256, size=idx.size)
    rec["result"].update(content=R.coords2sgfstr(mv), num_move=k, policies=pol.tolist(),
                         values=[float(np.float32(v)) for v in np.round(np.tanh(rng.standard_normal(k)) * 1e4) / 1e4])
    t = json.dumps(rec, separators=(",", ":"))
    rows = []
    for nfa in (1, 3):
        for ci, mt in enumerate([0, 1, 31, 32, 33, 63, 64, 65, 95, 96, 97, 100, 105, 106, 107, 108, 109, 120, 127, 128, 129, 136, k - 3, k - 1]):
            if mt > k - nfa:
                continue
            d4 = (ci + nfa) % 8
            o = R.train_sample(t, mt, d4, nfa)
            oa = np.zeros(3, np.int64)
            oa[:nfa] = o["offline_a"]
            rows.append(dict(rec=0, move_to=mt, d4=d4, nfa=nfa, s=np.packbits(o["s"].astype(np.uint8).ravel()), offline_a=oa,
                             winner=o["winner"], mcts_scores=o["mcts_scores"], predicted_value=o["predicted_value"],
                             move_idx=o["move_idx"], num_move=o["num_move"], aug_code=o["aug_code"], selfplay_ver=o["selfplay_ver"]))
    late = [r for r in rows if r["move_to"] > len(SUPERKO_GAME_9)]
    assert late and all(int(r["move_idx"]) == len(SUPERKO_GAME_9) for r in late)   # the replay stopped at the repetition
    np.savez_compressed(os.path.join(OUT, "train_9_superko.npz"), board_size=np.int32(n), records=np.array([t]),
                        **{kk: np.array([r[kk] for r in rows]) for kk in rows[0].keys()})
    print("train_9_superko: %d rows, %d of them beyond the repetition" % (len(rows), len(late)))


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--superko" in sys.argv:
        return dump_superko()
    for name, (n, kw) in TRAIN_ACT_CASES.items():
        dump_train_act(name, n, kw)
    for name, (n, kw) in RECORD_CASES_R3.items():
        dump_case(name, n, kw)
    if "--only-r3" in sys.argv:
        return
    by_size = {9: [], 19: []}
    for name, (n, kw) in RECORD_CASES.items():
        by_size[n] += dump_case(name, n, kw)
    for n in (9, 19):
        R = RefSelfPlay(n)
        rng = np.random.default_rng(100 + n)
        recs = list(by_size[n])
        for i, sd in enumerate(playout_seeds(6 if n == 19 else 8, base=300 + n)):
            recs.append(json.dumps(synth_record(n, sd, rng, with_policies=(i % 3 != 2)), separators=(",", ":")))
        # a record with moves GoState::forward refuses (a stone on an occupied point, twice): the replay skips them, so every
        # extractor's index (getPly() - 1) falls behind the requested ply
        bad = synth_record(n, playout_seeds(1, base=777 + n)[0], rng, with_policies=True)
        mvb = [int(c) for c in RefSelfPlay(n).sgfstr2coords(bad["result"]["content"])]
        mvb[10] = mvb[8]
        mvb[21] = mvb[19]
        bad["result"]["content"] = RefSelfPlay(n).coords2sgfstr(mvb)
        recs.append(json.dumps(bad, separators=(",", ":")))
        bad_index = len(recs) - 1
        rows = []
        for ri, t in enumerate(recs):
            j = json.loads(t)
            nm = len(R.sgfstr2coords(j["result"]["content"]))
            for nfa in (1, 3):
                if nm < nfa:
                    continue
                last = nm - nfa
                cand = sorted(set([0, 1, 2, 7, 8, 9, last // 2, max(last - 1, 0), last] + ([10, 11, 12, 21, 22, 23, 30] if ri == bad_index else [])))
                for ci, mt in enumerate(c for c in cand if c <= last):
                    d4 = (ri + ci + nfa) % 8
                    o = R.train_sample(t, mt, d4, nfa)
                    oa = np.zeros(3, np.int64)
                    oa[:nfa] = o["offline_a"]
                    rows.append(dict(rec=ri, move_to=mt, d4=d4, nfa=nfa, s=np.packbits(o["s"].astype(np.uint8).ravel()), offline_a=oa,
                                     winner=o["winner"], mcts_scores=o["mcts_scores"], predicted_value=o["predicted_value"],
                                     move_idx=o["move_idx"], num_move=o["num_move"], aug_code=o["aug_code"], selfplay_ver=o["selfplay_ver"]))
        keys = rows[0].keys()
        np.savez_compressed(os.path.join(OUT, "train_%d.npz" % n), board_size=np.int32(n), records=np.array(recs),
                            **{k: np.array([r[k] for r in rows]) for k in keys})
        print("train_%d: %d records, %d rows" % (n, len(recs), len(rows)))


if __name__ == "__main__":
    main()
