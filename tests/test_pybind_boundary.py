"""The drop-in boundary proper: the pybind11 modules _elf / _elfgames_go / _elfgames_go_inference (elf_amd/csrc/pybind_elf.cc,
built into elf_amd/ext/) with the names of src_cpp/elf/Pybind.cc:27-117, elfgames/go/train/Pybind.cc:18-63 and
elfgames/go/inference/Pybind.cc:18-45, driven by the Python half of the reference's batch interface.

CPU (here): names and defaults; the reference's UNMODIFIED src_py/elf/utils_elf.py (loaded by file path from /root/reference when
that tree is present) allocates its batches against our Context; tests/gcwrapper_restated.py -- the stand-in used on the GPU box,
where /root/reference does not exist -- is checked to make the same calls as the original on a recording mock.
GPU: a game.py:365-402-shaped GCWrapper session reproduces the reference fixture mcts_19_r256_dir bit for bit, with the batch
tensors in pinned host memory (what Allocator._alloc makes) and device-resident; chunking, rv check, game_start / game_end,
GameStats, online mode (human_actor)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_UTILS = "/root/reference/src_py/elf/utils_elf.py"
EXT = os.path.join(ROOT, "elf_amd", "ext")


@pytest.fixture(scope="module")
def mods(built):
    if EXT not in sys.path:
        sys.path.insert(0, EXT)
    import _elf
    import _elfgames_go
    import _elfgames_go_inference
    return _elf, _elfgames_go, _elfgames_go_inference


def load_reference_utils():
    if not os.path.exists(REF_UTILS):
        return None
    spec = importlib.util.spec_from_file_location("reference_utils_elf", REF_UTILS)   # the file itself, not the elf package:
    m = importlib.util.module_from_spec(spec)                                            # its __init__ pulls the ZMQ/option plumbing
    spec.loader.exec_module(m)
    return m


def game_py_desc(batchsize):
    """src_py/elfgames/go/game.py:379-402 (mode selfplay)"""
    return {"actor_black": dict(input=["s"], reply=["pi", "V", "a", "rv"], batchsize=batchsize, timeout_usec=10),
            "actor_white": dict(input=["s"], reply=["pi", "V", "a", "rv"], batchsize=batchsize, timeout_usec=10),
            "game_end": dict(batchsize=1),
            "game_start": dict(batchsize=1, input=["black_ver", "white_ver"], reply=None)}


def options_from_cfg(go, cfg, n=19, **over):
    co, opt = go.ContextOptions(), go.GameOptions()
    co.num_games, co.batchsize, co.job_id = int(cfg["num_games"]), int(cfg["batchsize"]), "test"
    ts = co.mcts_options
    ts.num_threads, ts.num_rollouts_per_thread, ts.num_rollouts_per_batch = int(cfg["mcts_threads"]), int(cfg["rollouts_per_thread"]), int(cfg["rollouts_per_batch"])
    ts.virtual_loss, ts.persistent_tree = int(cfg["virtual_loss"]), bool(cfg["persistent_tree"])
    ts.root_epsilon, ts.root_alpha = float(np.float32(cfg["root_epsilon"])), float(np.float32(cfg["root_alpha"]))
    ts.alg_opt.c_puct, ts.alg_opt.use_prior = float(np.float32(cfg["c_puct"])), bool(cfg["use_prior"])
    ts.alg_opt.unexplored_q_zero, ts.alg_opt.root_unexplored_q_zero = bool(cfg["unexplored_q_zero"]), bool(cfg["root_unexplored_q_zero"])
    opt.mode, opt.seed, opt.komi = "selfplay", int(cfg["seed"]), float(np.float32(cfg["komi"]))
    opt.policy_distri_cutoff, opt.ply_pass_enabled, opt.move_cutoff = int(cfg["policy_distri_cutoff"]), int(cfg["ply_pass_enabled"]), int(cfg["move_cutoff"])
    opt.use_mcts, opt.board_size = True, n
    if "white_ver" in cfg:          # round-3 fixtures: evaluation games, pick methods, policy-only play
        opt.white_puct = float(np.float32(cfg["white_puct"]))
        opt.white_mcts_rollout_per_batch, opt.white_mcts_rollout_per_thread = int(cfg["white_rollouts_per_batch"]), int(cfg["white_rollouts_per_thread"])
        opt.black_use_policy_network_only, opt.white_use_policy_network_only = bool(cfg["black_policy_only"]), bool(cfg["white_policy_only"])
        ts.pick_method = {0: "most_visited", 1: "strongest_prior", 2: "uniform_random"}[int(cfg["pick_method"])]
    for k, v in over.items():
        setattr(opt, k, v)
    return co, opt


# ---------------------------------------------------------------------------------------------------------------- CPU
def test_module_and_class_names_are_the_reference_s(mods):
    _elf, go, goi = mods
    for name in ("Context", "SharedMem", "SharedMemOptions", "AnyP", "FuncMapBase", "Size", "ReplyStatus", "TSOptions", "SearchAlgoOptions",
                 "SUCCESS", "FAILED", "UNKNOWN", "_logging", "_options"):
        assert hasattr(_elf, name), name                                   # elf/Pybind.cc:27-117
    for name in ("GameContext", "ContextOptions", "GameOptions", "GoGameSelfPlay"):
        assert hasattr(goi, name) and hasattr(go, name), name              # inference/Pybind.cc:18-45
    for name in ("Client", "Server", "GameStats", "WinRateStats"):
        assert hasattr(go, name), name                                     # train/Pybind.cc:34-57
    for cls, methods in ((_elf.Context, "wait step start stop version allocateSharedMem createSharedMemOptions"),
                         (_elf.SharedMem, "__getitem__ getSharedMemOptions effective_batchsize info"),
                         (_elf.SharedMemOptions, "idx batchsize label setTimeout"), (_elf.AnyP, "info field set"),
                         (_elf.FuncMapBase, "batchsize name sz type_name type_size"), (_elf.Size, "vec"),
                         (goi.GameContext, "ctx getParams getGame setRequest"), (go.GameContext, "ctx getParams getGame getClient getServer"),
                         (go.Client, "setRequest getGameStats"), (go.GameStats, "getWinRateStats getPlayedGames"),
                         (go.GoGameSelfPlay, "showBoard getNextPlayer getLastMove getScore getLastScore")):
        for m in methods.split():
            assert hasattr(cls, m), (cls, m)
    ts = _elf.TSOptions()                                                  # tree_search_options.h:77-94 defaults
    assert (ts.max_num_moves, ts.num_threads, ts.num_rollouts_per_thread, ts.num_rollouts_per_batch, ts.persistent_tree, ts.pick_method,
            ts.virtual_loss, ts.root_epsilon, ts.alg_opt.c_puct, ts.alg_opt.use_prior) == (0, 16, 100, 8, False, "most_visited", 0, 0.0, 5.0, True)
    opt = go.GameOptions()                                                 # go_game_specific.h:16-131 defaults
    assert (opt.num_future_actions, opt.move_cutoff, opt.policy_distri_cutoff, opt.komi, opt.ply_pass_enabled, opt.white_puct,
            opt.num_reset_ranking, opt.q_max_size, opt.preload_sgf_move_to) == (3, -1, 20, 7.5, 0, -1.0, 5000, 1000, -1)
    assert abs(opt.resign_thres - 0.05) < 1e-7 and abs(opt.resign_prob_never - 0.1) < 1e-7 and opt.list_files == []
    for f in ("seed mode data_aug start_ratio_pre_moves ratio_pre_moves list_files verbose num_games_per_thread use_mcts server_addr server_id port "
              "client_max_delay_sec q_min_size num_reader dump_record_prefix use_mcts_ai2 preload_sgf print_result resign_thres_lower_bound "
              "resign_thres_upper_bound resign_target_fp_rate following_pass use_df_feature policy_distri_training_for_all "
              "black_use_policy_network_only white_use_policy_network_only cheat_eval_new_model_wins_half cheat_selfplay_random_result "
              "eval_num_games selfplay_init_num selfplay_update_num selfplay_async white_mcts_rollout_per_batch white_mcts_rollout_per_thread "
              "eval_thres keep_prev_selfplay expected_num_clients").split():
        assert hasattr(opt, f), f                                          # REGISTER_PYBIND_FIELDS :218-267
    co = go.ContextOptions()
    assert (co.num_games, co.batchsize, co.T, co.job_id) == (1, 0, 1, "") and isinstance(co.mcts_options, _elf.TSOptions)
    w = go.WinRateStats()
    assert (w.black_wins, w.white_wins, w.total_games, w.sum_reward) == (0, 0, 0, 0.0)


def test_error_conventions(mods):
    _elf, go, goi = mods
    co, opt = go.ContextOptions(), go.GameOptions()
    co.batchsize, co.num_games = 16, 2
    co.mcts_options.num_threads = 1
    opt.mode = "bogus"
    with pytest.raises(ValueError):            # std::range_error, inference/game_context.h:38-40
        go.GameContext(co, opt)
    opt.mode = "selfplay"
    with pytest.raises(ValueError):            # the inference context "Only works for online setting"
        goi.GameContext(co, opt)
    opt.mode = "train"
    with pytest.raises(ValueError):            # training server: out of scope, said loudly
        go.GameContext(co, opt)
    opt.mode = "selfplay"
    co.mcts_options.num_threads, co.mcts_options.num_rollouts_per_batch = 64, 32
    with pytest.raises(ValueError):            # 64 x 32 leaves per step exceed the leaf table (1024): rejected, never truncated
        go.GameContext(co, opt)
    co.mcts_options.num_threads, co.mcts_options.num_rollouts_per_batch = 16, 8
    go.GameContext(co, opt)                    # 128 leaves per step are fine
    co.mcts_options.num_threads = 2
    co.mcts_options.pick_method = "softmax"
    with pytest.raises(ValueError):            # "MCTS Pick method unknown!" tree_search.h:521-524
        go.GameContext(co, opt)
    for m in ("strongest_prior", "uniform_random", "most_visited"):   # the three the reference knows (tree_search.h:506-519)
        co.mcts_options.pick_method = m
        go.GameContext(co, opt)
    opt.black_use_policy_network_only = True   # MCTSAI_T::actPolicyOnly is on this path
    go.GameContext(co, opt)
    opt.black_use_policy_network_only = False
    GC = go.GameContext(co, opt)
    assert GC.getParams() == {"num_action": 362, "board_size": 19, "num_future_actions": 3, "num_planes": 18, "our_stone_plane": 0,
                              "opponent_stone_plane": 1, "ACTION_SKIP": -100, "ACTION_PASS": -99, "ACTION_RESIGN": -98, "ACTION_CLEAR": -97}
    lim = GC.getLimits()      # this engine's own limits, outside the reference's getParams dictionary
    assert lim["max_rollouts_per_step"] == 1024 and lim["nodes_per_game"] % 64 == 0 and lim["nodes_per_game"] * 5888 < lim["tree_bytes_per_game_per_ai"] < lim["nodes_per_game"] * 6700   # 5888-B records + the big pool's share
    assert GC.getServer() is None and GC.getGame(7) is None
    ctx = GC.ctx()
    o = ctx.createSharedMemOptions("actor_black", 16)
    sm = ctx.allocateSharedMem(o, ["s", "pi", "no_such_key"])     # unknown keys are skipped with a warning (extractor.h:559-566)
    assert sm["no_such_key"] is None and sm["s"].field().sz().vec() == [16, 18, 19, 19]
    with pytest.raises(ValueError):
        sm["pi"].set(1 << 20, [4])                                  # one stride per dimension
    with pytest.raises(ValueError):
        sm["pi"].set(1 << 20, [362 * 4 - 4, 4])                     # smaller than contiguous (extractor.h:349-358)
    with pytest.raises(RuntimeError):
        ctx.wait()                                                  # before start()


class _Rec:
    """recording mock of what Allocator.spec2batches touches"""

    def __init__(self, log, fields):
        self.log, self.fields, self.n = log, fields, 0

    def createSharedMemOptions(self, name, bs):
        self.log.append(("createSharedMemOptions", name, bs))
        rec = self

        class O:
            def setTimeout(self, t):
                rec.log.append(("setTimeout", name, t))
        o = O()
        o.name, o.bs = name, bs
        return o

    def allocateSharedMem(self, opts, keys):
        idx = self.n
        self.n += 1
        self.log.append(("allocateSharedMem", opts.name, tuple(keys)))
        rec = self

        class F:
            def __init__(self, k):
                self.k = k

            def name(self):
                return self.k

            def type_name(self):
                return rec.fields[self.k][0]

            def sz(self):
                k = self.k

                class S:
                    def vec(self):
                        return list(rec.fields[k][1])
                return S()

        class P:
            def __init__(self, k):
                self.k = k

            def field(self):
                return F(self.k)

            def set(self, addr, strides):
                rec.log.append(("set", opts.name, self.k, tuple(int(s) for s in strides)))

        class SM:
            def __getitem__(self, k):
                return P(k)

            def getSharedMemOptions(self):
                class OO:
                    def idx(self):
                        return idx
                return OO()
        return SM()


def test_restated_wrapper_makes_the_reference_s_calls():
    """tests/gcwrapper_restated.py against the reference's unmodified utils_elf.py on a recording mock: same calls, same order,
    same arguments, same resulting batch layout (skipped where /root/reference is absent, i.e. on the GPU box)."""
    ref = load_reference_utils()
    if ref is None:
        pytest.skip("/root/reference not present")
    import gcwrapper_restated as mine
    fields = {"s": ("float", [16, 18, 19, 19]), "pi": ("float", [16, 362]), "V": ("float", [16]), "a": ("int64_t", [16]), "rv": ("int64_t", [16]),
              "black_ver": ("int64_t", [16]), "white_ver": ("int64_t", [16])}
    out = []
    for mod in (ref, mine):
        log = []
        ctx = _Rec(log, fields)

        class GC:
            def ctx(self_inner):
                return ctx
        if mod is ref:
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):          # Allocator prints every field
                w = mod.GCWrapper(GC(), 16, game_py_desc(16), num_recv=2, gpu=None, use_numpy=False, params={})
        else:
            w = mod.GCWrapper(GC(), 16, game_py_desc(16), num_recv=2, gpu=None, params={})
        layout = [({k: (tuple(t.shape), str(t.dtype)) for k, t in b["input"].items()}, {k: (tuple(t.shape), str(t.dtype)) for k, t in b["reply"].items()})
                  for b in w.batches]
        out.append((log, layout, dict(w.name2idx), dict(w.idx2name)))
        assert w.reg_callback("actor_black", lambda b: None) and not w.reg_callback_if_exists("nope", None)
        with pytest.raises(ValueError):
            w.reg_callback("nope", None)
        with pytest.raises(ValueError):
            w.start()                                                # callbacks missing (utils_elf.py:416-424)
    assert out[0] == out[1]


def test_reference_gcwrapper_allocates_against_the_context(mods):
    """The reference's unmodified GCWrapper.__init__ (Allocator.spec2batches) against the real pybind Context: every tensor it
    allocates gets registered (address + byte strides), indices and labels come back as the reference's would."""
    ref = load_reference_utils()
    if ref is None:
        pytest.skip("/root/reference not present")
    _elf, go, _ = mods
    import contextlib
    import io
    co, opt = go.ContextOptions(), go.GameOptions()
    co.num_games, co.batchsize, opt.mode = 8, 128, "selfplay"
    co.mcts_options.num_threads, co.mcts_options.num_rollouts_per_batch = 1, 16
    GC = go.GameContext(co, opt)
    with contextlib.redirect_stdout(io.StringIO()):
        w = ref.GCWrapper(GC, 128, game_py_desc(128), num_recv=2, gpu=None, use_numpy=False, params=GC.getParams())
    assert len(w.batches) == 8 and w.name2idx["actor_black"] == [0, 1] and w.name2idx["game_start"] == [6, 7] and w.idx2name[4] == "game_end"
    assert tuple(w.batches[0]["input"]["s"].shape) == (128, 18, 19, 19) and tuple(w.batches[1]["reply"]["pi"].shape) == (128, 362)
    assert w.batches[6]["input"]["black_ver"].dtype.is_floating_point is False and w.batches[4]["input"] == {}
    with pytest.raises(ValueError):
        w.start()


# ---------------------------------------------------------------------------------------------------------------- GPU
def _wrapper_module():
    ref = load_reference_utils()
    if ref is not None:
        return ref, True
    import gcwrapper_restated
    return gcwrapper_restated, False


def _session(mods, cfg, m, n=19, device_resident=False, force_restated=False, batchsize=None, rv_value=None, games=None, **over):
    """One GCWrapper session shaped like scripts/elfgames/go/selfplay.py:115-199; returns (search log, events)."""
    import contextlib
    import io
    import torch
    from pyoracle import stub_net
    _elf, go, _ = mods
    cfg = dict(cfg)
    if games:
        cfg["num_games"] = games
    if batchsize:
        cfg["batchsize"] = batchsize
    co, opt = options_from_cfg(go, cfg, n=n, log_searches=over.pop("log_searches", m), **over)
    GC = go.GameContext(co, opt)
    mod, is_ref = _wrapper_module()
    if device_resident or force_restated:
        import gcwrapper_restated as mod
        is_ref = False
    kw = dict(num_recv=2, gpu=0, params=GC.getParams())
    if not is_ref:
        kw["device_resident"] = device_resident
    else:
        kw["use_numpy"] = False
    with contextlib.redirect_stdout(io.StringIO()):
        gcw = mod.GCWrapper(GC, co.batchsize, game_py_desc(co.batchsize), **kw)
    ev = dict(rows=[], starts=[], ends=0, kinds=set(), is_ref=is_ref, white_rows=0)
    ties = int(cfg["net_tie_levels"])
    bv, wv = int(cfg.get("black_ver", 0)), int(cfg.get("white_ver", -1))

    def make_actor(salt, ver, white):
        def actor(batch):                              # Evaluator.actor-shaped: reply keys = the group's reply list
            s = batch["s"]
            ev["rows"].append(batch.batchsize)
            ev["white_rows"] += batch.batchsize if white else 0
            ev["kinds"].add(s.device.type)
            pi, v = stub_net(n, s.cpu().numpy(), salt, ties)
            k = s.shape[0]
            return dict(pi=torch.from_numpy(pi).cuda(), V=torch.from_numpy(v).cuda(), a=torch.zeros(k, dtype=torch.int64).cuda(),
                        rv=torch.full((k,), ver if rv_value is None else rv_value, dtype=torch.int64).cuda())
        return actor

    def game_start(batch):
        ev["starts"].append((int(batch["black_ver"][0]), int(batch["white_ver"][0])))

    def game_end(batch):
        ev["ends"] += 1

    gcw.reg_callback("actor_black", make_actor(int(cfg["net_salt"]), bv, False))
    gcw.reg_callback("actor_white", make_actor(int(cfg.get("white_net_salt", cfg["net_salt"])), wv, True))
    gcw.reg_callback_if_exists("game_start", game_start)
    gcw.reg_callback_if_exists("game_end", game_end)
    gcw.start()
    # Client::setRequest(black_ver, white_ver, thres, numThreads); the fixtures' harness sent num_game_thread_used = num_games
    if int(cfg.get("player_swap", 0)):   # ClientCtrl.player_swap: the optional fifth argument (the reference's server sends it in its MsgRequest)
        GC.getClient().setRequest(bv, wv, float(np.float32(cfg["resign_thres"])), int(cfg.get("thread_used", 0)) or -1, True)
    else:
        GC.getClient().setRequest(bv, wv, float(np.float32(cfg["resign_thres"])), int(cfg.get("thread_used", 0)) or -1)
    guard = 0
    while len(GC.ctx().searchLog()) < m:
        gcw.run()
        guard += 1
        assert guard < 200000
    log = GC.ctx().searchLog()
    ev["GC"], ev["gcw"] = GC, gcw
    return log, ev


def _check_fixture(log, g, m):
    for i in range(m):
        game, move, best, total, ne, coord, visits, reward = log[i]
        assert ne == int(g["n_edges"][i]), i
        assert coord == [int(c) for c in g["coord"][i, :ne]], "search %d: edge order" % i
        assert visits == [int(c) for c in g["visits"][i, :ne]], "search %d: visit counts" % i
        assert np.array_equal(np.array(reward, np.float32).view(np.uint32), g["reward"][i, :ne].view(np.uint32)), "search %d: rewards" % i
        want_total = int(g["total_visits"][i]) if "total_visits" in g.files else int(g["visits"][i, :ne].sum())
        assert (move, best, total) == (int(g["move_played"][i]), int(g["best_action"][i]), want_total), i


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_gcwrapper_session_reproduces_the_reference_fixture(mods, device_resident):
    """the reference's GCWrapper (unmodified file where /root/reference exists, its checked restatement on the
    GPU box) over the pybind boundary with a game.py-shaped desc reproduces mcts_19_r256_dir bit for bit -- batch tensors in pinned
    host memory (Allocator._alloc with gpu=0), and device-resident."""
    g = np.load(os.path.join(GOLDEN, "mcts_19_r256_dir.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    m = 10
    log, ev = _session(mods, cfg, m, device_resident=device_resident)
    _check_fixture(log, g, m)
    assert ev["starts"] == [(0, -1)]                       # one game_start batch, from the first request
    assert max(ev["rows"]) <= 16 and ev["kinds"] == {"cuda"}
    GC = ev["GC"]
    game = GC.getGame(0)                                   # GoGameSelfPlay accessors (inference/Pybind.cc:39-44)
    assert game.getNextPlayer() == ("B" if m % 2 == 0 else "W")
    last = int(g["move_played"][m - 1])
    x, y = last % 21 - 1, last // 21 - 1
    assert game.getLastMove() == chr(ord("A") + (x + 1 if x >= 8 else x)) + str(y + 1)
    sb = game.showBoard()
    # m stones minus captures, the last move marked "X)" / "O)" once; the two caption lines carry "(X)" and "(O)" themselves
    assert sb.count("X ") + sb.count("O ") + 1 >= m - 2 and sb.count("X)") + sb.count("O)") == 3 and "has captured" in sb
    assert isinstance(game.getScore(), float) and game.getLastScore() == 0.0
    assert GC.getClient().getGameStats().getWinRateStats().total_games == 0
    ev["gcw"].stop()


@pytest.mark.gpu
def test_steps_are_served_in_chunks_of_the_group_batchsize(mods, built):
    """4 games x 16 rollouts = up to 64 leaves per device step, handed out as wait()/step() rounds of at most batchsize 16 (two
    alternating SharedMem buffers, num_recv = 2): the same searches as the direct elfsp_* loop over the same games."""
    import torch
    import elf_amd
    from pyoracle import MCTS_DEFAULTS, stub_net
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=4, rollouts_per_thread=64, seed=321, net_salt=12, policy_distri_cutoff=4, max_searches=12)
    m = 12
    log, ev = _session(mods, cfg, m, force_restated=True)
    assert max(ev["rows"]) <= 16 and len(ev["rows"]) > 4 * (64 // 16) * 2
    sp = elf_amd.SelfPlay(board_size=19, num_games=4, mcts_rollout_per_thread=64, mcts_rollout_per_batch=16, mcts_puct=cfg["c_puct"],
                          mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=cfg["root_epsilon"], mcts_alpha=cfg["root_alpha"],
                          komi=7.5, policy_distri_cutoff=4, seed=321, log_searches=m)
    while sp.stats()["logged"] < m:
        rows = sp.begin_step()
        pi, v = stub_net(19, sp.s[:rows].cpu().numpy(), 12, 0)
        sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
    rec, coord, visits, _, reward = sp.search_log()
    for i in range(m):
        ne = rec[i].n_edges
        assert log[i][:5] == (rec[i].game, rec[i].move_played, rec[i].best_action, rec[i].total_visits, ne)
        assert log[i][5] == coord[i, :ne].tolist() and log[i][6] == visits[i, :ne].tolist()
    sp.close()
    ev["gcw"].stop()


@pytest.mark.gpu
def test_reply_version_is_checked(mods):
    """go/mcts/mcts.h:209-217: a reply whose rv differs from the requested model version is an error (std::runtime_error)"""
    from pyoracle import MCTS_DEFAULTS
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(rollouts_per_thread=32, seed=5, net_salt=1)
    with pytest.raises(RuntimeError) as e:
        _session(mods, cfg, 2, rv_value=7, force_restated=True)
    assert "version" in str(e.value)


@pytest.mark.gpu
def test_game_end_batches_and_game_stats(mods, tmp_path):
    """finished games: one game_end batch each (GameNotifier::OnGameEnd, distri_client.h:228-240), WinRateStats fed, records kept;
    GameOptions.dump_record_prefix: every finished game is written as SGF (finish_game :133-135 -> GoStateExt::dumpSgf)"""
    import json
    from pyoracle import MCTS_DEFAULTS
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=3, rollouts_per_thread=16, seed=88, net_salt=3, move_cutoff=5, policy_distri_cutoff=2)
    prefix = str(tmp_path / "dump")
    log, ev = _session(mods, cfg, 3 * 9, n=9, force_restated=True, keep_records=16, dump_record_prefix=prefix)
    GC = ev["GC"]
    wr = GC.getClient().getGameStats().getWinRateStats()
    assert wr.total_games >= 3 and wr.total_games == wr.black_wins + wr.white_wins
    for _ in range(8):                     # drain pending game_end batches
        if ev["ends"] == wr.total_games:
            break
        ev["gcw"].run()
    assert ev["ends"] == GC.getClient().getGameStats().getWinRateStats().total_games
    recs = GC.popRecords()
    assert len(recs) >= 3
    j = json.loads(recs[0])
    assert j["result"]["num_move"] == 4 and j["request"]["vers"]["black_ver"] == 0 and j["request"]["vers"]["mcts_opt"]["num_rollouts_per_thread"] == 16
    assert GC.getClient().getGameStats().getPlayedGames() == []
    import elf_amd
    from elf_amd.train import sgf_file_name
    files = sorted(os.listdir(str(tmp_path)))
    assert len(files) >= len(recs) >= 3
    for t in recs:
        name = sgf_file_name(prefix, t)                         # <prefix>_<game>_<seq>_<B|W>.sgf
        text = open(name).read()
        jr = json.loads(t)
        assert text.startswith("(;SZ[9]RE[") and ("Filename: " + name) in text and "PB[MCTS]PW[MCTS]KM[7.5]" in text and text.endswith(")\n")
        assert text.count(";B[") + text.count(";W[") == jr["result"]["num_move"]
        assert ("C[1: PredV: %f]" % jr["result"]["values"][0]) in text
    ev["gcw"].stop()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["online_9_script", "online_9_following_pass", "online_9_not_following"])
def test_online_mode_equals_reference(mods, name):
    """mode online against the REAL reference (oracle/ref_selfplay.cc online = 1: the reference's GoGameSelfPlay with a scripted
    "human_actor"): the same answers -- moves, an illegal move, SKIP (the AI searches and moves), PASS, CLEAR, RESIGN -- give the
    same prompts (all 18 feature planes of every prompt), the same searches (root edges, visits, rewards, moves) and the same
    Record texts; GameOptions.following_pass with a net that is sure of the result: the AI answers the human's pass with a pass
    exactly where the reference does (mcts_update_info, game_selfplay.cc:104-111)."""
    import contextlib
    import io
    import json
    import torch
    import gcwrapper_restated as mod
    from pyoracle import stub_net
    _elf, _, goi = mods
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    script = [int(a) for a in g["human_script"]]
    want_prompts = np.unpackbits(g["prompts"], axis=1)[:, :18 * n * n].reshape(-1, 18, n, n)
    want_records = [json.loads(str(t)) for t in g["records"]]
    m = int(g["searches"])
    co, opt = options_from_cfg(goi, cfg, n=n, log_searches=m + 4, keep_records=8, following_pass=bool(cfg["following_pass"]))
    opt.mode = "online"
    GC = goi.GameContext(co, opt)
    params = GC.getParams()
    assert (params["ACTION_SKIP"], params["ACTION_PASS"], params["ACTION_RESIGN"], params["ACTION_CLEAR"]) == (-100, -99, -98, -97)
    desc = {"human_actor": dict(input=["s"], reply=["pi", "a", "V"], batchsize=1),
            "actor_black": dict(input=["s"], reply=["pi", "V", "a", "rv"], timeout_usec=10, batchsize=co.batchsize)}
    with contextlib.redirect_stdout(io.StringIO()):
        gcw = mod.GCWrapper(GC, co.batchsize, desc, num_recv=2, gpu=0, params=params)
    prompts, records = [], []

    def human(batch):
        prompts.append(batch["s"].cpu().numpy()[0].copy())
        a = script[len(prompts) - 1] if len(prompts) <= len(script) else params["ACTION_SKIP"]
        return dict(pi=torch.zeros(1, n * n + 1).cuda(), V=torch.zeros(1).cuda(), a=torch.tensor([a], dtype=torch.int64).cuda())

    def actor(batch):
        s = batch["s"]
        pi, v = stub_net(n, s.cpu().numpy(), int(cfg["net_salt"]), int(cfg["net_tie_levels"]))
        if int(cfg["net_value_on"]):
            v[:] = np.float32(cfg["net_value"])
        k = s.shape[0]
        return dict(pi=torch.from_numpy(pi).cuda(), V=torch.from_numpy(v).cuda(), a=torch.zeros(k, dtype=torch.int64).cuda(),
                    rv=torch.zeros(k, dtype=torch.int64).cuda())

    gcw.reg_callback("human_actor", human)
    gcw.reg_callback("actor_black", actor)
    gcw.start()
    GC.setRequest(0, -1, 0.0, 1)             # numThreads as the harness sent it (it shows in Record.request)
    guard = 0
    while len(prompts) < len(script) + 1:
        gcw.run()
        records += GC.popRecords()
        guard += 1
        assert guard < 20000
    assert len(prompts) == len(want_prompts)
    for i, (p, w) in enumerate(zip(prompts, want_prompts)):
        assert np.array_equal(p != 0, w != 0), "%s: prompt %d" % (name, i)
    _check_fixture(GC.ctx().searchLog(), g, m)
    assert len(records) == len(want_records)
    for t, w in zip(records, want_records):
        j = json.loads(t)
        j["timestamp"] = w["timestamp"]
        for part in ("client_ctrl", "vers"):
            assert j["request"][part] == w["request"][part], (name, w["seq"], part, j["request"][part], w["request"][part])
        assert j == w, (name, w["seq"])
    gcw.stop()


@pytest.mark.gpu
def test_online_mode_human_actor(mods):
    """_elfgames_go_inference.GameContext (mode online, game.py:366-378 desc): the human_actor prompt of GoGameSelfPlay::act
    (game_selfplay.cc:290-330) -- a move, an illegal move (prompted again), SKIP (the AI searches and moves), CLEAR."""
    import contextlib
    import io
    import torch
    import gcwrapper_restated as mod
    from pyoracle import Port, stub_net
    _elf, _, goi = mods
    n = 9
    co, opt = goi.ContextOptions(), goi.GameOptions()
    co.num_games, co.batchsize = 1, 8
    ts = co.mcts_options
    ts.num_threads, ts.num_rollouts_per_thread, ts.num_rollouts_per_batch, ts.persistent_tree, ts.virtual_loss = 1, 32, 8, True, 1
    ts.alg_opt.c_puct = 1.5
    opt.mode, opt.board_size, opt.seed, opt.use_mcts = "online", n, 11, True
    GC = goi.GameContext(co, opt)
    desc = {"human_actor": dict(input=["s"], reply=["pi", "a", "V"], batchsize=1),
            "actor_black": dict(input=["s"], reply=["pi", "V", "a", "rv"], timeout_usec=10, batchsize=8)}
    with contextlib.redirect_stdout(io.StringIO()):
        gcw = mod.GCWrapper(GC, co.batchsize, desc, num_recv=2, gpu=0, params=GC.getParams())
    params = GC.getParams()
    port = Port(n)
    st = port.new()
    S = n + 2
    script = [3 * n + 3, 3 * n + 3, params["ACTION_SKIP"], 5 * n + 5, params["ACTION_SKIP"], params["ACTION_CLEAR"], 2 * n + 2]
    prompts = []

    def human(batch):
        prompts.append(batch["s"].cpu().numpy()[0].copy())
        a = script[len(prompts) - 1]
        return dict(pi=torch.zeros(1, n * n + 1).cuda(), V=torch.zeros(1).cuda(), a=torch.tensor([a], dtype=torch.int64).cuda())

    def actor(batch):
        s = batch["s"]
        pi, v = stub_net(n, s.cpu().numpy(), 9, 0)
        k = s.shape[0]
        return dict(pi=torch.from_numpy(pi).cuda(), V=torch.from_numpy(v).cuda(), a=torch.zeros(k, dtype=torch.int64).cuda(),
                    rv=torch.zeros(k, dtype=torch.int64).cuda())

    gcw.reg_callback("human_actor", human)
    gcw.reg_callback("actor_black", actor)
    gcw.start()
    GC.setRequest(0, -1, 0.0, -1)
    game = GC.getGame(0)
    guard = 0
    while len(prompts) < len(script):
        gcw.run()
        guard += 1
        assert guard < 5000
    # prompt 0: empty board; the move D4 (action 3*9+3) is played; prompt 1 shows it and repeats the action -> illegal -> prompt 2
    assert prompts[0][:16].sum() == 0 and prompts[0][16].all()
    c = (3 + 1) * S + (3 + 1)
    assert port.forward(st, c) == 1
    assert np.array_equal(prompts[1], port.extract_agz(st, 0)) and np.array_equal(prompts[2], prompts[1])
    # prompt 2 answered SKIP: the AI (White) searched and moved; prompt 3 shows two stones, Black to move
    assert prompts[3][:2].sum() == 2 and prompts[3][16].all()
    # prompt 3: F6; prompt 4 SKIP -> AI; prompt 5 CLEAR -> board empty at prompt 6
    assert prompts[5][:2].sum() == 4 and prompts[6][:16].sum() == 0
    assert game.getNextPlayer() == "W" and game.getLastMove() == "C3"        # prompt 6 played action 2*9+2
    assert GC.ctx().version().startswith("elf_amd")
    gcw.stop()


@pytest.mark.gpu
def test_new_request_is_looked_at_every_fifth_act(mods):
    """Client::setRequest with a new model version while the games play (GoGameSelfPlay::OnReceive, game_selfplay.cc:222-270): a
    game looks at its mailbox at the top of every fifth act (`_online_counter % 5 == 0`, :273-289), so the request sent during the
    4th search is received before the 6th; until then searches run -- and replies are checked -- under the old version.  Then
    every game restarts from the empty board, one game_start batch carries the new versions, and from then on replies must carry
    the new version in rv."""
    import contextlib
    import io
    import torch
    import gcwrapper_restated as mod
    from pyoracle import MCTS_DEFAULTS, stub_net
    _elf, go, _ = mods
    n = 9
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=2, rollouts_per_thread=32, seed=66, net_salt=2, policy_distri_cutoff=4)
    co, opt = options_from_cfg(go, cfg, n=n, log_searches=64)
    GC = go.GameContext(co, opt)
    with contextlib.redirect_stdout(io.StringIO()):
        gcw = mod.GCWrapper(GC, co.batchsize, game_py_desc(co.batchsize), num_recv=2, gpu=0, params=GC.getParams())
    state = dict(rv=0, starts=[], rows=0)

    def actor(batch):
        s = batch["s"]
        k = s.shape[0]
        state["rows"] += k
        pi, v = stub_net(n, s.cpu().numpy(), 2, 0)
        return dict(pi=torch.from_numpy(pi).cuda(), V=torch.from_numpy(v).cuda(), a=torch.zeros(k, dtype=torch.int64).cuda(),
                    rv=torch.full((k,), state["rv"], dtype=torch.int64).cuda())

    gcw.reg_callback("actor_black", actor)
    gcw.reg_callback("actor_white", actor)
    gcw.reg_callback("game_start", lambda b: state["starts"].append((int(b["black_ver"][0]), int(b["white_ver"][0]))))
    gcw.reg_callback("game_end", lambda b: None)
    gcw.start()
    client = GC.getClient()
    client.setRequest(0, -1, 0.0, -1)
    while len(GC.ctx().searchLog()) < 2 * 3:          # three moves per game under version 0
        gcw.run()
    assert state["starts"] == [(0, -1)] and GC.getGame(0).getNextPlayer() == "W"
    gcw.run()                                          # a batch of the 4th search: that search is now open
    client.setRequest(7, -1, 0.0, -1)                  # new model while the search runs
    while len(GC.ctx().searchLog()) < 2 * 5:           # acts 4 and 5 of both games still run under version 0 (rv = 0 accepted)
        gcw.run()
    assert state["starts"] == [(0, -1)] and GC.getGame(0).getNextPlayer() == "W"
    state["rv"] = 7                                    # the 6th act begins with the mailbox: restart under version 7
    gcw.run()                                          # the game_start batch comes before any row of the restarted games
    assert state["starts"] == [(0, -1), (7, -1)]
    assert GC.getGame(0).getNextPlayer() == "B" and GC.getGame(1).getNextPlayer() == "B"    # both restarted from the empty board
    for _ in range(3):
        gcw.run()
    state["rv"] = 0                                    # the old model's replies are now refused
    with pytest.raises(RuntimeError):
        for _ in range(8):
            gcw.run()
    gcw.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mcts_9_eval_two_ai", "mcts_19_eval_swap"])
def test_gcwrapper_session_plays_evaluation_games_with_two_ais(mods, name):
    """Client.setRequest(black_ver, white_ver >= 0, ...): the second MCTSGoAI's leaves arrive in the "actor_white" group and are
    answered by another model (its own version in rv); the session reproduces the reference's fixture bit for bit through the
    pybind boundary with the restated GCWrapper (game.py-shaped desc, num_recv = 2)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    n = int(g["board_size"])
    m = min(24, len(g["move_played"]))
    log, ev = _session(mods, cfg, m, n=n, force_restated=True)
    _check_fixture(log, g, m)
    assert ev["starts"] == [(int(cfg["black_ver"]), int(cfg["white_ver"]))]
    assert ev["white_rows"] > 0 and ev["white_rows"] < sum(ev["rows"])
    ev["gcw"].stop()


@pytest.mark.gpu
def test_idle_game_threads_and_wait_requests(mods):
    """setRequest(..., numThreads = k): games k.. receive the request as a wait request (DispatcherCallback::OnFirstSend,
    common/dispatcher_callback.h:28-44) and stay idle; a later request with more threads starts them from the empty board; a
    wait request (black_ver < 0) idles every game at its next mailbox look."""
    from pyoracle import MCTS_DEFAULTS
    cfg = dict(MCTS_DEFAULTS)
    cfg.update(num_games=3, rollouts_per_thread=32, seed=11, net_salt=4, thread_used=2)
    log, ev = _session(mods, cfg, 8, n=9, force_restated=True, log_searches=256)
    assert sorted(set(r[0] for r in log)) == [0, 1]                     # game 2 never searched
    GC, gcw = ev["GC"], ev["gcw"]
    assert ev["starts"] == [(0, -1)]
    GC.getClient().setRequest(0, -1, 0.0, 3)                            # all three threads: game 2 starts now (it was waiting)
    while len([r for r in GC.ctx().searchLog() if r[0] == 2]) < 2:
        gcw.run()
    assert ev["starts"] == [(0, -1), (0, -1)]                           # the waiting game's start is a game_start batch
    GC.getClient().setRequest(-1, -1, 0.0, -1)                          # [wait]
    with pytest.raises(RuntimeError) as e:                              # every game idles within five acts: nothing left to serve
        for _ in range(400):
            gcw.run()
    assert "waiting" in str(e.value)
    assert GC.ctx().wait(10) is None                                    # with a timeout: None, as the reference's wait(timeout) does
    gcw.stop()
