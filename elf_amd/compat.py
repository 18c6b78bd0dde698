"""Drop-in Python surface with the reference's own names, over the device-resident engine.

Mirrors what scripts written against the reference import and call:
  * `TSOptions`, `SearchAlgoOptions`           src_cpp/elf/ai/tree_search/tree_search_options.h:23-229 (fields :70-74,:215-228)
  * `ContextOptions`                           src_cpp/elf/legacy/python_options_utils_cpp.h:19-47
  * `GameOptions` (self-play fields)           src_cpp/elfgames/go/common/go_game_specific.h:16-268
  * `GameContext(co, opt)` .ctx() .getParams() .setRequest() .getClient()/getGameStats()
                                               src_cpp/elfgames/go/train/game_context.h:33-133, train/Pybind.cc:18-63
  * `GCWrapper(GC, batchsize, spec, ...)` .reg_callback() .reg_callback_if_exists() .start() .run() .stop(), `Batch`
                                               src_py/elf/utils_elf.py:112-290,291-437
so that `scripts/elfgames/go/selfplay.py:87-199` style code runs unchanged in structure:

    co, opt = ContextOptions(), GameOptions(); ...; GC = GameContext(co, opt)
    gcw = GCWrapper(GC, co.batchsize, desc, num_recv=2, gpu=0, params=GC.getParams())
    gcw.reg_callback("actor_black", evaluator.actor); gcw.start(); GC.getClient().setRequest(0, -1, thres, -1)
    while not done: gcw.run()

Differences are the ones listed in INTEGRATION.md: batch tensors already live in HBM (`cpu2gpu` is the identity),
a batch carries the leaves of every game of the process, and one search thread per game (`num_threads` is accepted
and must be 1 for determinism-equivalent results; larger values are served by more games per GPU instead).
"""
import torch

from .selfplay import SelfPlay


class SearchAlgoOptions:
    def __init__(self):
        self.use_prior = True
        self.c_puct = 5.0
        self.unexplored_q_zero = False
        self.root_unexplored_q_zero = False


class TSOptions:
    def __init__(self):
        self.max_num_moves = 0
        self.num_threads = 16
        self.num_rollouts_per_thread = 100
        self.num_rollouts_per_batch = 8
        self.verbose = False
        self.verbose_time = False
        self.seed = 0
        self.persistent_tree = False
        self.root_epsilon = 0.0
        self.root_alpha = 0.0
        self.log_prefix = ""
        self.pick_method = "most_visited"
        self.alg_opt = SearchAlgoOptions()
        self.virtual_loss = 0

    def info(self, verbose=False):
        return "[#th=%d][rl=%d][per=%d][eps=%g][alpha=%g][prior=%d][c_puct=%g]" % (
            self.num_threads, self.num_rollouts_per_thread, self.persistent_tree, self.root_epsilon, self.root_alpha,
            self.alg_opt.use_prior, self.alg_opt.c_puct)


class ContextOptions:
    def __init__(self):
        self.num_games = 1
        self.batchsize = 0
        self.T = 1
        self.job_id = ""
        self.mcts_options = TSOptions()

    def print(self):
        print("JobId: %s\n#Game: %d\nT: %d\n%s" % (self.job_id, self.num_games, self.T, self.mcts_options.info()))


class GameOptions:
    """The self-play subset of go_game_specific.h:16-131 (same names, same defaults)."""

    def __init__(self):
        self.seed = 0
        self.num_future_actions = 3
        self.mode = "selfplay"
        self.use_mcts = False
        self.use_mcts_ai2 = False
        self.move_cutoff = -1
        self.policy_distri_cutoff = 20
        self.policy_distri_training_for_all = False
        self.resign_thres = 0.05
        self.resign_prob_never = 0.1
        self.komi = 7.5
        self.ply_pass_enabled = 0
        self.white_puct = -1.0
        self.verbose = False
        self.preload_sgf = ""
        self.preload_sgf_move_to = -1
        self.use_df_feature = False
        self.board_size = 19          # the reference fixes this at compile time (BOARD9x9)
        self.gpu = 0                  # device that owns the boards and trees
        self.nodes_per_game = None
        self.log_searches = 0         # keep the first N search results (SelfPlay.search_log), for tests

    def info(self):
        return "Seed: %d\nMode: %s\nKomi: %g\nply_pass_enabled: %d" % (self.seed, self.mode, self.komi, self.ply_pass_enabled)


class GameStats:
    """GameStats / getGameStats (common/game_stats.h:19-68): only the counters this engine maintains."""

    def __init__(self, gc):
        self._gc = gc

    def getWinRateStats(self):
        st = self._gc._sp.stats()
        class _W:   # noqa: E306
            total_games = st["games"]
        return _W()


class _Client:
    def __init__(self, gc):
        self._gc = gc

    def setRequest(self, black_ver, white_ver, thres, numThreads=-1):
        self._gc.setRequest(black_ver, white_ver, thres, numThreads)

    def getGameStats(self):
        return GameStats(self._gc)


class _SharedMemOptions:
    def __init__(self, idx, label, batchsize):
        self._idx, self._label, self._bs = idx, label, batchsize

    def idx(self):
        return self._idx

    def label(self):
        return self._label

    def batchsize(self):
        return self._bs


class _SharedMem:
    def __init__(self, opts, rows):
        self._o, self._rows = opts, rows

    def getSharedMemOptions(self):
        return self._o

    def effective_batchsize(self):
        return self._rows


class _Context:
    """The two calls of elf::Context the Python side uses (Pybind.cc:27-89): wait() / step()."""

    def __init__(self, gc):
        self._gc = gc

    def version(self):
        return "elf_amd"

    def start(self):
        self._gc._started = True

    def stop(self):
        self._gc._started = False

    def wait(self, timeout_usec=0):
        self._gc._build()
        rows = self._gc._sp.begin_step()
        return _SharedMem(_SharedMemOptions(0, "actor_black", self._gc._sp.max_rows), rows)

    def step(self, success=0):
        sp = self._gc._sp
        rep = self._gc._reply
        sp.end_step(rep.get("pi"), rep.get("V"))
        self._gc._reply = {}


class GameContext:
    def __init__(self, co, opt):
        if opt.mode not in ("selfplay", "online"):
            raise ValueError("options.mode not recognized! " + str(opt.mode))   # inference/game_context.h:38-40
        self.co, self.opt = co, opt
        self._sp = None
        self._reply = {}
        self._started = False
        self._resign = 0.0

    def _build(self):
        if self._sp is not None:
            return
        co, opt, ts = self.co, self.opt, self.co.mcts_options
        self._sp = SelfPlay(
            board_size=opt.board_size, num_games=co.num_games, device=opt.gpu, mcts_rollout_per_thread=ts.num_rollouts_per_thread,
            mcts_rollout_per_batch=ts.num_rollouts_per_batch, mcts_puct=ts.alg_opt.c_puct, mcts_virtual_loss=ts.virtual_loss,
            mcts_use_prior=ts.alg_opt.use_prior, mcts_persistent_tree=ts.persistent_tree, mcts_epsilon=ts.root_epsilon,
            mcts_alpha=ts.root_alpha, mcts_unexplored_q_zero=ts.alg_opt.unexplored_q_zero,
            mcts_root_unexplored_q_zero=ts.alg_opt.root_unexplored_q_zero, komi=opt.komi, ply_pass_enabled=opt.ply_pass_enabled,
            policy_distri_cutoff=opt.policy_distri_cutoff, move_cutoff=opt.move_cutoff, resign_thres=self._resign,
            never_resign_prob=0.0, seed=opt.seed, nodes_per_game=opt.nodes_per_game, log_searches=opt.log_searches)
        if opt.preload_sgf:
            # GoGameSelfPlay::restart (game_selfplay.cc:202-219): follow the main line of the SGF while playing
            self._sp.preload(sgf_main_line(opt.preload_sgf, opt.board_size), opt.preload_sgf_move_to)

    def ctx(self):
        return _Context(self)   # the device engine is built at the first wait(): setRequest() may still arrive after start()

    def getParams(self):
        n = self.opt.board_size   # GoFeature::getParams, common/game_feature.h:208-221
        return {"num_action": n * n + 1, "board_size": n, "num_future_actions": self.opt.num_future_actions, "num_planes": 18,
                "our_stone_plane": 0, "opponent_stone_plane": 1, "ACTION_SKIP": -100, "ACTION_PASS": -99, "ACTION_RESIGN": -98,
                "ACTION_CLEAR": -97}

    def setRequest(self, black_ver, white_ver, thres, numThreads=-1):
        if self._sp is not None and thres != self._resign:
            raise ValueError("setRequest after start cannot change the resign threshold")
        self._resign = float(thres)

    def getClient(self):
        return _Client(self)

    def getGame(self, game_idx):
        """GoGameSelfPlay accessors the console uses (common/game_selfplay.h:41-56, inference/Pybind.cc:31-45)"""
        self._build()
        return _GameView(self, int(game_idx))


def sgf_main_line(path, board_size):
    """Main-line moves of an SGF file as reference Coords (what Sgf::load + SgfIterator yield, sgf/sgf.cc; "" or "tt" = pass)"""
    import re
    txt = open(path).read()
    S = board_size + 2
    out = []
    for _, mv in re.findall(r";\s*([BW])\s*\[([a-z]{0,2})\]", txt):
        if len(mv) < 2 or (mv == "tt" and board_size <= 19):
            out.append(0)
            continue
        x, y = ord(mv[0]) - 97, ord(mv[1]) - 97
        if not (0 <= x < board_size and 0 <= y < board_size):
            raise ValueError("SGF move off board: " + mv)
        out.append((y + 1) * S + (x + 1))
    return out


class _GameView:
    def __init__(self, gc, g):
        self._gc, self._g = gc, g

    def _boards(self):
        return self._gc._sp.board_engine()

    def _info(self):
        return self._boards().info_host(ids=[self._g])

    def getNextPlayer(self):
        return "B" if int(self._info()["next_player"][0]) == 1 else "W"      # player2str, sgf/sgf.h:59-72

    def getLastMove(self):
        """coord2str2 (sgf/sgf.h:74-85): "PASS", "RESIGN" or letter (no I) + row"""
        c = int(self._info()["last_move"][0])
        if c == 0:
            return "PASS"
        if c == 1:
            return "RESIGN"
        S = self._gc.opt.board_size + 2
        x, y = c % S - 1, c // S - 1
        if x >= 8:
            x += 1
        return chr(ord("A") + x) + str(y + 1)

    def getScore(self):
        return float(self._boards().evaluate(ids=[self._g], komi=self._gc.opt.komi).cpu()[0])

    def getLastScore(self):
        return float(self._gc._sp.last_score()[self._g])

    def showBoard(self):
        n = self._gc.opt.board_size
        col, _ = self._boards().export_board(ids=[self._g])
        col = col.cpu().numpy()[0].reshape(n, n)
        rows = []
        for y in range(n - 1, -1, -1):
            rows.append("%2d " % (y + 1) + " ".join(".XO"[int(col[x, y])] for x in range(n)))
        rows.append("   " + " ".join(chr(ord("A") + (x + 1 if x >= 8 else x)) for x in range(n)))
        return "\n".join(rows)


class Batch:
    """src_py/elf/utils_elf.py:112-290 (the tensors already live on the GPU)."""

    def __init__(self, _GC=None, _batchdim=0, _histdim=None, **kwargs):
        self.GC, self.batchdim, self.histdim = _GC, _batchdim, _histdim
        self.batch = kwargs

    def empty_copy(self):
        return Batch(self.GC, self.batchdim, self.histdim)

    def first_k(self, batchsize):
        b = self.empty_copy()
        b.batch = {k: v[:batchsize] for k, v in self.batch.items()}
        return b

    def __getitem__(self, key):
        if key in self.batch:
            return self.batch[key]
        last = "last_" + key
        if last in self.batch:
            return self.batch[last][1:]
        raise KeyError("Batch(): specified key: %s or %s not found!" % (key, last))

    def __contains__(self, key):
        return key in self.batch or "last_" + key in self.batch

    def add(self, key, value):
        self.batch[key] = value
        return self

    def hist(self, hist_idx, key=None):
        if self.histdim is None:
            raise ValueError("No histdim information for the batch")
        if key is None:
            b = self.empty_copy()
            b.batch = {k: v.select(self.histdim, hist_idx).unsqueeze(self.histdim) for k, v in self.batch.items()}
            return b
        return self[key].select(self.histdim, hist_idx).unsqueeze(self.histdim)

    def half(self):
        b = self.empty_copy()
        b.batch = {k: v.half() for k, v in self.batch.items()}
        return b

    def cpu2gpu(self, gpu, non_blocking=True):
        return self   # already resident in HBM

    def to_numpy(self):
        return {k: v.cpu().numpy() for k, v in self.batch.items()}


class GCWrapper:
    def __init__(self, GC, batchsize, spec, batchdim=0, histdim=None, use_numpy=False, gpu=None, params=dict(), verbose=True,
                 num_recv=1):
        self.GC, self.params, self.gpu = GC, params, gpu
        self.batchdim, self.histdim = batchdim, histdim
        self.name2idx = {k: [i] for i, k in enumerate(sorted(spec.keys()))}
        self.idx2name = {v[0]: k for k, v in self.name2idx.items()}
        self.spec = spec
        self._cb = {}
        self._games_seen = 0

    def reg_has_callback(self, key):
        return key in self.name2idx

    def reg_callback_if_exists(self, key, cb):
        if self.reg_has_callback(key):
            return self.reg_callback(key, cb)
        return False

    def reg_callback(self, key, cb):
        if key not in self.name2idx:
            raise ValueError("Callback[%s] is not in the specification" % key)
        self._cb[key] = cb
        return True

    def _check_callbacks(self):
        for key in self.name2idx:
            if key not in self._cb:
                raise ValueError("GCWrapper.start(): No callback function for key = %s" % key)

    def start(self):
        self._check_callbacks()
        self.GC.ctx().start()

    def stop(self):
        self.GC.ctx().stop()

    def run(self, *args, **kwargs):
        ctx = self.GC.ctx()
        smem = ctx.wait()
        rows = smem.effective_batchsize()
        sp = self.GC._sp
        if rows > 0:
            picked = Batch(_GC=self.GC, _batchdim=self.batchdim, _histdim=self.histdim, s=sp.s[:rows])
            picked.smem, picked.batchsize, picked.max_batchsize = smem, rows, sp.max_rows
            cb = self._cb.get("actor_black") or self._cb.get("actor_white")
            reply = cb(picked, *args, **kwargs)
            if not isinstance(reply, dict):
                raise ValueError("actor callback must return a dict with pi and V")
            extra = [k for k in reply if k not in ("pi", "V", "a", "rv")]
            if extra:
                raise ValueError("Receive extra keys %s from reply!" % str(extra))   # utils_elf.py:406-409
            missing = [k for k in ("pi", "V") if k not in reply]
            if missing:
                raise ValueError("Missing keys %s absent in reply!" % str(missing))
            pi, v = reply["pi"], reply["V"]
            self.GC._reply = {"pi": pi if torch.is_tensor(pi) else torch.as_tensor(pi, device=sp.device),
                              "V": v if torch.is_tensor(v) else torch.as_tensor(v, device=sp.device)}
        ctx.step()
        # game_end fires once per finished game (the reference sends it from the game thread, distri_client.h:228-240)
        done = sp.games_finished() if self._cb.get("game_end") is not None else self._games_seen
        while self._games_seen < done:
            self._cb["game_end"](Batch(_GC=self.GC), *args, **kwargs)
            self._games_seen += 1


def install_reference_module_names():
    """Register this module under the names the reference's Python imports for its pybind11 extensions, so that
    `import _elfgames_go_inference as go` / `import _elfgames_go as go` (src_py/elfgames/go/game_inference.py:15, game.py:15:
    go.ContextOptions(), go.GameOptions(), go.GameContext(co, opt)) and `from _elf import *` (src_py/elf/__init__.py:8: TSOptions,
    SearchAlgoOptions) resolve to the device-resident engine.  Idempotent; never overwrites a real extension module."""
    import sys
    import types
    me = sys.modules[__name__]
    names = {"_elf": ("TSOptions", "SearchAlgoOptions", "ContextOptions"),
             "_elfgames_go_inference": ("ContextOptions", "GameOptions", "GameContext", "TSOptions", "SearchAlgoOptions"),
             "_elfgames_go": ("ContextOptions", "GameOptions", "GameContext", "TSOptions", "SearchAlgoOptions", "GameStats")}
    installed = []
    for mod, attrs in names.items():
        if mod in sys.modules and not getattr(sys.modules[mod], "__elf_amd_shim__", False):
            continue
        m = types.ModuleType(mod, "elf_amd.compat shim for the reference's pybind11 module " + mod)
        m.__elf_amd_shim__ = True
        for a in attrs:
            setattr(m, a, getattr(me, a))
        m.__all__ = list(attrs)
        sys.modules[mod] = m
        installed.append(mod)
    return installed
