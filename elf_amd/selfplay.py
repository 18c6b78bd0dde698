"""Host side of MCTS self-play over the C ABI (include/elf_amd.h, elfsp_* / elfmcts_*).

Mirrors the reference's option structs and batch interface:
  TSOptions / SearchAlgoOptions      src_cpp/elf/ai/tree_search/tree_search_options.h:23-229
  GameOptions (subset)               src_cpp/elfgames/go/common/go_game_specific.h:16-268
  GCWrapper.reg_callback / run       src_py/elf/utils_elf.py:340-359,426-437
The policy/value net stays a PyTorch module called through the callback with batch["s"]; it returns
"pi" [B, N*N+1] and "V" [B] exactly as Evaluator.actor does (src_py/rlpytorch/trainer/trainer.py:73-116).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


class MctsOptions(C.Structure):
    """ElfMctsOptions"""
    _fields_ = [("num_rollouts_per_batch", C.c_int32), ("virtual_loss", C.c_int32), ("use_prior", C.c_int32),
                ("unexplored_q_zero", C.c_int32), ("root_unexplored_q_zero", C.c_int32), ("c_puct", C.c_float),
                ("komi", C.c_float), ("ply_pass_enabled", C.c_int32), ("remove_pass_if_dangerous", C.c_int32),
                ("rotation_flip", C.c_int32), ("num_threads", C.c_int32), ("reserved0", C.c_int32),
                ("required_version", C.c_int64)]


class SpOptions(C.Structure):
    """ElfSpOptions"""
    _fields_ = [("board_size", C.c_int32), ("num_games", C.c_int32), ("nodes_per_game", C.c_int32),
                ("num_rollouts_per_thread", C.c_int32), ("persistent_tree", C.c_int32), ("root_epsilon", C.c_float),
                ("root_alpha", C.c_float), ("seed", C.c_uint32), ("policy_distri_cutoff", C.c_int32),
                ("move_cutoff", C.c_int32), ("resign_thres", C.c_float), ("never_resign_prob", C.c_float),
                ("log_searches", C.c_int32), ("keep_records", C.c_int32), ("policy_distri_training_for_all", C.c_int32),
                ("model_ver", C.c_int32), ("game_idx_base", C.c_int32), ("job_hash", C.c_uint64), ("mcts", MctsOptions),
                ("white_puct", C.c_float), ("white_mcts_rollout_per_batch", C.c_int32), ("white_mcts_rollout_per_thread", C.c_int32),
                ("black_use_policy_network_only", C.c_int32), ("white_use_policy_network_only", C.c_int32), ("pick_method", C.c_int32),
                ("cheat_eval_new_model_wins_half", C.c_int32), ("cheat_selfplay_random_result", C.c_int32),
                ("following_pass", C.c_int32), ("reserved1", C.c_int32)]


class SpRequest(C.Structure):
    """ElfSpRequest"""
    _fields_ = [("black_ver", C.c_int64), ("white_ver", C.c_int64), ("black_resign_thres", C.c_float),
                ("white_resign_thres", C.c_float), ("never_resign_prob", C.c_float), ("num_game_thread_used", C.c_int32),
                ("player_swap", C.c_int32), ("async_", C.c_int32), ("client_type", C.c_int32)]


PICK_METHODS = {"most_visited": 0, "strongest_prior": 1, "uniform_random": 2}


class SpSearch(C.Structure):
    """ElfSpSearch"""
    _fields_ = [("game", C.c_int32), ("move_played", C.c_int32), ("best_action", C.c_int32), ("total_visits", C.c_int32),
                ("n_edges", C.c_int32), ("root_value", C.c_float), ("max_score", C.c_float), ("predicted_value", C.c_float)]


def job_hash(job_id):
    """64-bit hash of ContextOptions.job_id for the seed == 0 rule (the reference uses std::hash<std::string>; any stable
    64-bit hash serves: the seeds of that rule are time-based anyway)"""
    import hashlib
    return int.from_bytes(hashlib.blake2b((job_id or "").encode(), digest_size=8).digest(), "little")


STAT_FIELDS = ("moves", "games", "rollouts", "rows", "steps", "logged", "steps_per_move", "step_in_move", "node_visits", "boundary_ns",
               "boundaries", "boundary_wait_ns")


class SelfPlay:
    """G self-play games stepped together on one GPU; trees, boards and leaf features live in HBM.

    Seeds: game g of this context is the job-wide game game_idx_base + g and is seeded seed + game_idx_base + g (seed != 0).  Two
    contexts created with the same seed AND the same game_idx_base therefore play identical games: give every context of a job
    its own game_idx_base (PipelinedSelfPlay and bench.py do)."""

    def __init__(self, board_size=19, num_games=16, device=0, mcts_rollout_per_thread=8192, mcts_rollout_per_batch=16,
                 mcts_puct=1.5, mcts_virtual_loss=1, mcts_use_prior=True, mcts_persistent_tree=True, mcts_epsilon=0.0,
                 mcts_alpha=0.0, mcts_unexplored_q_zero=False, mcts_root_unexplored_q_zero=False, komi=7.5,
                 ply_pass_enabled=0, policy_distri_cutoff=0, move_cutoff=-1, resign_thres=0.0, never_resign_prob=0.0,
                 seed=0, nodes_per_game=None, log_searches=0, rotation_flip=True, remove_pass_if_dangerous=True,
                 feature_format="f32_nchw", keep_records=0, policy_distri_training_for_all=False, model_ver=0,
                 mcts_threads=1, game_idx_base=0, job_id="", required_version=-1, white_puct=-1.0, white_mcts_rollout_per_batch=-1,
                 white_mcts_rollout_per_thread=-1, black_use_policy_network_only=False, white_use_policy_network_only=False,
                 mcts_pick_method="most_visited", cheat_eval_new_model_wins_half=False, cheat_selfplay_random_result=False,
                 following_pass=False, dump_record_prefix=""):
        if not torch.cuda.is_available():
            raise RuntimeError("elf_amd.SelfPlay needs a ROCm GPU (no CPU fallback exists)")
        self.dump_record_prefix = dump_record_prefix      # pop_records() also writes every finished game as SGF (needs keep_records)
        self.L = _lib.lib()
        self.n = int(board_size)
        self.num_games = int(num_games)
        self.device = torch.device("cuda", device)
        self.num_action = self.n * self.n + 1
        if nodes_per_game is None:
            # one node per rollout of the current search + the subtree kept from the previous ones
            nodes_per_game = 4 * mcts_rollout_per_thread * mcts_threads + 1024
        nodes_per_game = (int(nodes_per_game) + 63) // 64 * 64
        mo = MctsOptions(mcts_rollout_per_batch, mcts_virtual_loss, int(mcts_use_prior), int(mcts_unexplored_q_zero),
                         int(mcts_root_unexplored_q_zero), mcts_puct, komi, ply_pass_enabled, int(remove_pass_if_dangerous),
                         int(rotation_flip), int(mcts_threads), 0, int(required_version))
        self.opt = SpOptions(self.n, self.num_games, nodes_per_game, mcts_rollout_per_thread, int(mcts_persistent_tree),
                             mcts_epsilon, mcts_alpha, seed, policy_distri_cutoff, move_cutoff, resign_thres, never_resign_prob,
                             log_searches, int(keep_records), int(policy_distri_training_for_all), int(model_ver), int(game_idx_base),
                             job_hash(job_id), mo, white_puct, white_mcts_rollout_per_batch, white_mcts_rollout_per_thread,
                             int(black_use_policy_network_only), int(white_use_policy_network_only), self._pick(mcts_pick_method),
                             int(cheat_eval_new_model_wins_half), int(cheat_selfplay_random_result), int(following_pass), 0)
        z = np.fromfile(_lib.ZOBRIST_BIN, dtype="<u8")
        zz = np.ascontiguousarray(z[: (self.n + 2) ** 2])
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        check(self.L.elfsp_create(C.byref(self.opt), device, zz.ctypes.data, C.byref(h)))
        self._h = h
        self.max_rows = self.L.elfsp_max_rows(self._h)
        self.edge_stride = self.L.elfmcts_edge_stride(self.L.elfsp_mcts(self._h))
        # the batcher's "s" tensor (common/game_feature.h:159-163): [B, 18, N, N] -- resident in HBM.  "f32_nchw" is the
        # reference's layout; "f16_nhwc" is the same logical tensor as fp16 channels_last, written directly by the select
        # kernel for an fp16 channels_last net
        if feature_format == "f16_nhwc":
            self.s = torch.zeros((self.max_rows, self.n, self.n, 18), dtype=torch.float16, device=self.device).permute(0, 3, 1, 2)
            check(self.L.elfmcts_set_feature_format(self.L.elfsp_mcts(self._h), 1))
        elif feature_format == "f32_nchw":
            self.s = torch.zeros((self.max_rows, 18, self.n, self.n), dtype=torch.float32, device=self.device)
        else:
            raise ValueError("feature_format must be 'f32_nchw' or 'f16_nhwc'")
        self.feature_format = feature_format
        self._rows = C.c_int(0)
        self._waited = True
        self._cb = {}

    @staticmethod
    def _pick(name):
        if name not in PICK_METHODS:
            raise ValueError("MCTS Pick method unknown! " + str(name))   # tree_search.h:521-524
        return PICK_METHODS[name]

    def close(self):
        if getattr(self, "_h", None):
            self.L.elfsp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        # self.stream (a torch.cuda.Stream, or None = torch's current stream): PipelinedSelfPlay gives every group its launch stream
        # once, instead of entering a torch.cuda.stream() context per call (~40 us of host time each: the per-step host cost of a
        # search-only loop was mostly these look-ups)
        st = getattr(self, "stream", None)
        if st is not None:
            return C.c_void_p(st.cuda_stream)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- batch interface
    def begin_step(self, wait_rows=True):
        """select + leaf features into self.s.  wait_rows=False: nothing is waited for (the row count stays on the device, the
        net evaluates all max_rows rows, end_step expands the counted ones); returns None then."""
        rows = C.byref(self._rows) if wait_rows else None
        check(self.L.elfsp_begin_step(self._h, C.c_void_p(self.s.data_ptr()), 18 * self.n * self.n, rows, self._stream()))
        self._waited = wait_rows
        return self._rows.value if wait_rows else None

    def last_rows(self):
        check(self.L.elfsp_last_rows(self._h, C.byref(self._rows)))
        return self._rows.value

    def end_step(self, pi, v, rv=None):
        rows = self._rows.value if self._waited else self.max_rows
        if rows:
            if pi.dtype != torch.float32 or not pi.is_contiguous():
                pi = pi.float().contiguous()
            if v.dtype != torch.float32 or not v.is_contiguous():
                v = v.float().contiguous()
            v = v.reshape(-1)
            if pi.shape[0] < rows or pi.shape[1] != self.num_action or v.shape[0] < rows:
                raise ValueError("reply shapes do not match the batch")
            rvp = None
            if rv is not None:
                rv = rv.to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
                if rv.shape[0] < rows:
                    raise ValueError("reply shapes do not match the batch")
                rvp = C.c_void_p(rv.data_ptr())
            check(self.L.elfsp_end_step(self._h, C.c_void_p(pi.data_ptr()), pi.stride(0), C.c_void_p(v.data_ptr()), rvp, self._stream()))
        else:
            check(self.L.elfsp_end_step(self._h, None, 0, None, None, self._stream()))

    def set_request(self, black_ver, white_ver=-1, resign_thres=0.0, never_resign_prob=0.0, async_=False, num_game_thread_used=-1,
                    player_swap=False, white_resign_thres=None, mcts_opt=None, client_type=0):
        """Client::setRequest (train/distri_client.h:318-331): every game receives it at the top of its next fifth act (at once
        while it waits); white_ver >= 0 starts evaluation games with a second AI for White (step them with begin_step2 / end_step2).
        mcts_opt (elf_amd.client.TsOptions, e.g. from parse_request_seq): the search options the request carries -- the reference's
        server dictates them (evaluation requests switch the Dirichlet noise off); None = the options the context was created with."""
        q = SpRequest(int(black_ver), int(white_ver), float(resign_thres), float(resign_thres if white_resign_thres is None else white_resign_thres),
                      float(never_resign_prob), int(num_game_thread_used), int(player_swap), int(async_), int(client_type))
        self.send_request(q, mcts_opt)

    def send_request(self, request, mcts_opt=None):
        """an SpRequest (+ TsOptions) as parse_request_seq returns them"""
        if mcts_opt is not None:
            # the rows of a step are num_games x threads x rollouts per batch: make room before the step that delivers the request
            rows = self.num_games * int(mcts_opt.num_threads) * int(mcts_opt.num_rollouts_per_batch)
            wk = int(self.opt.white_mcts_rollout_per_batch)
            self._grow_rows(rows, self.num_games * int(mcts_opt.num_threads) * wk if wk > 0 else rows)
        check(self.L.elfsp_set_request3(self._h, C.byref(request), C.byref(mcts_opt) if mcts_opt is not None else None))

    def _grow_rows(self, rows, rows_white):
        grow_b = rows > self.max_rows
        grow_w = getattr(self, "s_white", None) is not None and rows_white > self.max_rows_white
        if grow_b or grow_w:
            torch.cuda.synchronize(self.device)      # nothing may still be reading or writing the row tensors that are replaced
        if grow_b:
            # a step may be open (PipelinedSelfPlay has always run begin_step for the next step already): its rows sit in the old
            # tensor and the coming end_step evaluates self.s[:rows] -- they move into the front of the new one
            new = self._alloc_rows(rows)
            new[:self.max_rows].copy_(self.s)
            self.max_rows = rows
            self.s = new
        self._want_white = max(getattr(self, "_want_white", 0), rows_white)
        if grow_w:
            new = self._alloc_rows(rows_white)
            new[:self.max_rows_white].copy_(self.s_white)
            self.max_rows_white = rows_white
            self.s_white = new

    def _alloc_rows(self, rows):
        if self.feature_format == "f16_nhwc":
            return torch.zeros((rows, self.n, self.n, 18), dtype=torch.float16, device=self.device).permute(0, 3, 1, 2)
        return torch.zeros((rows, 18, self.n, self.n), dtype=torch.float32, device=self.device)

    # ---- games with two AIs: rows of the "actor_black" AI in self.s, rows of the "actor_white" AI in self.s_white
    def _white_rows(self):
        if getattr(self, "s_white", None) is None:
            rows = max(self.L.elfsp_max_rows_actor(self._h, 1), getattr(self, "_want_white", 0))   # incl. a request still on its way
            self.s_white = self._alloc_rows(rows)
            self.max_rows_white = rows
        return self.s_white

    def begin_step2(self):
        """-> (rows of actor_black, rows of actor_white)"""
        sw = self._white_rows()
        dst = (C.c_void_p * 2)(self.s.data_ptr(), sw.data_ptr())
        rows = (C.c_int * 2)(0, 0)
        check(self.L.elfsp_begin_step2(self._h, dst, 18 * self.n * self.n, rows, self._stream()))
        self._rows2 = (rows[0], rows[1])
        return self._rows2

    def end_step2(self, replies):
        """replies = [(pi, V, rv or None) or None for the two actors]"""
        pis, vs, rvs, keep = (C.c_void_p * 2)(), (C.c_void_p * 2)(), (C.c_void_p * 2)(), []
        stride = self.num_action
        for a in range(2):
            if not self._rows2[a]:
                continue
            pi, v, rv = replies[a]
            pi = pi.float().contiguous()
            v = v.float().contiguous().reshape(-1)
            if pi.shape[0] < self._rows2[a] or pi.shape[1] != self.num_action or v.shape[0] < self._rows2[a]:
                raise ValueError("reply shapes do not match the batch")
            keep += [pi, v]
            pis[a], vs[a] = pi.data_ptr(), v.data_ptr()
            if rv is not None:
                rv = rv.to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
                keep.append(rv)
                rvs[a] = rv.data_ptr()
        check(self.L.elfsp_end_step2(self._h, pis, stride, vs, rvs, self._stream()))

    def set_pick_seed(self, seed):
        """seed of the uniform_random pick generator (the reference: time(NULL) at the first search of the process)"""
        check(self.L.elfsp_set_pick_seed(self._h, int(seed) & 0xFFFFFFFF))

    def progress(self):
        out = (C.c_int64 * 6)()
        check(self.L.elfsp_progress(self._h, out))
        return dict(zip(("searches", "games", "open", "steps", "waiting", "barrier"), [int(x) for x in out]))

    def reg_callback(self, key, cb):
        """GCWrapper.reg_callback (utils_elf.py:340-359): cb(batch) -> dict(pi=..., V=...)"""
        self._cb[key] = cb

    def run(self):
        """GCWrapper.run (utils_elf.py:426-437): serve one batch."""
        rows = self.begin_step()
        reply = {"pi": None, "V": None}
        if rows:
            cb = self._cb.get("actor_black") or self._cb.get("actor")
            reply = cb({"s": self.s[:rows]})
        self.end_step(reply["pi"], reply["V"], reply.get("rv"))
        return rows

    # ---- results
    def stats(self):
        out = (C.c_int64 * 12)()
        check(self.L.elfsp_stats(self._h, out))
        return dict(zip(STAT_FIELDS, [int(x) for x in out]))

    def games_finished(self):
        return int(self.L.elfsp_games_finished(self._h))

    # ---- interactive play (the human_actor half of GoGameSelfPlay::act, game_selfplay.cc:290-330)
    def play(self, moves):
        """Forward externally chosen moves between two searches: moves[g] = reference Coord, -1 = none.  Raises ElfGoError if a
        move is refused (that game is left untouched)."""
        mv = np.ascontiguousarray(moves, dtype=np.int32)
        if mv.size != self.num_games:
            raise ValueError("one entry per game")
        check(self.L.elfsp_play(self._h, mv.ctypes.data, self._stream()))

    def preload(self, moves, move_to=-1):
        """GameOptions.preload_sgf / preload_sgf_move_to: follow `moves` (reference Coords) in every game; the first move_to are
        forwarded now, later searches have their move replaced by the next listed one (game_selfplay.cc:202-219,392-405)."""
        mv = np.ascontiguousarray(moves, dtype=np.uint16)
        check(self.L.elfsp_preload(self._h, mv.ctypes.data, mv.size, int(move_to), self._stream()))

    def restart(self, games):
        """finish_game(FR_CLEAR) + restart for the listed games"""
        g = np.ascontiguousarray(games, dtype=np.int32)
        check(self.L.elfsp_restart(self._h, g.ctypes.data, g.size, self._stream()))

    def last_score(self):
        out = np.zeros(self.num_games, np.float32)
        check(self.L.elfsp_last_score(self._h, out.ctypes.data))
        return out

    def last_moves(self):
        """per game: the Coord the last finished search forwarded, 1 (M_RESIGN) if the engine resigned, -1 if none yet"""
        out = np.zeros(self.num_games, np.int32)
        check(self.L.elfsp_last_moves(self._h, out.ctypes.data))
        return out

    def validate_trees(self):
        """elfmcts_validate: (violations, code, game, node, position) of the node-record invariants; synchronises (tests / debugging)"""
        out = np.zeros(5, np.int32)
        check(self.L.elfmcts_validate(self.L.elfsp_mcts(self._h), out.ctypes.data))
        return tuple(int(x) for x in out)

    def pool_info(self, actor=0, reset_peaks=False):
        """elfmcts_pool_info of one AI's trees: the context's shared node pool (synchronises) -> dict(small_total, small_free, big_total,
        big_free, live, live_max_game, peak_max_game, peak_sum_games); peak_* since the last reset"""
        out = np.zeros(8, np.int64)
        m = self.L.elfsp_mcts_actor(self._h, int(actor)) if actor else self.L.elfsp_mcts(self._h)
        check(self.L.elfmcts_pool_info(m, out.ctypes.data, int(bool(reset_peaks))))
        k = ("small_total", "small_free", "big_total", "big_free", "live", "live_max_game", "peak_max_game", "peak_sum_games")
        return dict(zip(k, (int(x) for x in out)))

    def count_live(self, actor=0):
        """elfmcts_count_live: node ids per game by a scan of the pool (tests)"""
        out = np.zeros(self.num_games, np.int32)
        m = self.L.elfsp_mcts_actor(self._h, int(actor)) if actor else self.L.elfsp_mcts(self._h)
        check(self.L.elfmcts_count_live(m, out.ctypes.data))
        return out

    def finish(self, games, reason):
        """finish_game(reason) + restart for the listed games (FinishReason: 0 resign, 1 two passes, 2 max step, 3 clear, 4 illegal)"""
        g = np.ascontiguousarray(games, dtype=np.int32)
        check(self.L.elfsp_finish(self._h, g.ctypes.data, g.size, int(reason), self._stream()))

    def board_engine(self):
        """GoEngine view of the game boards (slot g = game g), for showBoard / getNextPlayer / getLastMove / getScore"""
        from .engine import GoEngine
        return GoEngine.borrow(self.L.elfsp_engine(self._h), self.n, self.num_games, self.device)

    def pop_records(self):
        """Record JSON text of every finished game not yet collected (GameNotifier::OnGameEnd -> GoStateExt::dumpRecord,
        go_state_ext.h:131-148); needs keep_records > 0."""
        out = []
        need = C.c_size_t(0)
        while self.L.elfsp_records_pending(self._h) > 0:
            rc = self.L.elfsp_pop_record(self._h, None, 0, C.byref(need))
            buf = C.create_string_buffer(need.value + 1)
            check(self.L.elfsp_pop_record(self._h, buf, need.value + 1, C.byref(need)))
            out.append(buf.raw[:need.value].decode())
        if getattr(self, "dump_record_prefix", ""):
            # GameOptions.dump_record_prefix: finish_game writes every finished game as SGF (game_selfplay.cc:133-135)
            from .train import record_to_sgf, sgf_file_name
            for r in out:
                name = sgf_file_name(self.dump_record_prefix, r)
                with open(name, "w") as fh:
                    fh.write(record_to_sgf(self.n, r, self.opt, name) + "\n")
        return out

    def search_log(self):
        n = self.stats()["logged"]
        rec = (SpSearch * n)()
        ne = self.edge_stride
        coord = np.zeros((n, ne), np.int32); visits = np.zeros((n, ne), np.int32)
        prior = np.zeros((n, ne), np.float32); reward = np.zeros((n, ne), np.float32)
        if n:
            check(self.L.elfsp_search_log(self._h, 0, n, rec, coord.ctypes.data, visits.ctypes.data, prior.ctypes.data,
                                          reward.ctypes.data))
        return list(rec), coord, visits, prior, reward
