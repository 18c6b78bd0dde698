"""Training-side replay loader over the C ABI (include/elf_amd.h, elftrain_* / elfrec_*): the trainer's input pipeline.

Mirrors, for a whole batch per call, what the reference does with one game thread per sample:
  GoGameTrain::act                      src_cpp/elfgames/go/train/game_train.cc:23-58
  GoStateExtOffline                     src_cpp/elfgames/go/common/go_state_ext.h:236-330
  GoFeature "train" extractors/schema   src_cpp/elfgames/go/common/game_feature.h:73-145,159-206
  Record / MsgResult JSON               src_cpp/elfgames/go/common/record.h:184-330
The batch dict has the keys the reference's "train" SharedMem group exposes to rlpytorch (game.py:385-402):
s, offline_a, winner, mcts_scores, predicted_value, move_idx, num_move, aug_code, selfplay_ver -- as CUDA tensors.
"""
import ctypes as C
import json

import numpy as np
import torch

from . import _lib
from ._lib import check
from .engine import GoEngine


class TrainBatch(C.Structure):
    """ElfTrainBatch"""
    _fields_ = [("s", C.c_void_p), ("s_stride", C.c_int64), ("s_format", C.c_int32), ("num_future_actions", C.c_int32),
                ("offline_a", C.c_void_p), ("winner", C.c_void_p), ("mcts_scores", C.c_void_p), ("predicted_value", C.c_void_p),
                ("move_idx", C.c_void_p), ("num_move", C.c_void_p), ("aug_code", C.c_void_p), ("selfplay_ver", C.c_void_p)]


def sgfstr_to_coords(board_size, sgf):
    """sgfstr2coords (sgf/sgf.h:97-125) -> uint16 reference Coords"""
    L = _lib.lib()
    k = L.elfrec_sgfstr_to_coords(board_size, sgf.encode(), None, 0)
    if k < 0:
        check(k)
    out = np.zeros(max(k, 1), np.uint16)
    L.elfrec_sgfstr_to_coords(board_size, sgf.encode(), out.ctypes.data, k)
    return out[:k]


def coords_to_sgfstr(board_size, coords):
    """coords2sgfstr (sgf/sgf.h:87-95)"""
    L = _lib.lib()
    c = np.ascontiguousarray(coords, dtype=np.uint16)
    n = L.elfrec_coords_to_sgfstr(board_size, c.ctypes.data, c.size, None, 0)
    if n < 0:
        check(n)
    buf = C.create_string_buffer(n + 1)
    L.elfrec_coords_to_sgfstr(board_size, c.ctypes.data, c.size, buf, n + 1)
    return buf.raw[:n].decode("latin-1")


def parse_record(board_size, rec):
    """Record::createFromJson (record.h:256-268) for the fields GoStateExtOffline::fromRecord reads (go_state_ext.h:248-258).
    rec: JSON text or the parsed dict.  -> dict(moves u16, reward, black_ver, policies u8 [k, (N+2)^2], values f32)"""
    j = json.loads(rec) if isinstance(rec, (str, bytes)) else rec
    res = j["result"]
    P = (board_size + 2) ** 2
    pol = res.get("policies") or []
    policies = np.zeros((len(pol), P), np.uint8)
    for i, row in enumerate(pol):
        policies[i, : len(row)] = row
    return dict(moves=sgfstr_to_coords(board_size, res["content"]), reward=float(res["reward"]),
                black_ver=int(j["request"]["vers"]["black_ver"]), policies=policies,
                values=np.asarray(res["values"], np.float32), seq=int(j.get("seq", 0)))


class ReplayLoader:
    """HBM-resident replay store + one-launch batch extraction (k_replay_extract).
    keep_states=False (default, the trainer's mode): a sample replays from its record's checkpoint (the state after every 16th
    move, written once per put) -- at most 15 board steps instead of ~160, the same rows.  keep_states=True: the reference's own
    procedure (reset + forward x move_to), and the replayed GoState of sample i stays in board slot i of `self.engine`."""

    def __init__(self, board_size=19, capacity=1024, batchsize=2048, device=0, max_moves=None, with_policies=True,
                 num_future_actions=1, seed=0, feature_format="f32_nchw", batches_per_launch=1, keep_states=False):
        if not torch.cuda.is_available():
            raise RuntimeError("elf_amd.ReplayLoader needs a ROCm GPU (no CPU fallback exists)")
        self.L = _lib.lib()
        self.n = int(board_size)
        self.na = self.n * self.n + 1
        self.batchsize = int(batchsize)
        self.nfa = int(num_future_actions)
        self.max_moves = int(max_moves or 2 * self.n * self.n)   # BOARD_MAX_MOVE (go_common.h:15)
        self.f16 = feature_format == "f16_nhwc"
        if not self.f16 and feature_format != "f32_nchw":
            raise ValueError("feature_format must be 'f32_nchw' or 'f16_nhwc'")
        # the trainer prefetches: `batches_per_launch` train batches are drawn and extracted by ONE launch (sample_batches); the
        # replay of a sample is a dependent chain of ~160 board steps, so the kernel wants many samples in flight
        self.batches_per_launch = max(1, int(batches_per_launch))
        # keep_states: sample i's replayed GoState lives in board slot i of this engine; otherwise the engine only lends its constants
        self.engine = GoEngine(self.n, self.batchsize * self.batches_per_launch if keep_states else 1, device)
        self.device = self.engine.device
        h = C.c_void_p()
        check(self.L.elftrain_create(self.engine._h, int(capacity), self.max_moves, int(with_policies), int(seed) & 0xFFFFFFFF, C.byref(h)))
        self._h = h
        self.keep_states = bool(keep_states)
        check(self.L.elftrain_set_keep_states(self._h, int(self.keep_states)))
        B, dev = self.batchsize * self.batches_per_launch, self.device
        self._draw = torch.zeros((3, B), dtype=torch.int32, device=dev)

    def close(self):
        if getattr(self, "_h", None):
            self.L.elftrain_destroy(self._h)
            self._h = None
            self.engine.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.L.elftrain_num_records(self._h)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def put(self, slot, rec):
        """GoStateExtOffline::fromRecord into record slot `slot`; rec = Record JSON text / dict, or a parse_record() dict"""
        r = rec if isinstance(rec, dict) and "moves" in rec else parse_record(self.n, rec)
        mv = np.ascontiguousarray(r["moves"], np.uint16)
        pol = np.ascontiguousarray(r["policies"], np.uint8)
        val = np.ascontiguousarray(r["values"], np.float32)
        # ordered on the stream the extractions run on: a put into the slot of an evicted record waits for the samples that still read it
        check(self.L.elftrain_put_async(self._h, int(slot), mv.ctypes.data, mv.size, float(r["reward"]), int(r["black_ver"]),
                                        pol.ctypes.data if pol.size else None, pol.shape[0] if pol.size else 0,
                                        val.ctypes.data if val.size else None, val.size, self._stream()))

    def _alloc(self, n):
        dev = self.device
        if self.f16:
            s = torch.empty((n, self.n, self.n, 18), dtype=torch.float16, device=dev).permute(0, 3, 1, 2)
        else:
            s = torch.empty((n, 18, self.n, self.n), dtype=torch.float32, device=dev)
        return dict(s=s, offline_a=torch.empty((n, self.nfa), dtype=torch.int64, device=dev),
                    winner=torch.empty(n, dtype=torch.float32, device=dev),
                    mcts_scores=torch.empty((n, self.na), dtype=torch.float32, device=dev),
                    predicted_value=torch.empty(n, dtype=torch.float32, device=dev),
                    move_idx=torch.empty(n, dtype=torch.int32, device=dev), num_move=torch.empty(n, dtype=torch.int32, device=dev),
                    aug_code=torch.empty(n, dtype=torch.int32, device=dev), selfplay_ver=torch.empty(n, dtype=torch.int64, device=dev))

    def extract(self, rec, move_to, d4, out=None):
        """The "train" batch for explicit (record slot, move_to, D4 code) triples (int32 tensors / sequences of equal length)."""
        dev = self.device
        t = [x.to(device=dev, dtype=torch.int32).contiguous() if isinstance(x, torch.Tensor)
             else torch.tensor(np.asarray(x, np.int32), device=dev) for x in (rec, move_to, d4)]
        n = t[0].numel()
        if n > self.batchsize * self.batches_per_launch:
            raise ValueError("more samples than the loader's batchsize x batches_per_launch")
        b = out if out is not None else self._alloc(n)
        tb = TrainBatch(b["s"].data_ptr(), 18 * self.n * self.n, 1 if self.f16 else 0, self.nfa, b["offline_a"].data_ptr(),
                        b["winner"].data_ptr(), b["mcts_scores"].data_ptr(), b["predicted_value"].data_ptr(), b["move_idx"].data_ptr(),
                        b["num_move"].data_ptr(), b["aug_code"].data_ptr(), b["selfplay_ver"].data_ptr())
        check(self.L.elftrain_extract(self._h, C.c_void_p(t[0].data_ptr()), C.c_void_p(t[1].data_ptr()), C.c_void_p(t[2].data_ptr()),
                                      n, C.byref(tb), self._stream()))
        self._keep = t   # keep the index tensors alive until the stream has consumed them
        return b

    def sample(self, n=None, out=None):
        """GoGameTrain::act for n samples: draw (record, move_to, D4) with the store's mt19937, then extract()."""
        n = int(n or self.batchsize)
        d = self._draw
        check(self.L.elftrain_draw(self._h, n, self.nfa, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()),
                                   C.c_void_p(d[2].data_ptr()), self._stream()))
        return self.extract(d[0, :n], d[1, :n], d[2, :n], out=out)

    def sample_batches(self, k=None, out=None):
        """k train batches (default batches_per_launch) of `batchsize` samples drawn and extracted by one launch: the same
        samples, in the same order, as k consecutive sample() calls (the draws come from the same mt19937 stream).
        -> list of k batch dicts (views of one allocation)."""
        k = int(k or self.batches_per_launch)
        B = self.batchsize
        big = self.sample(k * B, out=out)
        return [{key: t[i * B:(i + 1) * B] for key, t in big.items()} for i in range(k)]


class SgfHeader(C.Structure):
    """ElfSgfHeader"""
    _fields_ = [("size", C.c_int32), ("handi", C.c_int32), ("winner", C.c_int32), ("komi", C.c_float), ("win_margin", C.c_float)]


def parse_sgf(board_size, text):
    """Sgf::load + iterator (sgf/sgf.cc) -> (players int32 [k] (1 Black, 2 White), coords uint16 [k] (0 pass, 3 invalid), header dict),
    or None where the reference's loader fails (no header / no entry)"""
    L = _lib.lib()
    raw = text.encode("latin-1") if isinstance(text, str) else text
    h = SgfHeader()
    k = L.elfrec_sgf_parse(int(board_size), raw, None, None, 0, C.byref(h))
    if k < 0:
        check(k)
    if k == 0:
        return None
    pl, mv = np.zeros(k, np.int32), np.zeros(k, np.uint16)
    L.elfrec_sgf_parse(int(board_size), raw, pl.ctypes.data, mv.ctypes.data, k, C.byref(h))
    return pl, mv, dict(size=h.size, komi=h.komi, handi=h.handi, winner=h.winner, win_margin=h.win_margin)


def record_to_sgf(board_size, rec, opt, filename, git_hash=None, git_staged=None):
    """GoStateExt::dumpSgf (go_state_ext.cc:26-82) for a finished game: rec = Record JSON text / dict, opt = the SpOptions the game
    was played under (komi, policy-only flags) -> SGF text (result, player names, komi, every move with its predicted value)"""
    L = _lib.lib()
    j = json.loads(rec) if isinstance(rec, (str, bytes)) else rec
    mv = sgfstr_to_coords(board_size, j["result"]["content"])
    val = np.asarray(j["result"]["values"], np.float32)
    args = (C.byref(opt), mv.ctypes.data, mv.size, val.ctypes.data if val.size else None, val.size, float(j["result"]["reward"]),
            filename.encode(), git_hash.encode() if git_hash is not None else None, git_staged.encode() if git_staged is not None else None)
    n = L.elfrec_game_sgf(*args, None, 0)
    if n < 0:
        check(int(n))
    buf = C.create_string_buffer(n + 1)
    L.elfrec_game_sgf(*args, buf, n + 1)
    return buf.raw[:n].decode("latin-1")


def sgf_file_name(prefix, rec):
    """the file finish_game writes the game to (GoStateExt::dumpSgf(), go_state_ext.h:48-56): <prefix>_<game>_<seq>_<B|W>.sgf"""
    j = json.loads(rec) if isinstance(rec, (str, bytes)) else rec
    return "%s_%d_%d_%s.sgf" % (prefix, j["thread_id"], j["seq"], "B" if j["result"]["reward"] > 0 else "W")


def records_of_message(text):
    """Records::createFromJsonString (common/record.h:465-475), what TrainCtrl::OnReceive does with a client's message or with the
    content of an offline data file (DistriServer::loadOfflineSelfplayData): an object with "identity" is a Records message, anything
    else a plain array of Records.  -> (identity, list of Record dicts)"""
    j = json.loads(text)
    if isinstance(j, dict) and "identity" in j:
        return j["identity"], list(j.get("records", []))
    if not isinstance(j, list):
        raise ValueError("neither a Records message nor an array of Records")
    return "", list(j)


class ReaderQueues:
    """elf::shared::ReaderQueuesT<Record> + the draws of GoGameTrain::act over it (elfrq_*, host only): see include/elf_amd.h.
    Records are handles (slots of a ReplayLoader)."""

    def __init__(self, num_reader=50, queue_min_size=10, queue_max_size=1000, insert_seed=0, num_threads=1, seed=0, job_id=""):
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.elfrq_create(int(num_reader), int(queue_min_size), int(queue_max_size), int(insert_seed) & 0xFFFFFFFF, C.byref(h)))
        self._h = h
        self.num_reader = int(num_reader)
        from .selfplay import job_hash
        check(self.L.elfrq_set_threads(self._h, int(num_threads), int(seed), job_hash(job_id)))

    def close(self):
        if getattr(self, "_h", None):
            self.L.elfrq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def insert(self, slot, num_moves, black_win):
        """InsertWithParity -> (queue index, evicted handle or -1)"""
        ev = C.c_int32(-1)
        q = self.L.elfrq_insert(self._h, int(slot), int(num_moves), int(bool(black_win)), C.byref(ev))
        if q < 0:
            check(q)
        return q, ev.value

    def sizes(self):
        out = np.zeros(self.num_reader, np.int32)
        check(min(0, self.L.elfrq_sizes(self._h, out.ctypes.data)))
        return out

    def draw(self, num_acts, num_future_actions=1):
        """num_acts x 64 draws of GoGameTrain::act -> (slot, move_to, d4) int32 arrays"""
        k = 64 * int(num_acts)
        out = np.zeros((3, k), np.int32)
        check(self.L.elfrq_draw(self._h, int(num_acts), int(num_future_actions), out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return out[0], out[1], out[2]

    def draw_into(self, host, num_acts, num_future_actions=1):
        """the same into a (pinned) int32 host tensor [3, >= num_acts * 64]"""
        k = 64 * int(num_acts)
        if host.dtype != torch.int32 or host.dim() != 2 or host.shape[0] != 3 or host.shape[1] < k or not host.is_contiguous():
            raise ValueError("host must be a contiguous int32 tensor [3, >= num_acts * 64]")
        check(self.L.elfrq_draw(self._h, int(num_acts), int(num_future_actions), host[0].data_ptr(), host[1].data_ptr(), host[2].data_ptr()))


class ReplayBuffer:
    """The trainer's replay buffer as the reference keeps and samples it: records arrive (TrainCtrl::OnReceive ->
    InsertWithParity), live in ReaderQueues until their queue drops them, and train batches are what GoGameTrain game threads
    draw from the queues -- here the records sit in an HBM store (ReplayLoader), the queues hold their slots, and one
    k_replay_extract launch produces the rows of `acts` acts (64 rows each)."""

    def __init__(self, board_size=19, num_reader=50, queue_min_size=10, queue_max_size=1000, batchsize=2048, batches_per_launch=1,
                 insert_seed=0, num_threads=None, seed=0, job_id="", **loader_kw):
        if batchsize % 64:
            raise ValueError("batchsize must be a multiple of 64 (GoGameTrain sends 64 states per act)")
        capacity = int(num_reader) * int(queue_max_size) + 1     # a slot is freed only after the insert that evicts it
        self.loader = ReplayLoader(board_size=board_size, capacity=capacity, batchsize=batchsize, batches_per_launch=batches_per_launch,
                                   **loader_kw)
        self.queues = ReaderQueues(num_reader, queue_min_size, queue_max_size, insert_seed,
                                   num_threads if num_threads is not None else batchsize // 64, seed, job_id)
        self._free = list(range(capacity - 1, -1, -1))
        # draws go host -> device through a small ring of pinned staging tensors: the host draws the next launch's samples while the
        # device still replays the current one (nothing here waits for the stream)
        B = self.loader.batchsize * self.loader.batches_per_launch
        self._ring = [(torch.zeros((3, B), dtype=torch.int32).pin_memory(), torch.zeros((3, B), dtype=torch.int32, device=self.loader.device),
                       torch.cuda.Event()) for _ in range(3)]
        self._turn = 0

    def close(self):
        self.queues.close()
        self.loader.close()

    def insert(self, rec):
        """one finished game (Record JSON text / dict) -> the queue it went to"""
        r = rec if isinstance(rec, dict) and "moves" in rec else parse_record(self.loader.n, rec)
        slot = self._free.pop()
        try:
            self.loader.put(slot, r)
        except Exception:
            self._free.append(slot)              # a refused record (too long, malformed policies) must not leak its slot
            raise
        q, ev = self.queues.insert(slot, len(r["moves"]), r["reward"] > 0)
        if ev >= 0:
            self._free.append(ev)
        return q

    def insert_message(self, text, keep=None):
        """a client's message or an offline data file: every record goes into the buffer, in order (the reference's server first
        asks its model-version bookkeeping whether to keep a record, ctrl_selfplay.h: pass keep(record_dict) -> bool for that).
        -> number of records inserted"""
        k = 0
        for r in records_of_message(text)[1]:
            if keep is None or keep(r):
                self.insert(r)
                k += 1
        return k

    def sample(self, acts=None, out=None, events=None):
        """the rows of `acts` acts (default: one train batch = batchsize / 64 acts), in the order the game threads drew them;
        events = (start, end) torch.cuda.Events recorded around the extraction kernel"""
        acts = int(acts or self.loader.batchsize // 64)
        k = 64 * acts
        host, dev, done = self._ring[self._turn]
        self._turn = (self._turn + 1) % len(self._ring)
        if k > host.shape[1]:
            raise ValueError("more samples than the loader's batchsize x batches_per_launch")
        done.synchronize()                       # the copy that last read this staging tensor (three launches ago)
        self.queues.draw_into(host, acts, self.loader.nfa)
        dev[:, :k].copy_(host[:, :k], non_blocking=True)
        done.record(torch.cuda.current_stream(self.loader.device))
        if events:
            events[0].record()
        b = self.loader.extract(dev[0, :k], dev[1, :k], dev[2, :k], out=out)
        if events:
            events[1].record()
        return b

    def sample_batches(self, k=None, out=None):
        """k train batches (default batches_per_launch) from one launch -> list of k batch dicts (views of one allocation)"""
        k = int(k or self.loader.batches_per_launch)
        B = self.loader.batchsize
        big = self.sample(k * B // 64, out=out)
        return [{key: t[i * B:(i + 1) * B] for key, t in big.items()} for i in range(k)]
