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
    """HBM-resident replay store + one-launch batch extraction (k_replay_extract)."""

    def __init__(self, board_size=19, capacity=1024, batchsize=2048, device=0, max_moves=None, with_policies=True,
                 num_future_actions=1, seed=0, feature_format="f32_nchw", batches_per_launch=1):
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
        self.engine = GoEngine(self.n, self.batchsize * self.batches_per_launch, device)   # replay scratch: sample i replays in board slot i
        self.device = self.engine.device
        h = C.c_void_p()
        check(self.L.elftrain_create(self.engine._h, int(capacity), self.max_moves, int(with_policies), int(seed) & 0xFFFFFFFF, C.byref(h)))
        self._h = h
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
        check(self.L.elftrain_put(self._h, int(slot), mv.ctypes.data, mv.size, float(r["reward"]), int(r["black_ver"]),
                                  pol.ctypes.data if pol.size else None, pol.shape[0] if pol.size else 0,
                                  val.ctypes.data if val.size else None, val.size))

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
