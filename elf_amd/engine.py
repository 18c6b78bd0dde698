"""Host side of the board engine: a thin, batch-oriented mirror of the reference's GoState /
BoardFeature interface over the C ABI (include/elf_amd.h).

Reference interface mirrored (src_cpp/elfgames/go/base/):
  GoState::reset / copy-ctor / forward / checkMove / terminated / evaluate   go_state.{h,cc}
  BoardFeature::extractAGZ / coord2Action / action2Coord / setD4Code         board_feature.{h,cc}
Every call works on a batch of board slots that live in HBM; torch is used only to own device
buffers and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

M_PASS, M_RESIGN, M_SKIP, M_INVALID, M_CLEAR = 0, 1, 2, 3, 4  # base/common.h:43-47
S_EMPTY, S_BLACK, S_WHITE, S_OFF_BOARD = 0, 1, 2, 3           # base/common.h:36-39
NUM_AGZ_PLANES = 18                                           # base/board_feature.h:38
INFO_WORDS = 16
INFO_FIELDS = ("ply", "next_player", "last_move", "last_move2", "ko_age", "simple_ko", "simple_ko_color",
               "b_cap", "w_cap", "terminated", "superko", "hist_len", "sk_len", "hash_lo", "hash_hi")


def coord(n, x, y):
    """OFFSETXY (base/board.h:183-184)"""
    return (y + 1) * (n + 2) + (x + 1)


def coord_xy(n, c):
    """X(c), Y(c) (base/board.h:178-179)"""
    return c % (n + 2) - 1, c // (n + 2) - 1


def d4_transform(n, d4, x, y):
    """BoardFeature::Transform (base/board_feature.h:97-113)"""
    rot, flip = d4 % 4, (d4 >> 2) == 1
    if rot == 1:
        x, y = y, n - x - 1
    elif rot == 2:
        x, y = n - x - 1, n - y - 1
    elif rot == 3:
        x, y = n - y - 1, x
    if flip:
        x, y = y, x
    return x, y


def d4_inv_transform(n, d4, x, y):
    """BoardFeature::InvTransform (base/board_feature.h:115-130)"""
    rot, flip = d4 % 4, (d4 >> 2) == 1
    if flip:
        x, y = y, x
    if rot == 1:
        x, y = n - y - 1, x
    elif rot == 2:
        x, y = n - x - 1, n - y - 1
    elif rot == 3:
        x, y = y, n - x - 1
    return x, y


def coord2action(n, d4, c):
    """BoardFeature::coord2Action (base/board_feature.h:132-137)"""
    if c == M_PASS:
        return n * n
    x, y = d4_transform(n, d4, *coord_xy(n, c))
    return x * n + y


def action2coord(n, d4, a):
    """BoardFeature::action2Coord (base/board_feature.h:139-144)"""
    if a == -1 or a == n * n:
        return M_PASS
    x, y = d4_inv_transform(n, d4, a // n, a % n)
    return coord(n, x, y)


class GoEngine:
    """A pool of `capacity` boards of one size resident in HBM on `device`."""

    def __init__(self, board_size=19, capacity=4096, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("elf_amd.GoEngine needs a ROCm GPU (no CPU fallback exists)")
        self.L = _lib.lib()
        self.n = int(board_size)
        self.capacity = int(capacity)
        self.device = torch.device("cuda", device)
        self.num_action = self.n * self.n + 1
        z = np.fromfile(_lib.ZOBRIST_BIN, dtype="<u8")
        if z.size != 441:
            raise RuntimeError("bad zobrist table")
        zz = np.ascontiguousarray(z[: (self.n + 2) ** 2])
        h = C.c_void_p()
        check(self.L.elfgo_create(self.n, self.capacity, device, zz.ctypes.data, C.byref(h)))
        self._h = h

    @classmethod
    def borrow(cls, handle, board_size, capacity, device):
        """A non-owning view of an engine another object owns (the game boards of a SelfPlay): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = _lib.lib()
        self.n, self.capacity, self.device = int(board_size), int(capacity), device
        self.num_action = self.n * self.n + 1
        self._h = C.c_void_p(handle)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.L.elfgo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ids(self, ids, n=None):
        """-> (device int32 tensor or None, pointer, count)"""
        if ids is None:
            k = self.capacity if n is None else n
            return None, None, k
        t = torch.as_tensor(ids, dtype=torch.int32, device=self.device).contiguous()
        return t, C.c_void_p(t.data_ptr()), t.numel()

    def _i32(self, v, n):
        if v is None:
            return None, None
        t = torch.as_tensor(v, dtype=torch.int32, device=self.device).contiguous()
        assert t.numel() == n
        return t, C.c_void_p(t.data_ptr())

    def sync(self):
        check(self.L.elfgo_sync(self._h, self._stream()))

    # ---- GoState surface (batched)
    def reset(self, ids=None):
        t, p, k = self._ids(ids)
        check(self.L.elfgo_reset(self._h, p, k, self._stream()))

    def copy(self, dst_ids, src_ids):
        d, dp, k = self._ids(dst_ids)
        s, sp, k2 = self._ids(src_ids)
        assert k == k2
        check(self.L.elfgo_copy(self._h, dp, sp, k, self._stream()))

    def forward(self, ids, moves):
        """GoState::forward for each (slot, Coord). Returns uint8 tensor: 1 played, 0 refused, 255 M_INVALID."""
        mv = torch.as_tensor(moves, dtype=torch.int32, device=self.device).contiguous()
        t, p, k = self._ids(ids, mv.numel())
        assert k == mv.numel()
        ok = torch.empty(k, dtype=torch.uint8, device=self.device)
        check(self.L.elfgo_forward(self._h, p, C.c_void_p(mv.data_ptr()), k, C.c_void_p(ok.data_ptr()), self._stream()))
        return ok

    def legal_mask(self, ids=None, n=None):
        t, p, k = self._ids(ids, n)
        out = torch.empty((k, self.num_action), dtype=torch.uint8, device=self.device)
        check(self.L.elfgo_legal_mask(self._h, p, k, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def extract_agz(self, ids=None, d4=None, out=None, n=None, fmt="f32_nchw"):
        """BoardFeature::extractAGZ into `out` [k,18,N,N] (allocated if None): the batcher's "s" tensor.
        fmt "f32_nchw": fp32 contiguous rows (the reference's layout); "f16_nhwc": fp16 channels_last rows, i.e. what
        an fp16 channels_last net reads without a cast/permute pass."""
        t, p, k = self._ids(ids, n)
        d, dp = self._i32(d4, k)
        f16 = fmt == "f16_nhwc"
        if not f16 and fmt != "f32_nchw":
            raise ValueError("fmt must be 'f32_nchw' or 'f16_nhwc'")
        row = NUM_AGZ_PLANES * self.n * self.n
        if out is None:
            if f16:
                out = torch.empty((k, self.n, self.n, NUM_AGZ_PLANES), dtype=torch.float16, device=self.device).permute(0, 3, 1, 2)
            else:
                out = torch.empty((k, NUM_AGZ_PLANES, self.n, self.n), dtype=torch.float32, device=self.device)
        assert out.is_cuda and out.shape[0] >= k and out.dtype == (torch.float16 if f16 else torch.float32)
        inner = (1, self.n * NUM_AGZ_PLANES, NUM_AGZ_PLANES) if f16 else (self.n * self.n, self.n, 1)
        assert tuple(out.stride()[1:]) == inner, "rows must be %s" % ("channels_last" if f16 else "contiguous [18,N,N]")
        stride = out.stride(0) if out.shape[0] > 1 else row
        check(self.L.elfgo_extract_agz_fmt(self._h, p, dp, k, C.c_void_p(out.data_ptr()), stride, 1 if f16 else 0, self._stream()))
        return out

    def evaluate(self, ids=None, komi=7.5, n=None):
        t, p, k = self._ids(ids, n)
        out = torch.empty(k, dtype=torch.float32, device=self.device)
        check(self.L.elfgo_evaluate(self._h, p, k, float(komi), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def info(self, ids=None, n=None):
        """-> int32 tensor [k,16] on device; see INFO_FIELDS"""
        t, p, k = self._ids(ids, n)
        out = torch.empty((k, INFO_WORDS), dtype=torch.int32, device=self.device)
        check(self.L.elfgo_info(self._h, p, k, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def info_host(self, ids=None, n=None):
        a = self.info(ids, n).cpu().numpy()
        d = {f: a[:, i].copy() for i, f in enumerate(INFO_FIELDS)}
        d["hash"] = (a[:, 13].astype(np.uint32).astype(np.uint64)) | (a[:, 14].astype(np.uint32).astype(np.uint64) << np.uint64(32))
        return d

    def export_board(self, ids=None, n=None):
        t, p, k = self._ids(ids, n)
        col = torch.empty((k, self.n * self.n), dtype=torch.uint8, device=self.device)
        lib = torch.empty((k, self.n * self.n), dtype=torch.int16, device=self.device)
        check(self.L.elfgo_export_board(self._h, p, k, C.c_void_p(col.data_ptr()), C.c_void_p(lib.data_ptr()), self._stream()))
        return col, lib

    def playout(self, seeds, ids=None, max_steps=1 << 20, out=None):
        """config-2 protocol: random legal non-true-eye play to game end, whole games in one launch.
        seeds: uint64 per board (numpy or int64-viewed tensor). Returns uint32-as-int64 tensor [k,4] =
        hash_lo, hash_hi, ply, steps."""
        if isinstance(seeds, torch.Tensor):
            sd = seeds.to(self.device).contiguous()
        else:
            sd = torch.from_numpy(np.ascontiguousarray(seeds, dtype=np.uint64).view(np.int64)).to(self.device)
        k = sd.numel()
        t, p, k2 = self._ids(ids, k)
        assert k2 == k
        if out is None:
            out = torch.empty((k, 4), dtype=torch.int32, device=self.device)
        check(self.L.elfgo_playout(self._h, p, C.c_void_p(sd.data_ptr()), k, int(max_steps), C.c_void_p(out.data_ptr()), self._stream()))
        return out
