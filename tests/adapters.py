"""Engine adapters so the same known-answer cases run on the CPU oracle and on the HIP engine."""
import numpy as np


class PortState:
    def __init__(self, port, s=None):
        self.P, self.s = port, (s if s is not None else port.new())

    def forward(self, c):
        return self.P.forward(self.s, c) == 1

    def clone(self):
        return PortState(self.P, self.P.clone(self.s))

    def ply(self):
        return int(self.P.info(self.s)[0])

    def colours(self):
        return self.P.board(self.s)[0]

    def libs(self):
        return self.P.board(self.s)[1]

    def caps(self):
        i = self.P.info(self.s)
        return int(i[7]), int(i[8])

    def evaluate(self, komi):
        return self.P.evaluate(self.s, komi)

    def terminated(self):
        return self.P.terminated(self.s)

    def features(self, d4):
        return self.P.extract_agz(self.s, d4)

    def legal_mask(self):
        return self.P.legal_mask(self.s)


class GpuState:
    """One slot of a GoEngine; slots are handed out by a simple bump allocator."""

    def __init__(self, eng, alloc, slot=None):
        self.E, self.alloc = eng, alloc
        if slot is None:
            slot = alloc()
            eng.reset([slot])
        self.slot = slot

    def forward(self, c):
        return int(self.E.forward([self.slot], [c]).cpu()[0]) == 1

    def clone(self):
        d = self.alloc()
        self.E.copy([d], [self.slot])
        return GpuState(self.E, self.alloc, d)

    def _info(self):
        return self.E.info([self.slot]).cpu().numpy()[0]

    def ply(self):
        return int(self._info()[0])

    def colours(self):
        return self.E.export_board([self.slot])[0].cpu().numpy()[0]

    def libs(self):
        return self.E.export_board([self.slot])[1].cpu().numpy()[0]

    def caps(self):
        i = self._info()
        return int(i[7]), int(i[8])

    def evaluate(self, komi):
        return float(self.E.evaluate([self.slot], komi).cpu()[0])

    def terminated(self):
        return bool(self._info()[9])

    def features(self, d4):
        return self.E.extract_agz([self.slot], [d4]).cpu().numpy()[0]

    def legal_mask(self):
        return self.E.legal_mask([self.slot]).cpu().numpy()[0]


class HashNet:
    """A deterministic policy/value function evaluated by torch ON THE GPU whose result depends only on the logical 0/1 content of
    the feature rows -- not on their dtype (fp32 / fp16) or memory layout (NCHW / channels_last): integer hashing of the planes,
    then a fixed elementwise map.  Lets serial fp32-NCHW self-play be compared bit for bit with the pipelined, HIP-graph-replayed,
    fp16 channels_last path (tests only)."""

    def __init__(self, n, device, salt=12345):
        import torch
        self.n, self.na = n, n * n + 1
        g = torch.Generator().manual_seed(salt)
        self.w = torch.randint(1, 2 ** 31 - 1, (18 * n * n,), generator=g, dtype=torch.int64).to(device)
        self.a = (torch.arange(self.na, dtype=torch.int64, device=device) + 1) * 2654435761

    def __call__(self, s):
        import torch
        b = s.shape[0]
        x = (s != 0).reshape(b, -1).to(torch.int64)          # logical NCHW order whatever the strides
        h = (x * self.w).sum(1) & 0x7FFFFFFF                  # exact integer arithmetic
        z = ((h[:, None] * 40503 + self.a[None, :]) >> 7) & 0xFFFF
        wgt = ((z + 1).to(torch.float32) * (1.0 / 65536.0)) ** 8   # peaky, like a trained policy
        pi = wgt / wgt.sum(1, keepdim=True)
        v = ((h % 513) - 256).to(torch.float32) * (1.0 / 256.0)
        return pi, v


def adversarial_net(n):
    """A deterministic "net" whose policy rows are the inputs std::sort is sensitive to (go/mcts/mcts.h:292-297 sorts all n*n+1
    (coord, prior) pairs by prior with an UNSTABLE introsort): per row, chosen by a hash of the feature row, one of -- a median-of-3
    killer (drives __introsort_loop into its depth limit: the __partial_sort fallback), ascending / descending / organ-pipe ramps,
    2 and 5 distinct values (large groups of equal priors), all equal, and a generic row on a coarse grid.  Values on the 1/256 grid
    (exact fp32 sums in any backup order).  Returns net(s [b,18,n,n]) -> (pi [b,n*n+1] f32, v [b] f32)."""
    import numpy as np
    na = n * n + 1

    def rows_for(kind, h):
        i = np.arange(na, dtype=np.float64)
        if kind == 0:      # median-of-3 killer for the descending comparator (tests/native/stl_emul_check.cc), made positive
            v = np.zeros(na)
            k = na // 2
            for j in range(1, k + 1):
                if j % 2 == 1:
                    v[j - 1] = -j
                    v[j] = -(k + j)
                v[k + j - 1] = -2 * j
            return 1.0 + v / 1024.0
        if kind == 1:
            return 1.0 + i
        if kind == 2:
            return 1.0 + (na - i)
        if kind == 3:
            return 1.0 + np.minimum(i, na - i)
        if kind == 4:
            return 1.0 + ((i * 7 + h) % 2)
        if kind == 5:
            return 1.0 + ((i * 11 + h) % 5)
        if kind == 6:
            return np.ones(na)
        return 1.0 + ((i * 2654435761 + h * 40503) % 97)

    def net(s):
        s = np.asarray(s)
        b = s.shape[0]
        pi = np.zeros((b, na), np.float32)
        v = np.zeros((b,), np.float32)
        w = (np.arange(s[0].size, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(12345)) & np.uint64(0xFFFFF)
        for r in range(b):
            h = int((s[r].reshape(-1).astype(np.uint64) * w).sum() & np.uint64(0x7FFFFFFF))
            row = rows_for(h % 8, h // 8)
            pi[r] = (row / row.sum()).astype(np.float32)
            v[r] = np.float32(((h >> 5) % 257 - 128) / 256.0)
        return pi, v
    return net
