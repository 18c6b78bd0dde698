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
