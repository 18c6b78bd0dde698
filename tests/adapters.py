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
