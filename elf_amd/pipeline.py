"""Two lock-step game groups pipelined over two HIP streams: while the net evaluates the leaves of one group
(stream `net`), the other group's expand/backup/select/feature kernels run on stream `search`.  This is the
device-side analogue of the reference's double-buffered batches (`num_recv = 2` SharedMem per actor group,
src_py/elfgames/go/game.py:428, src_py/elf/utils_elf.py:82-97): the net never waits for the search.
"""
import torch

from .selfplay import SelfPlay


class PipelinedSelfPlay:
    def __init__(self, groups=2, seed=0, **kw):
        self.groups = [SelfPlay(seed=seed + 7919 * i if seed else 0, **kw) for i in range(groups)]
        dev = self.groups[0].device
        self.device = dev
        self.search_stream = torch.cuda.Stream(device=dev)
        self.net_stream = torch.cuda.Stream(device=dev)
        self._rows = [0] * groups
        self._primed = False
        self.num_games = sum(g.num_games for g in self.groups)
        self.timing = False
        self.t_select, self.t_expand = [], []   # (start, end) HIP event pairs on the search stream
        self.t_net = []                         # (start, end) pairs around the net callback on the net stream

    def close(self):
        for g in self.groups:
            g.close()

    def _begin(self, i):
        with torch.cuda.stream(self.search_stream):
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.search_stream)
            self._rows[i] = self.groups[i].begin_step()     # select + features on `search`; host waits for the row count
            if self.timing:
                e1.record(self.search_stream)
                self.t_select.append((e0, e1))

    def step(self, net_fn):
        """One batch for every group. net_fn(s_tensor, rows) -> (pi, V) is enqueued on the net stream."""
        if not self._primed:
            self._begin(0)
            self._primed = True
        n = len(self.groups)
        total = 0
        for i in range(n):
            g = self.groups[i]
            ev_sel = torch.cuda.Event()
            ev_sel.record(self.search_stream)
            with torch.cuda.stream(self.net_stream):
                self.net_stream.wait_event(ev_sel)          # features of group i are in g.s
                if self.timing:
                    n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    n0.record(self.net_stream)
                pi, v = net_fn(g.s, self._rows[i])
                if self.timing:
                    n1.record(self.net_stream)
                    self.t_net.append((n0, n1))
                ev_net = torch.cuda.Event()
                ev_net.record(self.net_stream)
            # next group's select overlaps this group's net
            self._begin((i + 1) % n) if n > 1 else None
            with torch.cuda.stream(self.search_stream):
                self.search_stream.wait_event(ev_net)
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.search_stream)
                g.end_step(pi, v)                           # expand + backup of group i
                if self.timing:
                    e1.record(self.search_stream)
                    self.t_expand.append((e0, e1))
                if pi is not None:
                    pi.record_stream(self.search_stream)
                    v.record_stream(self.search_stream)
            total += self._rows[i]
            if n == 1:
                self._begin(0)
        return total

    def stats(self):
        out = {}
        for g in self.groups:
            for k, v in g.stats().items():
                out[k] = out.get(k, 0) + v
        out["steps_per_move"] = self.groups[0].stats()["steps_per_move"]
        return out
