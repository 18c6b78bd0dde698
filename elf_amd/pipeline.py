"""Lock-step game groups pipelined against the net: while the net evaluates the leaves of one group (stream `net`), the other
groups' expand/backup/select/feature kernels run on their own streams.  This is the device-side analogue of the reference's
double-buffered batches (`num_recv = 2` SharedMem per actor group, src_py/elfgames/go/game.py:428,
src_py/elf/utils_elf.py:82-97).

Per group i and step, in stream order:
    search[i]:  wait(net done i) -> expand + backup (end_step) -> select + leaf features (begin_step) -> record(sel i)
    net:        wait(sel i) -> net(s_i) -> record(net done i)
so a group's search work never sits between two net calls of the net stream, and the net stream only ever waits for the features
of the group it is about to evaluate.  With wait_rows=False (default) no host synchronisation happens inside a move: the row count
of a step stays on the device (elfsp_begin_step with n_rows = NULL), the net evaluates the fixed-shape tensor, the expansion kernel
reads the count.  Host work remains at move boundaries (Dirichlet draws, move choice, records).

Seeds: every group gets the same GameOptions.seed and a distinct game_idx_base, so game g of group i is the job-wide game
game_idx_base + i * num_games + g of the per-game seed rule (include/elf_amd.h, ElfSpOptions).
"""
import torch

from .selfplay import SelfPlay


class PipelinedSelfPlay:
    def __init__(self, groups=2, seed=0, game_idx_base=0, wait_rows=False, net_streams=1, search_priority=None, **kw):
        # the first group tells how many games a group holds (SelfPlay's own default when the caller does not say)
        self.groups = [SelfPlay(seed=seed, game_idx_base=game_idx_base, **kw)]
        ng = self.groups[0].num_games
        self.groups += [SelfPlay(seed=seed, game_idx_base=game_idx_base + i * ng, **kw) for i in range(1, groups)]
        dev = self.groups[0].device
        self.device = dev
        self.wait_rows = bool(wait_rows)
        # search_priority: HIP stream priority of the groups' search streams (negative = higher).  The search kernels are short and
        # latency-bound and share the GPU with the net's convolutions: at high priority their waves are dispatched ahead of the
        # convolution's next workgroups instead of waiting for slots (ELF_SEARCH_STREAM_PRIORITY overrides; default: the runtime's)
        import os
        if search_priority is None and os.environ.get("ELF_SEARCH_STREAM_PRIORITY", "") != "":
            search_priority = int(os.environ["ELF_SEARCH_STREAM_PRIORITY"])
        self.search_priority = search_priority
        if search_priority is None:
            self.search_streams = [torch.cuda.Stream(device=dev) for _ in range(groups)]
        else:
            self.search_streams = [torch.cuda.Stream(device=dev, priority=int(search_priority)) for _ in range(groups)]
        # net_streams = 1: the groups' net calls queue on one stream, one after the other.  net_streams = groups: every group has its
        # own net stream, so the memory-bound tails of one group's call (conv epilogues, heads) can run beside the other group's
        # convolutions
        self.net_streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, int(net_streams)))]
        self.net_stream = self.net_streams[0]
        self._ev_sel = [None] * groups
        self._rows = [0] * groups
        self._primed = False
        self.num_games = sum(g.num_games for g in self.groups)
        self.timing = False
        self.t_select, self.t_expand = [], []   # (start, end) HIP event pairs on the groups' search streams
        self.t_net = []                         # (start, end) pairs around the net callback on the net stream

    def close(self):
        for g in self.groups:
            g.close()

    def _pair(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def _begin(self, i):
        # no torch.cuda.stream() context here: the group launches on its own stream (SelfPlay.stream), events name theirs
        st = self.search_streams[i]
        if self.timing:
            e0, e1 = self._pair()
            e0.record(st)
        self._rows[i] = self.groups[i].begin_step(wait_rows=self.wait_rows)   # select + features
        if self.timing:
            e1.record(st)
            self.t_select.append((e0, e1))
        ev = torch.cuda.Event()
        ev.record(st)                       # right behind this group's feature kernel
        self._ev_sel[i] = ev

    def step(self, net_fn):
        """One batch for every group. net_fn(s_tensor, rows) -> (pi, V) is enqueued on the net stream; rows is None when the
        count stays on the device.  Returns the number of net rows of the step (None with wait_rows=False)."""
        n = len(self.groups)
        if not self._primed:
            for i in range(n):
                self.groups[i].stream = self.search_streams[i]
                self._begin(i)
            self._primed = True
        total = 0
        for i in range(n):
            g = self.groups[i]
            ns = self.net_streams[i % len(self.net_streams)]
            ns.wait_event(self._ev_sel[i])                           # features of group i are in g.s
            with torch.cuda.stream(ns):                              # the net is PyTorch code: it runs on torch's current stream
                if self.timing:
                    n0, n1 = self._pair()
                    n0.record(ns)
                pi, v = net_fn(g.s, self._rows[i])
                if self.timing:
                    n1.record(ns)
                    self.t_net.append((n0, n1))
            ev_net = torch.cuda.Event()
            ev_net.record(ns)
            st = self.search_streams[i]
            st.wait_event(ev_net)
            if self.timing:
                e0, e1 = self._pair()
                e0.record(st)
            if pi is not None and (pi.dtype != torch.float32 or not pi.is_contiguous() or v.dtype != torch.float32):
                with torch.cuda.stream(st):                          # the conversions are PyTorch kernels: on the group's stream
                    g.end_step(pi, v)
            else:
                g.end_step(pi, v)                                    # expand + backup of group i (+ the move boundary)
            if self.timing:
                e1.record(st)
                self.t_expand.append((e0, e1))
            if pi is not None:
                pi.record_stream(st)
                v.record_stream(st)
            if self._rows[i] is not None:
                total += self._rows[i]
            self._begin(i)                                           # next step's select, behind the expansion on the same stream
        return total if self.wait_rows else None

    # ---- requests and evaluation games (two AIs): the same pipeline through begin_step2 / end_step2 -------------------------------
    def set_request(self, *args, **kw):
        """SelfPlay.set_request for every group (each group is a context of its own: it restarts at its own barrier)"""
        for g in self.groups:
            g.set_request(*args, **kw)

    def send_request(self, request, mcts_opt=None):
        for g in self.groups:
            g.send_request(request, mcts_opt)

    def pop_records(self):
        return [r for g in self.groups for r in g.pop_records()]

    def _begin2(self, i):
        import ctypes as C
        g, st = self.groups[i], self.search_streams[i]
        with torch.cuda.stream(st):
            self._rows2[i] = g.begin_step2()        # select + features of both AIs; the row counts are waited for
            bv, wv = C.c_int64(0), C.c_int64(0)
            if g.L.elfsp_take_game_starts(g._h, C.byref(bv), C.byref(wv)):   # the "game_start" batch of this group
                self.versions[i] = (bv.value, wv.value)
            ev = torch.cuda.Event()
            ev.record(st)
            self._ev_sel[i] = ev

    def step2(self, net_fns):
        """One batch for every group, for contexts that receive requests and may play evaluation games.  net_fns = (black_fn,
        white_fn); fn(s, rows, version) -> (pi, V) evaluates rows `s[:rows]` with the model of that version (the versions of the
        group's last "game_start"; groups restart under a request at different steps).  Replies carry the version in rv.
        Returns the number of net rows of the step."""
        n = len(self.groups)
        if not getattr(self, "_primed2", False):
            if self._primed:
                raise RuntimeError("step() and step2() cannot be mixed on one pipeline")
            self._rows2 = [(0, 0)] * n
            self.versions = [(g.opt.model_ver, -1) for g in self.groups]
            for i in range(n):
                self._begin2(i)
            self._primed2 = True
        total = 0
        for i in range(n):
            g = self.groups[i]
            ns = self.net_streams[i % len(self.net_streams)]
            rows = self._rows2[i]
            replies = [None, None]
            with torch.cuda.stream(ns):
                ns.wait_event(self._ev_sel[i])
                for a in range(2):
                    if rows[a]:
                        s = g.s if a == 0 else g.s_white
                        ver = self.versions[i][a]
                        pi, v = net_fns[a](s, rows[a], ver)
                        replies[a] = (pi, v, torch.full((rows[a],), ver, dtype=torch.int64, device=self.device))
                ev_net = torch.cuda.Event()
                ev_net.record(ns)
            st = self.search_streams[i]
            with torch.cuda.stream(st):
                st.wait_event(ev_net)
                g.end_step2(replies)
                for r in replies:
                    if r is not None:
                        for t in r:
                            t.record_stream(st)
            total += rows[0] + rows[1]
            self._begin2(i)
        return total

    def synchronize(self):
        for st in self.search_streams:
            st.synchronize()
        for ns in self.net_streams:
            ns.synchronize()

    def stats(self):
        out = {}
        for g in self.groups:
            for k, v in g.stats().items():
                out[k] = out.get(k, 0) + v
        out["steps_per_move"] = self.groups[0].stats()["steps_per_move"]
        out["step_in_move"] = self.groups[0].stats()["step_in_move"]
        return out
