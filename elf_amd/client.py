"""The self-play client's wire formats over the C ABI (elfrec_client_*, elfrec_parse_request_seq; host only, no GPU needed):
what src_cpp/elfgames/go/train/distri_client.h sends to and receives from the training server.

  Records {identity, states, records}      common/record.h:401-470, kept and dumped by GuardedRecords (distri_client.h:111-170)
  ThreadState                              common/record.h:354-380 = GoStateExt::getThreadState (go_state_ext.h:149-157)
  MsgRequestSeq {request, seq}             common/record.h:152-171

The transport (ZeroMQ, elf/distributed) is not part of this library: `ClientRecords.dump_and_clear()` is the text the reference's
writer would send, `parse_request_seq()` reads the text its reply carries.
"""
import ctypes as C

from . import _lib
from ._lib import check
from .selfplay import SpRequest

PICK_NAMES = {0: "most_visited", 1: "strongest_prior", 2: "uniform_random"}


class ThreadState(C.Structure):
    """ElfThreadState"""
    _fields_ = [("thread_id", C.c_int32), ("seq", C.c_int32), ("move_idx", C.c_int32), ("reserved", C.c_int32),
                ("black", C.c_int64), ("white", C.c_int64)]


class TsOptions(C.Structure):
    """ElfTsOptions"""
    _fields_ = [("max_num_moves", C.c_int32), ("num_threads", C.c_int32), ("num_rollouts_per_thread", C.c_int32),
                ("num_rollouts_per_batch", C.c_int32), ("verbose", C.c_int32), ("verbose_time", C.c_int32), ("persistent_tree", C.c_int32),
                ("pick_method", C.c_int32), ("seed", C.c_int64), ("root_epsilon", C.c_float), ("root_alpha", C.c_float),
                ("virtual_loss", C.c_int32), ("use_prior", C.c_int32), ("unexplored_q_zero", C.c_int32),
                ("root_unexplored_q_zero", C.c_int32), ("c_puct", C.c_float), ("log_prefix", C.c_char * 60)]


class ClientRecords:
    """GuardedRecords: the records and thread states a client has collected since its last message"""

    def __init__(self, identity):
        self.L = _lib.lib()
        h = C.c_void_p()
        check(self.L.elfrec_client_create(identity.encode(), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.L.elfrec_client_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.L.elfrec_client_size(self._h)

    def feed(self, record_json):
        """one finished game: the Record text SelfPlay.pop_records() returns"""
        check(self.L.elfrec_client_feed(self._h, record_json.encode("latin-1")))

    def update_state(self, thread_id, seq, move_idx, black, white=-1):
        ts = ThreadState(int(thread_id), int(seq), int(move_idx), 0, int(black), int(white))
        check(self.L.elfrec_client_update_state(self._h, C.byref(ts)))

    def update_from(self, sp):
        """GameNotifier::OnStateUpdate for every game of a SelfPlay context, in game order; also feeds its finished games"""
        if hasattr(sp, "groups"):            # PipelinedSelfPlay: every group is a context
            for g in sp.groups:
                self.update_from(g)
            return
        n = sp.num_games
        buf = (ThreadState * n)()
        k = self.L.elfsp_thread_states(sp._h, buf, n)
        if k < 0:
            check(k)
        for t in buf[:k]:
            check(self.L.elfrec_client_update_state(self._h, C.byref(t)))
        for r in sp.pop_records():
            self.feed(r)

    def dump_and_clear(self):
        """GuardedRecords::dumpAndClear -> the message text"""
        n = self.L.elfrec_client_dump_and_clear(self._h, None, 0)
        if n < 0:
            check(int(n))
        buf = C.create_string_buffer(n + 1)
        m = self.L.elfrec_client_dump_and_clear(self._h, buf, n + 1)
        if m < 0:
            check(int(m))
        return buf.raw[:n].decode("latin-1")


def parse_request_seq(text):
    """MsgRequestSeq::createFromJson -> (SpRequest, seq, TsOptions); ElfGoError for what the reference throws on"""
    L = _lib.lib()
    q, t, seq = SpRequest(), TsOptions(), C.c_int64(-1)
    check(L.elfrec_parse_request_seq(text.encode("latin-1"), C.byref(q), C.byref(seq), C.byref(t)))
    return q, seq.value, t


def request_seq_to_json(request, ts_options, seq):
    """MsgRequestSeq::dumpJsonString"""
    L = _lib.lib()
    n = L.elfrec_request_seq_to_json(C.byref(request), C.byref(ts_options), int(seq), None, 0)
    if n < 0:
        check(int(n))
    buf = C.create_string_buffer(n + 1)
    check(min(0, int(L.elfrec_request_seq_to_json(C.byref(request), C.byref(ts_options), int(seq), buf, n + 1))))
    return buf.raw[:n].decode("latin-1")
