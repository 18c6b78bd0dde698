"""CPU: the client's wire formats (elfrec_client_*, elfrec_parse_request_seq, elfrec_request_seq_to_json; host-only code of
libelf_amd.so) against texts written by the REAL reference objects -- Records / ThreadState / MsgRequestSeq of
src_cpp/elfgames/go/common/record.h (tests/golden/wire_formats.json, oracle/gen_golden_wire.py)."""
import json
import os

import pytest

from conftest import GOLDEN
from pyoracle import MCTS_DEFAULTS, RefSelfPlay


@pytest.fixture(scope="module")
def wire():
    with open(os.path.join(GOLDEN, "wire_formats.json")) as fh:
        return json.load(fh)


def test_records_messages_equal_the_reference_s(built, wire):
    """GuardedRecords sessions: the same state updates and finished games in the same order give the same message text at every
    dumpAndClear -- identity escaping, "records" / "states" present only when non-empty, and the order of "states" (the iteration
    order of the reference's std::unordered_map<int, ThreadState>, which survives clear() with its buckets)."""
    import elf_amd
    for ses in wire["sessions"]:
        c = elf_amd.ClientRecords(ses["identity"])
        dumps = iter(ses["dumps"])
        for op in ses["ops"]:
            if op["op"] == "state":
                c.update_state(op["thread_id"], op["seq"], op["move_idx"], op["black"], op["white"])
            elif op["op"] == "feed":
                c.feed(wire["records"][op["rec"]])
            else:
                want = next(dumps)
                got = c.dump_and_clear()
                assert got == want, (ses["identity"], got[:200], want[:200])
                assert len(c) == 0
        c.close()


def test_request_texts(built, wire):
    """MsgRequestSeq: the server's text is parsed into the request / TSOptions it was written from, written back byte for byte, and
    a text with a field removed is accepted or refused exactly as MsgRequestSeq::createFromJson accepts it or throws."""
    import elf_amd
    for r in wire["requests"]:
        p = dict(MCTS_DEFAULTS)
        p.update(black_ver=0, white_ver=-1, client_type=1, num_game_thread_used=-1, black_thres=0.0, white_thres=0.0, never_resign_prob=0.0,
                 player_swap=0, async_=0, seq=0)
        p.update(r["params"])
        q, seq, t = elf_amd.parse_request_seq(r["text"])
        assert (q.black_ver, q.white_ver, q.client_type, q.num_game_thread_used, q.player_swap, q.async_, seq) == \
            (p["black_ver"], p["white_ver"], p["client_type"], p["num_game_thread_used"], p["player_swap"], p["async_"], p["seq"])
        import numpy as np
        f32 = lambda v: float(np.float32(v))
        assert (q.black_resign_thres, q.white_resign_thres, q.never_resign_prob) == (f32(p["black_thres"]), f32(p["white_thres"]), f32(p["never_resign_prob"]))
        assert (t.num_threads, t.num_rollouts_per_thread, t.num_rollouts_per_batch, t.virtual_loss, t.persistent_tree, t.pick_method) == \
            (p["mcts_threads"], p["rollouts_per_thread"], p["rollouts_per_batch"], p["virtual_loss"], p["persistent_tree"], p["pick_method"])
        assert (t.c_puct, t.root_epsilon, t.root_alpha) == (f32(p["c_puct"]), f32(p["root_epsilon"]), f32(p["root_alpha"]))
        assert (t.use_prior, t.unexplored_q_zero, t.root_unexplored_q_zero) == (p["use_prior"], p["unexplored_q_zero"], p["root_unexplored_q_zero"])
        assert (t.max_num_moves, t.seed, t.verbose, t.verbose_time, t.log_prefix) == (0, 0, 0, 0, b"")
        assert elf_amd.request_seq_to_json(q, t, seq) == r["text"]
        for v in r["variants"]:
            if v["roundtrip"] is None:
                with pytest.raises(elf_amd.ElfGoError):
                    elf_amd.parse_request_seq(v["text"])
            else:
                q2, seq2, t2 = elf_amd.parse_request_seq(v["text"])
                assert elf_amd.request_seq_to_json(q2, t2, seq2) == v["roundtrip"], v["removed"]
    for bad in ("", "{", "[]", '{"seq":1}', '{"request":{},"seq":1}', r["text"] + "x", r["text"][:-1]):
        with pytest.raises(elf_amd.ElfGoError):
            elf_amd.parse_request_seq(bad)
    # the JSON grammar itself, as nlohmann enforces it: numbers (no hex, no leading '+' or zeros, no inf / nan, digits after '.' and
    # the exponent mark), string escapes (known ones only, no raw control characters, well-formed surrogate pairs)
    good = r["text"]
    assert '"seq":' in good
    head, tail = good.rsplit('"seq":', 1)
    num_end = len(tail) - len(tail.lstrip("-0123456789"))
    for lit in ("0x10", "+1", "01", "1.", ".5", "1e", "1e+", "inf", "nan", "-", "--1"):
        with pytest.raises(elf_amd.ElfGoError):
            elf_amd.parse_request_seq(head + '"seq":' + lit + tail[num_end:])
    for lit in ("7", "-0", "1e2", "1.5E+1", "12.25"):
        elf_amd.parse_request_seq(head + '"seq":' + lit + tail[num_end:])
    for strbad in ('"a\\qb"', '"a\tb"', '"\\ud83d"', '"\\ude00"', '"\\ud83d\\u0041"', '"\\u12g4"'):
        with pytest.raises(elf_amd.ElfGoError):
            elf_amd.parse_request_seq(good[:-1] + ',"extra":' + strbad + "}")
    for strok in ('"a\\/b\\"c\\\\"', '"\\ud83d\\ude00"', '"\\u00e9"'):
        elf_amd.parse_request_seq(good[:-1] + ',"extra":' + strok + "}")


def test_reference_server_reads_our_messages(built, wire):
    """Records::createFromJsonString (TrainCtrl::OnReceive) on messages built here"""
    if not RefSelfPlay.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import elf_amd
    R = RefSelfPlay(9)
    c = elf_amd.ClientRecords("box-1")
    for t in range(6):
        c.update_state(t, 2 + t, 10 * t, 7, -1)
    c.feed(wire["records"][0])
    c.feed(wire["records"][3])
    text = c.dump_and_clear()
    assert R.records_parse(text) == (2, 6, sum(10 * t for t in range(6)), "box-1")
    assert R.records_parse(c.dump_and_clear()) == (0, 0, 0, "box-1")      # nothing collected: {"identity":"box-1"}
    q, seq, t = elf_amd.parse_request_seq(wire["requests"][1]["text"])
    assert R.request_seq_roundtrip(elf_amd.request_seq_to_json(q, t, seq + 1)) == wire["requests"][1]["text"].replace('"seq":%d' % seq, '"seq":%d' % (seq + 1))
    c.close()


def test_records_of_message(built, wire):
    """the two texts TrainCtrl::OnReceive accepts: a client's Records message and a plain array of Records (offline data files)"""
    from elf_amd.train import records_of_message
    ses = wire["sessions"][0]
    ident, recs = records_of_message(ses["dumps"][0])
    assert ident == ses["identity"] and len(recs) == sum(1 for op in ses["ops"][: [i for i, o in enumerate(ses["ops"]) if o["op"] == "dump"][0]] if op["op"] == "feed")
    ident, recs = records_of_message("[" + ",".join(wire["records"][:3]) + "]")
    assert ident == "" and [r["seq"] for r in recs] == [json.loads(t)["seq"] for t in wire["records"][:3]]
    assert records_of_message('{"identity":"x"}') == ("x", [])
    with pytest.raises(ValueError):
        records_of_message('{"records":[]}')


def test_random_request_texts_against_the_reference_parser(built, wire):
    """2000 server replies with fields removed or given another JSON type (number for a flag, flag for a number, string, null,
    array ...): accepted or refused exactly as MsgRequestSeq::createFromJson does -- nlohmann's conversions by target type (int
    takes numbers and booleans, int64_t numbers only, float numbers and booleans, bool only booleans, the enum only numbers) --
    and, when accepted, written back to the same text"""
    if not RefSelfPlay.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import copy
    import numpy as np
    import elf_amd
    R = RefSelfPlay(9)
    rng = np.random.default_rng(3)
    alts = [True, False, 0, 1, -1, 2, 1.5, 0.25, -0.75, 1e-3, 40000, "most_visited", "strongest_prior", "x", None, [], {}]

    def leaves(d, path=()):
        for k, v in d.items():
            if isinstance(v, dict):
                yield from leaves(v, path + (k,))
            else:
                yield path + (k,)

    accepted = refused = 0
    for _ in range(2000):
        j = copy.deepcopy(json.loads(wire["requests"][rng.integers(0, len(wire["requests"]))]["text"]))
        paths = list(leaves(j))
        for _ in range(rng.integers(1, 4)):
            p = paths[rng.integers(0, len(paths))]
            d = j
            for k in p[:-1]:
                d = d.get(k) if isinstance(d, dict) else None
                if not isinstance(d, dict):
                    break
            if not isinstance(d, dict) or p[-1] not in d:
                continue
            if rng.random() < 0.25:
                del d[p[-1]]
            else:
                d[p[-1]] = alts[rng.integers(0, len(alts))]
        t = json.dumps(j, separators=[(",", ":"), (", ", ": ")][rng.integers(0, 2)])
        want = R.request_seq_roundtrip(t)
        try:
            q, seq, ts = elf_amd.parse_request_seq(t)
        except elf_amd.ElfGoError:
            assert want is None, t
            refused += 1
            continue
        assert want is not None, t
        accepted += 1
        if ts.pick_method >= 0:                      # (an unknown method name is kept as code -1: it cannot be written back)
            assert elf_amd.request_seq_to_json(q, ts, seq) == want, t
    assert accepted > 300 and refused > 300


def test_live_differential_of_client_messages(built, wire):
    """random GuardedRecords sessions against the real Records object (oracle/_ref): thread ids up to 5000 in random order (the
    unordered_map rehashes several times and keeps its buckets across clear()), repeated updates, dumps with and without records"""
    if not RefSelfPlay.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import numpy as np
    import elf_amd
    R = RefSelfPlay(9)
    rng = np.random.default_rng(12)
    for ses in range(4):
        ident = "client-%d" % ses
        R.client_reset(ident)
        c = elf_amd.ClientRecords(ident)
        for rnd in range(6):
            k = int(rng.choice([0, 3, 40, 700, 2500]))
            ids = rng.integers(0, 5000, size=k)
            for t in ids:
                st = (int(t), int(rng.integers(1, 50)), int(rng.integers(-1, 400)), int(rng.integers(0, 90)), int(rng.integers(-1, 90)))
                R.client_update_state(*st)
                c.update_state(*st)
            for _ in range(int(rng.integers(0, 3))):
                r = wire["records"][int(rng.integers(0, len(wire["records"])))]
                R.client_feed(r)
                c.feed(r)
            assert c.dump_and_clear() == R.client_dump(), (ses, rnd)
        c.close()
