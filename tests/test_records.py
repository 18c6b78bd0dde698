"""CPU: the record format half of libelf_amd.so (elfrec_*, host-only code; no GPU needed): SGF move strings, policy
quantisation and the Record JSON text, against the records the REAL reference dumped (tests/golden/records_*.npz,
oracle/gen_golden_train.py) and, where oracle/_ref is present, against the reference directly."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import RefSelfPlay, sgfstr2coords

CASES = ["records_9_cutoff", "records_9_resign", "records_9_twopass", "records_9_neverresign", "records_9_preload", "records_19_resign", "records_19_cutoff", "records_19_sgf_preload",
         "records_9_eval", "records_9_eval_swap_resign", "records_9_req2_restart",
         "records_9_cheat_selfplay", "records_9_cheat_eval", "records_9_cheat_eval_swap", "online_9_script", "online_9_following_pass"]


def sp_options(elf_amd, n, cfg, num_games=1):
    from elf_amd.selfplay import MctsOptions, SpOptions
    mo = MctsOptions(int(cfg["rollouts_per_batch"]), int(cfg["virtual_loss"]), int(cfg["use_prior"]), int(cfg["unexplored_q_zero"]),
                     int(cfg["root_unexplored_q_zero"]), float(cfg["c_puct"]), float(cfg["komi"]), int(cfg["ply_pass_enabled"]), 1, 1, 1, 0, -1)
    return SpOptions(n, num_games, 1024, int(cfg["rollouts_per_thread"]), int(cfg["persistent_tree"]), float(cfg["root_epsilon"]),
                     float(cfg["root_alpha"]), int(cfg["seed"]), int(cfg["policy_distri_cutoff"]), int(cfg["move_cutoff"]),
                     float(cfg["resign_thres"]), float(cfg["never_resign_prob"]), 0, 1, 0, 0, 0, 0, mo)


def to_json(elf_amd, opt, p, j):
    from elf_amd.selfplay import SpRequest
    L = elf_amd.lib()
    mv, pol, val = p["moves"], p["policies"], p["values"]
    args = (mv.ctypes.data, mv.size, pol.ctypes.data if pol.size else None, pol.shape[0], val.ctypes.data, val.size,
            C.c_float(p["reward"]), int(j["result"]["black_never_resign"]), j["seq"], j["thread_id"], j["timestamp"])
    vers, ctrl = j["request"]["vers"], j["request"]["client_ctrl"]
    if vers["white_ver"] >= 0 or ctrl["player_swap"] or vers["black_ver"] != 0:
        # evaluation games, a later request: the record carries the request it was played under (elfrec_record_to_json2)
        q = SpRequest(vers["black_ver"], vers["white_ver"], ctrl["black_resign_thres"], ctrl["white_resign_thres"], ctrl["never_resign_prob"],
                      ctrl["num_game_thread_used"], int(ctrl["player_swap"]), int(ctrl["async"]))
        fn, head = L.elfrec_record_to_json2, (C.byref(opt), C.byref(q))
    else:
        fn, head = L.elfrec_record_to_json, (C.byref(opt),)
    n = fn(*head, *args, None, 0)
    assert n > 0
    buf = C.create_string_buffer(n + 1)
    assert fn(*head, *args, buf, n + 1) == n
    assert fn(*head, *args, buf, n) == -2   # ELFGO_E_BADSIZE
    return buf.raw[:n].decode()


@pytest.mark.parametrize("name", CASES)
def test_record_json_text_equals_reference_dump(built, name):
    """parse the reference's record, re-serialise it through elfrec_record_to_json: the text must be identical."""
    import elf_amd
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["board_size"])
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    opt = sp_options(elf_amd, n, cfg)
    for t in g["records"]:
        t = str(t)
        j = json.loads(t)
        p = elf_amd.parse_record(n, t)
        assert np.array_equal(p["moves"], sgfstr2coords(n, j["result"]["content"]))
        assert elf_amd.coords_to_sgfstr(n, p["moves"]) == j["result"]["content"]
        assert to_json(elf_amd, opt, p, j) == t


def test_sgf_strings_edge_cases(built):
    import elf_amd
    n = 19
    S = n + 2
    assert elf_amd.coords_to_sgfstr(n, []) == "()"
    assert elf_amd.coords_to_sgfstr(n, [0, 1 * S + 1, 19 * S + 19]) == "(;B[];W[aa];B[ss])"   # pass, A1 corner, far corner
    for s, want in [("", []), ("(", []), ("x(;B[aa])", []), ("(;B[aa", []), ("(;B[aa];W[bb])", [S + 1, 2 * S + 2]),
                    ("(;B[aa]W[bb])", [S + 1]),                      # the loop stops at the first non-';' (sgf.h:104-106)
                    ("(;B[];W[a];B[tt];W[ t];B[a b])", [0, 0, 3, 3, 2 * S + 1])]:   # <2 chars = pass; off board / blank = M_INVALID
        got = elf_amd.sgfstr_to_coords(n, s)
        assert list(got) == want, (s, list(got))
        assert list(sgfstr2coords(n, s)) == want, s


def test_quantise_policy(built):
    """GoStateExt::addMCTSPolicy: c = (unsigned char)(p / max * 255), fp32 arithmetic (go_state_ext.h:158-181)"""
    import elf_amd
    L = elf_amd.lib()
    rng = np.random.default_rng(3)
    for n in (9, 19):
        P = (n + 2) ** 2
        for _ in range(50):
            k = int(rng.integers(1, 60))
            coord = rng.choice(P, size=k, replace=False).astype(np.int32)
            visits = rng.integers(0, 500, size=k).astype(np.float32)
            visits[rng.integers(0, k)] += 1
            prob = (visits / visits.sum(dtype=np.float32)).astype(np.float32)
            out = np.full(P, 7, np.uint8)
            assert L.elfrec_quantise_policy(n, coord.ctypes.data, prob.ctypes.data, k, out.ctypes.data) == 0
            want = np.zeros(P, np.uint8)
            mx = np.float32(prob.max())
            want[coord] = (np.float32(prob / mx) * np.float32(255)).astype(np.uint8)
            assert np.array_equal(out, want)
    bad = np.array([P], np.int32)
    one = np.array([1.0], np.float32)
    assert L.elfrec_quantise_policy(19, bad.ctypes.data, one.ctypes.data, 1, out.ctypes.data) == -1


def test_against_reference_library(built):
    if not RefSelfPlay.available(19):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import elf_amd
    R = RefSelfPlay(19)
    rng = np.random.default_rng(0)
    for _ in range(100):
        k = int(rng.integers(0, 80))
        coords = [0 if rng.random() < 0.1 else int((rng.integers(0, 19) + 1) * 21 + rng.integers(0, 19) + 1) for _ in range(k)]
        s = R.coords2sgfstr(coords)
        assert elf_amd.coords_to_sgfstr(19, coords) == s
        assert np.array_equal(R.sgfstr2coords(s), elf_amd.sgfstr_to_coords(19, s))
    g = np.load(os.path.join(GOLDEN, "records_19_cutoff.npz"))
    for t in g["records"]:
        assert R.record_roundtrip(str(t)) == str(t)   # the reference reads its own dump back to the same text


def _values_text(elf_amd, vals):
    import re
    g = np.load(os.path.join(GOLDEN, "records_9_cutoff.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    opt = sp_options(elf_amd, 9, cfg)
    L = elf_amd.lib()
    v = np.ascontiguousarray(vals, np.float32)
    args = (C.byref(opt), None, 0, None, 0, v.ctypes.data, v.size, C.c_float(0.5), 0, 2, 0, 0)
    n = L.elfrec_record_to_json(*args, None, 0)
    buf = C.create_string_buffer(n + 1)
    L.elfrec_record_to_json(*args, buf, n + 1)
    t = buf.raw[:n].decode()
    return t, re.search(r'"values":\[(.*?)\]', t).group(1).split(",")


def test_float_text_is_nlohmann_grisu2(built):
    """Floats are printed with Grisu2 in nlohmann's layout, not with the shortest-round-trip digits of std::to_chars (they differ on
    ~0.7 % of values): 296 float32 values over the whole range, with the text the reference's json::dump() gave for them."""
    import elf_amd
    g = np.load(os.path.join(GOLDEN, "json_float_text.npz"))
    vals = g["bits"].view(np.float32)
    _, got = _values_text(elf_amd, vals)
    assert got == [str(t) for t in g["text"]]
    assert "0.7005996704101563" in got and "-0.0" in got and "16777216.0" in got and "1.401298464324817e-45" in got


def test_float_text_fuzz_against_reference(built):
    if not RefSelfPlay.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import elf_amd
    R = RefSelfPlay(9)
    rng = np.random.default_rng(7)
    for rep in range(12):
        if rep % 2 == 0:
            vals = rng.integers(0, 2 ** 32, size=3000, dtype=np.uint64).astype(np.uint32).view(np.float32)
            vals = vals[np.isfinite(vals)]
        else:
            vals = (np.tanh(rng.standard_normal(3000)) * rng.choice([1, 1e-3, 1e3, 1e-6])).astype(np.float32)
        t, _ = _values_text(elf_amd, vals)
        assert R.record_roundtrip(t) == t      # the reference parses our text and dumps the same text


@pytest.mark.parametrize("name", ["records_9_sgf", "records_9_sgf_policy_only"])
def test_game_sgf_equals_reference_dumpSgf(built, name):
    """GameOptions.dump_record_prefix: the SGF text finish_game writes for a finished game (GoStateExt::dumpSgf,
    go_state_ext.cc:26-82) rebuilt from the game's Record: RE[] ("B+R" / "W+R" / margin), PB / PW with "(policy only)", KM, every
    move with its predicted value -- equal to what the reference's dumpSgf returned for the same game."""
    import elf_amd
    from elf_amd.train import record_to_sgf, sgf_file_name
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(g["board_size"])
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    opt = sp_options(elf_amd, n, cfg)
    opt.black_use_policy_network_only, opt.white_use_policy_network_only = int(cfg["black_policy_only"]), int(cfg["white_policy_only"])
    assert len(g["sgfs"]) == len(g["records"]) >= 2
    for t, want in zip(g["records"], g["sgfs"]):
        t, want = str(t), str(want)
        fname = sgf_file_name("game", t)
        assert ("Filename: " + fname + "\n") in want                      # <prefix>_<game>_<seq>_<B|W>.sgf
        got = record_to_sgf(n, t, opt, fname, git_hash="GIT_COMMIT_HASH", git_staged="GIT_STAGED")   # the harness build's two lines
        assert got == want
    ours = record_to_sgf(n, str(g["records"][0]), opt, "x.sgf")
    assert "Git hash: elf_amd" in ours and "Staged: 0" in ours
    if name == "records_9_sgf":
        assert any("RE[B+R]" in str(x) for x in g["sgfs"]) and any("RE[W+R]" in str(x) for x in g["sgfs"]) and "KM[6.5]" in str(g["sgfs"][0])
    else:
        assert all("RE[W+6.500000]" in str(x) and "PW[MCTS(policy only)]" in str(x) for x in g["sgfs"])


def test_game_sgf_result_strings(built):
    """RE[] for values the fixtures do not reach, against the reference's dumpSgf where oracle/_ref is present: |value| == 1 is a
    resignation, anything else a margin printed with std::to_string; komi through an ostream"""
    import elf_amd
    from elf_amd.train import record_to_sgf
    g = np.load(os.path.join(GOLDEN, "records_9_sgf.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
    R = RefSelfPlay(9) if RefSelfPlay.available(9) else None
    for v, komi, want in ((1.5, 7.5, "RE[B+1.500000]"), (-1.5, 0.5, "RE[W+1.500000]"), (1.0, 7.5, "RE[B+R]"), (-1.0, 6.5, "RE[W+R]"),
                          (0.0, 7.0, "RE[W+-0.000000]"), (12.25, 7.125, "RE[B+12.250000]"), (-361.0, 1234567.0, "RE[W+361.000000]")):
        opt = sp_options(elf_amd, 9, cfg)
        opt.mcts.komi = komi
        rec = dict(result=dict(content="()", values=[], reward=v), thread_id=0, seq=1)
        got = record_to_sgf(9, rec, opt, "f.sgf", "GIT_COMMIT_HASH", "GIT_STAGED")
        assert want in got
        if R is not None:
            ref = R._text(R.L.refsp_sgf_of_value, C.c_float(v), C.c_float(komi))
            assert got == ref.replace("PB[Policy]PW[Policy]", "PB[MCTS]PW[MCTS]")      # the harness state has use_mcts = false
