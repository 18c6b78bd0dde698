"""CPU: the SGF reader (elfrec_sgf_parse, host-only code of libelf_amd.so) = the reference's Sgf::load + iterator.
The four cases of the reference's own sgf/sgf_test.cc (gtest is absent in the image, so they are restated here: the texts are
the test's, the replay runs on the CPU oracle) and, where oracle/_ref is present, a differential against the real loader on
hand-made texts (comments with brackets, escapes, variations, setup stones, blanks inside moves) and on the ladder suite."""
import glob
import os

import numpy as np
import pytest

from pyoracle import Port, Ref

MAKE_SGF = ("(;CA[UTF-8]SZ[9]PB[Murakawa Daisuke]PW[Iyama Yuta]KM[6.5]HA[0]RE[W+1.5]GM[1];"
            "B[fd];W[cf];B[eg];W[dd];B[dc];W[cc];B[de];W[cd];B[ed];W[he];B[ce];W[be];B[df];W[bf];"
            "B[hd];W[ge];B[gd];W[gg];B[db];W[cb];B[cg];W[bg];B[gh];W[fh];B[hh];W[fg];B[eh];W[ei];"
            "B[di];W[fi];B[hg];W[dh];B[ch];W[ci];B[bh];W[ff];B[fe];W[hf];B[id];W[bi];B[ah];W[ef];"
            "B[dg];W[ee];B[di];W[ig];B[ai];W[ih];B[fb];W[hi];B[ag];W[ab];B[bd];W[bc];B[ae];W[ad];"
            "B[af];W[bd];B[ca];W[ba];B[da];W[ie])")
CHINESE = ("(;GM[1]FF[4]CA[UTF-8]AP[CGoban:3]ST[2]RU[Chinese]SZ[9]HA[2]RE[Void]KM[5.50]PW[test_white]PB[test_black]RE[B+39.50];"
           "B[gc];B[cg];W[ee];B[gg];W[eg];B[ge];W[ce];B[ec];W[cc];B[dd];W[de];B[cd];W[bd];B[bc];W[bb];B[be];"
           "W[ac];B[bf];W[dh];B[ch];W[ci];B[bi];W[di];B[ah];W[gh];B[hh];W[fh];B[hg];W[gi];B[fg];"
           "W[dg];B[ei];W[cf];B[ef];W[ff];B[fe];W[bg];B[bh];W[af];B[ag];W[ae];B[ad];W[ae];B[ed];"
           "W[db];B[df];W[eb];B[fb];W[ea];B[fa])")
JAPANESE = ("(;GM[1]FF[4]CA[UTF-8]AP[CGoban:3]ST[2]RU[Japanese]SZ[9]HA[2]RE[Void]KM[5.50]PW[test_white]PB[test_black]"
            "AB[gc][cg];W[ee];B[dg])")
FINAL = ["....OX...", ".O.OOX...", "O.O.X.X..", ".OXXX....", "OX...XX..", ".X.XXO...", "X.XOOXXX.", "XXXO.OOX.", ".XOOX.O.."]


def replay(port, players, coords):
    """the loop of sgf_test.cc: a pass is inserted when the entry's player is not the one to move (handicap stones)"""
    st = port.new()
    for p, c in zip(players, coords):
        if int(port.info(st)[1]) != int(p):
            assert port.forward(st, 0) == 1
        assert port.forward(st, int(c)) == 1, int(c)
    return st


def test_sgf_test_cc_cases(built):
    import elf_amd
    from elf_amd.train import parse_sgf
    port = Port(9)
    # testMakeSgf (sgf_test.cc:32-58): every move of the game is legal
    pl, mv, h = parse_sgf(9, MAKE_SGF)
    assert len(mv) == 62 and list(pl[:4]) == [1, 2, 1, 2] and (h["komi"], h["handi"], h["winner"], h["win_margin"]) == (6.5, 0, 2, 1.5)
    port.free(replay(port, pl, mv))
    # testSgfProps (:60-87): two Black moves in a row (handicap), komi 5.5
    pl, mv, h = parse_sgf(9, CHINESE)
    assert h["komi"] == 5.5 and h["handi"] == 2 and list(pl[:3]) == [1, 1, 2] and (h["winner"], h["win_margin"]) == (1, 39.5)
    # testChineseHandicap (:113-152): the final position
    st = replay(port, pl, mv)
    col, _ = port.board(st)                        # 0 empty, 1 Black, 2 White, index x * 9 + y
    for y, row in enumerate(FINAL):
        for x, ch in enumerate(row):
            assert col[x * 9 + y] == {".": 0, "X": 1, "O": 2}[ch], (x, y)
    port.free(st)
    # testJapaneseHandicap (:91-111): setup stones (AB) are not entries; White moves first
    pl, mv, h = parse_sgf(9, JAPANESE)
    assert list(pl) == [2, 1] and list(mv) == [5 * 11 + 5, 7 * 11 + 4]
    port.free(replay(port, pl, mv))


HAND_MADE = [
    "(;SZ[9]KM[6.5]RE[W+1.5];B[fd];W[cf]C[a ;B[aa\\] trick];B[];W[tt];B[ a\nb ](;W[cc])(;W[dd]))",
    "(;SZ[9]RE[B+R];B[aa];W[bb]C[x]W[cc];AB[dd][ee]B[ff];C[only a comment]W[gg])",
    "(;FF[4]\nSZ[9]\nKM[0.5]\n;B[ee]\n;W[ \n e f]\n;B[e]\n;W[])",
    ";SZ[9];B[aa]",
    "(;SZ[9]RE[b+12];W[ii];w[aa];B [bb];B[Ab])",
    "(;SZ[9]C[header \\] with ; and ) inside];B[cd])",
    "no sgf at all", "(;SZ[9])", "", "(;SZ[9];)",
]


def test_differential_against_the_reference_loader(built):
    if not Ref.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    from elf_amd.train import parse_sgf
    R = Ref(9)
    for t in HAND_MADE + [MAKE_SGF, CHINESE, JAPANESE]:
        want, got = R.sgf_parse(t), parse_sgf(9, t)
        if want is None:
            assert got is None, t
            continue
        assert got is not None, t
        wm, wp, wh = want
        # an entry without a move is uninitialised memory in the reference: compare the entries that have one
        has_move = got[0] != 0
        assert len(got[0]) == len(wm), t
        assert np.array_equal(got[1][has_move], wm[has_move]) and np.array_equal(got[0][has_move], wp[has_move]), t
        assert got[2] == pytest.approx(wh), t
    files = sorted(glob.glob("/root/reference/ladder_suite/*/*.sgf"))
    R19 = Ref(19)
    for f in files[:60]:
        text = open(f, encoding="latin-1").read()
        wm, wp, wh = R19.sgf_parse(text)
        pl, mv, h = parse_sgf(19, text)
        ok = pl != 0                    # (a trailing node without a move is uninitialised memory in the reference)
        assert len(mv) == len(wm) and np.array_equal(mv[ok], wm[ok]) and np.array_equal(pl[ok], wp[ok]) and h == pytest.approx(wh), f
        assert ok.sum() >= len(mv) - 1


def test_random_texts_against_the_reference_loader(built):
    """800 random SGF-like texts (blanks and newlines inside moves, lower-case and two-letter keys, comments with brackets and
    escapes, stray parentheses, ";(" -- where the reference calls the file corrupted and drops the rest): same entries, same header"""
    if not Ref.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    from elf_amd.train import parse_sgf
    R = Ref(9)
    rng = np.random.default_rng(1)
    letters = "abcdefghij t"

    def mv():
        if rng.integers(0, 6) == 0:
            return ""
        a, b = letters[rng.integers(0, len(letters))], letters[rng.integers(0, len(letters))]
        return ["", " ", "\n", "  "][rng.integers(0, 4)] + a + ["", " ", "\n"][rng.integers(0, 3)] + b + ["", " ", "\n"][rng.integers(0, 3)]

    pieces = [lambda: ";", lambda: "B[%s]" % mv(), lambda: "W[%s]" % mv(), lambda: "C[te;xt ) ( \\] more]", lambda: "AB[aa][bb]",
              lambda: "(", lambda: ")", lambda: " ", lambda: "\n", lambda: "B [%s]" % mv(), lambda: "b[aa]", lambda: "BB[aa]",
              lambda: ";B[%s]" % mv(), lambda: ";W[%s]" % mv(), lambda: "\nW[%s]" % mv()]
    for _ in range(800):
        t = "(;SZ[9]KM[%s]" % ["6.5", "0.5", " 7", "5.50"][rng.integers(0, 4)] + ["", "RE[B+R]", "RE[W+2.5]", "RE[b+10]", "HA[2]", "\nHA[3]"][rng.integers(0, 6)]
        for _ in range(rng.integers(1, 14)):
            t += pieces[rng.integers(0, len(pieces))]()
        want, got = R.sgf_parse(t), parse_sgf(9, t)
        assert (want is None) == (got is None), repr(t)
        if want is None:
            continue
        wm, wp, wh = want
        ok = got[0] != 0
        assert len(got[0]) == len(wm) and np.array_equal(got[1][ok], wm[ok]) and np.array_equal(got[0][ok], wp[ok]), repr(t)
        assert got[2] == pytest.approx(wh), repr(t)
