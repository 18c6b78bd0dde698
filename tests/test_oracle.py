"""CPU: pins the C restatement (oracle/go_oracle.c) against the committed golden vectors generated
from the real reference, and -- where oracle/_ref is present -- against the real reference move by move."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import Port, Ref, playout_seeds


def unpack(bits, n):
    return np.unpackbits(bits)[:n]


@pytest.fixture(scope="module")
def port19(built):
    return Port(19)


@pytest.fixture(scope="module")
def port9(built):
    return Port(9)


def test_config1_sgf_every_ply(port19):
    g = np.load(os.path.join(GOLDEN, "sgf_406844.npz"))
    P = port19
    s = P.new()
    feat_at = {int(p): i for i, p in enumerate(g["feat_ply"])}
    for i in range(len(g["moves"]) + 1):
        assert P.hash(s) == int(g["hash"][i])
        assert np.array_equal(P.info(s), g["info"][i])
        assert np.array_equal(P.legal_mask(s), unpack(g["mask"][i], 362))
        assert P.evaluate(s, 7.5) == g["value"][i]
        if i in feat_at:
            for d4 in range(8):
                want = unpack(g["feat"][feat_at[i]][d4], 18 * 361).reshape(18, 19, 19).astype(np.float32)
                assert np.array_equal(P.extract_agz(s, d4), want)
        if i < len(g["moves"]):
            assert P.forward(s, int(g["moves"][i])) == 1
    assert P.hash(s) == 0x63C1B2F803BBCEAE and P.info(s)[0] == 201  # SURVEY.md 8c golden facts
    assert int(P.legal_mask(s)[:361].sum()) == 190


def test_ladder_suite_replays(port19):
    g = np.load(os.path.join(GOLDEN, "ladder_suite.npz"))
    P = port19
    assert len(g["names"]) == 115
    for k in range(len(g["names"])):
        s = P.new()
        for c in g["moves"][g["offsets"][k]:g["offsets"][k + 1]]:
            assert P.forward(s, int(c)) == 1
        assert P.hash(s) == int(g["final_hash"][k]) and P.info(s)[0] == g["final_ply"][k]
        assert np.array_equal(P.legal_mask(s), unpack(g["final_mask"][k], 362))
        want = unpack(g["final_feat"][k], 18 * 361).reshape(18, 19, 19).astype(np.float32)
        assert np.array_equal(P.extract_agz(s, 3), want)
        P.free(s)


@pytest.mark.parametrize("n,count", [(19, 48), (9, 256)])
def test_playout_protocol_golden(built, n, count):
    g = np.load(os.path.join(GOLDEN, "playout_%d.npz" % n))
    P = Port(n)
    assert np.array_equal(g["seeds"], playout_seeds(len(g["seeds"])))
    for i in range(count):
        s = P.new()
        mv = P.playout_moves(s, int(g["seeds"][i]))
        h = P.hash(s)
        assert (h & 0xFFFFFFFF, h >> 32, P.info(s)[0], len(mv)) == tuple(int(v) for v in g["out"][i])
        assert P.terminated(s)
        P.free(s)


def test_survey_golden_facts(port19):
    # D4, Q16, D16 -> ply 4, hash f9e27e42e80a3d73, 358 legal points, AGZ sum 367
    P = port19
    s = P.new()
    for x, y in ((3, 3), (15, 15), (3, 15)):
        assert P.forward(s, (y + 1) * 21 + x + 1) == 1
    assert P.info(s)[0] == 4 and P.hash(s) == 0xF9E27E42E80A3D73
    assert int(P.legal_mask(s)[:361].sum()) == 358
    assert P.extract_agz(s, 0).sum() == 367


@pytest.mark.parametrize("n,games", [(19, 6), (9, 40)])
def test_port_vs_real_reference_move_by_move(built, n, games):
    if not Ref.available(n):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    R, P = Ref(n), Port(n)
    for seed in playout_seeds(games, base=1000):
        r, p = R.new(), P.new()
        mv = R.playout_moves(R.clone(r), int(seed))
        assert np.array_equal(mv, P.playout_moves(P.clone(p), int(seed)))
        for i, c in enumerate(mv):
            assert R.forward(r, c) == 1 and P.forward(p, c) == 1
            assert R.hash(r) == P.hash(p) and np.array_equal(R.info(r), P.info(p))
            assert R.terminated(r) == P.terminated(p)
            if i % 5 == 0:
                assert np.array_equal(R.legal_mask(r), P.legal_mask(p))
                assert all(np.array_equal(a, b) for a, b in zip(R.board(r), P.board(p)))
                assert np.array_equal(R.extract_agz(r, i % 8), P.extract_agz(p, i % 8))
                assert R.evaluate(r, 7.5) == P.evaluate(p, 7.5)
        # illegal / special moves on the final position behave identically
        for c in (3, 2, 4, 1, 0, 5, (n + 2) ** 2 + 3):
            rr, pp = R.clone(r), P.clone(p)
            assert R.forward(rr, c) == P.forward(pp, c)
            assert R.hash(rr) == P.hash(pp) and R.terminated(rr) == P.terminated(pp)
            R.free(rr); P.free(pp)
        R.free(r); P.free(p)


def test_action_map_matches_reference(built):
    for n in (19, 9):
        P = Port(n)
        R = Ref(n) if Ref.available(n) else None
        for d4 in range(8):
            seen = set()
            for a in range(n * n + 1):
                c = P.action2coord(d4, a)
                assert P.coord2action(d4, c) == a
                seen.add(c)
                if R is not None:
                    assert R.action2coord(d4, a) == c and R.coord2action(d4, c) == a
            assert len(seen) == n * n + 1
