"""CPU: the trainer-side oracle.  The restatement (pyoracle.port_train_sample over the C port of the board engine) against the
committed golden rows tests/golden/train_*.npz, which oracle/gen_golden_train.py produced with the REAL reference
(GoStateExtOffline + GoFeature extractors compiled in place); and, where oracle/_ref is present, the reference again."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import Port, RefSelfPlay, port_train_sample, sgfstr2coords


def rows_of(n):
    g = np.load(os.path.join(GOLDEN, "train_%s.npz" % n))
    recs = [json.loads(str(t)) for t in g["records"]]
    return g, recs


def check_row(n, g, i, o):
    nfa = int(g["nfa"][i])
    s = np.unpackbits(g["s"][i])[: 18 * n * n].reshape(18, n, n).astype(np.float32)
    assert np.array_equal(o["s"], s), i
    assert np.array_equal(np.asarray(o["offline_a"])[:nfa], g["offline_a"][i][:nfa]), i
    assert np.float32(o["winner"]) == g["winner"][i] and int(o["move_idx"]) == g["move_idx"][i], i
    assert int(o["num_move"]) == g["num_move"][i] and int(o["aug_code"]) == g["aug_code"][i], i
    assert int(o["selfplay_ver"]) == g["selfplay_ver"][i], i
    assert np.float32(o["predicted_value"]) == g["predicted_value"][i], i
    np.testing.assert_array_equal(np.asarray(o["mcts_scores"], np.float32), g["mcts_scores"][i])   # NaN == NaN (all-zero policy row)


@pytest.mark.parametrize("n", [9, 19])
def test_restatement_matches_reference_rows(built, n):
    g, recs = rows_of(n)
    port = Port(n)
    assert len(g["rec"]) > 100
    for i in range(len(g["rec"])):
        o = port_train_sample(port, recs[int(g["rec"][i])], int(g["move_to"][i]), int(g["d4"][i]), int(g["nfa"][i]))
        check_row(n, g, i, o)


def test_restatement_on_a_record_that_continues_past_a_superko_repetition(built):
    """train_9_superko.npz (oracle/gen_golden_train.py --superko): the game ends by positional superko at move 107, the record holds
    32 more moves; GoState::forward refuses them (go_state.cc:78-79), so rows drawn beyond the repetition show the position of ply 108."""
    g, recs = rows_of("9_superko")
    port = Port(9)
    late = 0
    for i in range(len(g["rec"])):
        o = port_train_sample(port, recs[0], int(g["move_to"][i]), int(g["d4"][i]), int(g["nfa"][i]))
        check_row(9, g, i, o)
        if g["move_to"][i] > 107:
            late += 1
            assert g["move_idx"][i] == 107
    assert late >= 10


@pytest.mark.parametrize("n", [9, 19])
def test_reference_reproduces_rows(built, n):
    if not RefSelfPlay.available(n):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    g, _ = rows_of(n)
    R = RefSelfPlay(n)
    for i in range(0, len(g["rec"]), 3):
        o = R.train_sample(str(g["records"][int(g["rec"][i])]), int(g["move_to"][i]), int(g["d4"][i]), int(g["nfa"][i]))
        check_row(n, g, i, o)


def test_sgfstr2coords_restatement_on_fixture_records(built):
    for n in (9, 19):
        g, recs = rows_of(n)
        for j in recs:
            mv = sgfstr2coords(n, j["result"]["content"])
            assert len(mv) == j["result"]["num_move"]
            assert all(c == 0 or (1 <= c % (n + 2) <= n and 1 <= c // (n + 2) <= n) for c in mv)
