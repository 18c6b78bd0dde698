"""The trainer's replay buffer and GoGameTrain::act's draws (elfrq_*, host-only code of libelf_amd.so) against the REAL reference:
tests/golden/train_act_*.npz hold the "train" batches one real GoGameTrain game thread produced from a real ReaderQueuesT<Record>
(oracle/ref_selfplay.cc reftrain_act, oracle/gen_golden_train.py).  CPU: the draws (record, move, D4 code); the rows themselves are
compared on the GPU (tests/test_gpu_train.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from pyoracle import RefSelfPlay, sgfstr2coords

CASES = ["train_act_9", "train_act_9_evict", "train_act_19"]


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [int(v) for v in g["cfg_vals"]]))
    return g, cfg, int(g["board_size"]), [str(t) for t in g["records"]]


def fill(queues, n, recs):
    """TrainCtrl::OnReceive: InsertWithParity(record, rng, reward > 0), in arrival order; the handle of record i is i"""
    for i, t in enumerate(recs):
        j = json.loads(t)
        queues.insert(i, len(sgfstr2coords(n, j["result"]["content"])), j["result"]["reward"] > 0)


@pytest.mark.parametrize("name", CASES)
def test_draws_equal_the_reference_game_thread(built, name):
    import elf_amd
    g, cfg, n, recs = load_case(name)
    q = elf_amd.ReaderQueues(cfg["num_reader"], cfg["q_min_size"], cfg["q_max_size"], cfg["insert_seed"], num_threads=1, seed=cfg["game_seed"])
    fill(q, n, recs)
    slot, move_to, d4 = q.draw(cfg["num_acts"], cfg["num_future_actions"])
    assert np.array_equal(slot, g["rec"]), "records drawn (queue pair, parity, index)"
    assert np.array_equal(d4, g["aug_code"]), "generateD4Code"
    # extractMoveIdx is getPly() - 1 = move_to, except for the one record whose illegal moves the replay skips (the last one)
    ok = slot != len(recs) - 1
    assert ok.sum() > 32 and np.array_equal(move_to[ok], g["move_idx"][ok]), "switchRandomMove"
    # every drawn record is long enough for num_future_actions, and short ones exist in the buffer (they were drawn again)
    lens = np.array([len(sgfstr2coords(n, json.loads(t)["result"]["content"])) for t in recs])
    assert (lens[slot] >= cfg["num_future_actions"]).all()
    sizes = q.sizes()
    assert sizes.max() <= cfg["q_max_size"] and sizes.sum() == min(len(recs), sizes.sum())
    if name == "train_act_9_evict":
        assert sizes.sum() < len(recs)           # the even queue overflowed: its oldest records are gone ...
        assert 0 not in set(slot.tolist())       # ... and are never drawn
    q.close()


def test_parity_keeps_black_and_white_wins_balanced(built):
    """getSamplerWithParity: whatever the share of Black's wins in the buffer (here 1 in 10), the odd (Black won) queues are
    drawn 45 .. 55 % of the time (kSafeMargin, shared_reader.h:256-269)"""
    import elf_amd
    q = elf_amd.ReaderQueues(4, 1, 1000, insert_seed=9, num_threads=4, seed=21)
    for i in range(400):
        q.insert(i, 30, i % 10 == 0)
    sizes = q.sizes()
    assert sizes[1::2].sum() == 40 and sizes[0::2].sum() == 360
    slot, move_to, d4 = q.draw(64, 1)
    black = (slot % 10 == 0).mean()
    assert 0.40 < black < 0.50, black            # 1 - even_ratio = 0.45 of the draws
    assert move_to.min() == 0 and move_to.max() == 29 and set(d4.tolist()) == set(range(8))
    q.close()


def test_queue_arguments(built):
    import elf_amd
    L = elf_amd.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.elfrq_create(3, 1, 10, 1, C.byref(h)) == -1          # num_reader must be even (shared_reader.h:178)
    assert L.elfrq_create(2, 0, 10, 1, C.byref(h)) == -1
    q = elf_amd.ReaderQueues(2, 2, 10, insert_seed=1)
    q.insert(0, 20, True)
    q.insert(1, 20, False)
    with pytest.raises(elf_amd.ElfGoError):                        # a queue below queue_min_size: the reference waits, we refuse
        q.draw(1, 1)
    q.insert(2, 20, True)
    q.insert(3, 20, False)
    with pytest.raises(elf_amd.ElfGoError):                        # no record long enough for 21 future actions
        q.draw(1, 22)
    s, m, d = q.draw(1, 3)
    assert s.shape == (64,) and m.max() <= 17
    q.close()


def test_reference_reproduces_fixture():
    g, cfg, n, recs = load_case("train_act_9_evict")
    if not RefSelfPlay.available(n):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    a = RefSelfPlay(n).train_act(recs, **cfg)
    assert np.array_equal(a["selfplay_ver"] - 1000, g["rec"]) and np.array_equal(a["move_idx"], g["move_idx"])


def test_live_differential_over_queue_shapes_and_seeds(built):
    """the same comparison live (oracle/_ref present): other numbers of queues, queue bounds small enough to evict, other seeds,
    num_future_actions 1..3 -- draws of one real GoGameTrain thread vs elfrq_draw"""
    if not RefSelfPlay.available(9):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import elf_amd
    g, _, n, recs = load_case("train_act_9")
    R = RefSelfPlay(n)
    lens = [len(sgfstr2coords(n, json.loads(t)["result"]["content"])) for t in recs]
    rng = np.random.default_rng(8)
    done = 0
    for trial in range(12):
        cfg = dict(num_reader=int(rng.choice([2, 4, 6])), q_min_size=1, q_max_size=int(rng.choice([3, 5, 1000])),
                   insert_seed=int(rng.integers(1, 1 << 30)), game_seed=int(rng.integers(1, 1 << 30)), num_acts=2,
                   num_future_actions=int(rng.integers(1, 4)))
        try:
            a = R.train_act(recs, **cfg)
        except RuntimeError:
            continue                      # a queue stayed below q_min_size: the reference would wait for more data
        q = elf_amd.ReaderQueues(cfg["num_reader"], cfg["q_min_size"], cfg["q_max_size"], cfg["insert_seed"], num_threads=1, seed=cfg["game_seed"])
        fill(q, n, recs)
        slot, move_to, d4 = q.draw(cfg["num_acts"], cfg["num_future_actions"])
        assert np.array_equal(slot, a["selfplay_ver"] - 1000), cfg
        assert np.array_equal(d4, a["aug_code"]), cfg
        ok = slot != len(recs) - 1
        assert np.array_equal(move_to[ok], a["move_idx"][ok]), cfg
        assert all(lens[s_] >= cfg["num_future_actions"] for s_ in slot)
        q.close()
        done += 1
    assert done >= 6


def test_draws_do_not_depend_on_the_host_threads_that_make_them(built):
    """elfrq_draw spreads the acts of different game threads (each with its own generator) over the host worker pool: the arrays
    equal the one-thread result, act for act, also when the acts do not divide evenly among the game threads."""
    import numpy as np
    from elf_amd.train import ReaderQueues
    outs = []
    for host_threads in ("1", "8", "3"):
        os.environ["ELF_AMD_HOST_THREADS"] = host_threads
        try:
            q = ReaderQueues(num_reader=4, queue_min_size=2, queue_max_size=50, insert_seed=5, num_threads=5, seed=21, job_id="")
            rng = np.random.default_rng(3)
            for s in range(40):
                q.insert(s, int(rng.integers(1, 200)), bool(rng.integers(0, 2)))
            a = q.draw(7, 2)            # fewer acts than the pool's threshold: serial path
            b = q.draw(23, 2)           # 23 acts over 5 game threads, starting at thread 7 % 5
            c = q.draw(16, 1)
            outs.append(np.concatenate([np.asarray(x).ravel() for x in (*a, *b, *c)]))
            q.close()
        finally:
            os.environ.pop("ELF_AMD_HOST_THREADS", None)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_draws_from_two_host_threads_at_once(built):
    """Two replay buffers drawn from concurrently by two Python threads (ctypes releases the GIL): the shared host worker pool
    serialises its parallel regions, nothing deadlocks, and each buffer's draws equal what it draws alone."""
    import threading
    import numpy as np
    from elf_amd.train import ReaderQueues

    def make(seed):
        q = ReaderQueues(num_reader=4, queue_min_size=2, queue_max_size=100, insert_seed=seed, num_threads=8, seed=100 + seed, job_id="")
        rng = np.random.default_rng(seed)
        for s in range(80):
            q.insert(s, int(rng.integers(2, 300)), bool(rng.integers(0, 2)))
        return q

    def run(q, out, reps):
        for _ in range(reps):
            out.append(np.concatenate([np.asarray(x).ravel() for x in q.draw(32, 1)]))

    alone = []
    for seed in (1, 2):
        q = make(seed)
        o = []
        run(q, o, 3)
        alone.append(o)
        q.close()
    qs = [make(1), make(2)]
    outs = [[], []]
    th = [threading.Thread(target=run, args=(qs[i], outs[i], 3)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive(), "a draw hung"
    for i in range(2):
        assert len(outs[i]) == 3 and all(np.array_equal(a, b) for a, b in zip(outs[i], alone[i]))
        qs[i].close()
