"""Hazard H2 measured on the CPU (no GPU needed): the REAL reference stack (oracle/_ref/libelfsp19.so) against the restatement
that uses the engine's documented backup order (oracle/mcts_oracle.cc, first occurrence), stub net with an UN-quantised value
head (salt bit 31), config-3 search settings.  The HIP engine equals the restatement bit for bit on such runs
(tests/test_gpu_mcts.py::test_backup_order_is_first_occurrence_with_unquantised_values), so this is the engine-vs-reference
divergence of everything a backup order can touch.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import real_net_parity as rp  # noqa: E402
from pyoracle import PortSelfPlay, RefSelfPlay  # noqa: E402


def one(n, seed, rollouts, moves):
    cfg = rp.search_cfg(rollouts_per_thread=rollouts, seed=seed, net_salt=0x80000000 | 41, max_searches=moves)
    r = RefSelfPlay(n).run(**cfg)
    p = PortSelfPlay(n).run(**cfg)
    A = {0: [rp._tuple(r["search"][i], r["coord"][i], r["visits"][i], r["prior"][i], r["reward"][i]) for i in range(len(r["search"]))]}
    B = {0: [rp._tuple(p["search"][i], p["coord"][i], p["visits"][i], p["prior"][i], p["reward"][i]) for i in range(len(p["search"]))]}
    return rp.compare(A, B, 1, moves)


def main():
    n = 19
    plan = [(512, 8, range(100, 100 + int(sys.argv[1]) if len(sys.argv) > 1 else 132)), (8192, 2, range(500, 500 + (int(sys.argv[2]) if len(sys.argv) > 2 else 3)))]
    out = []
    for rollouts, moves, seeds in plan:
        agg = dict(rollouts=rollouts, moves_per_game=moves, games=0, searches_compared=0, bit_equal=0, reward_ulps_only=0,
                   decision_diverged=0, max_reward_ulps=0, first_decision_divergence=[])
        t0 = time.time()
        for s in seeds:
            c = one(n, s, rollouts, moves)
            agg["games"] += 1
            for k in ("searches_compared", "bit_equal", "reward_ulps_only", "decision_diverged"):
                agg[k] += c[k]
            agg["max_reward_ulps"] = max(agg["max_reward_ulps"], c["max_reward_ulps"])
            f = c["first_difference_per_game"][0]
            if f and f["kind"] != "reward_ulps":
                agg["first_decision_divergence"].append(dict(seed=s, **f))
        agg["seconds"] = round(time.time() - t0, 1)
        out.append(agg)
        sys.stderr.write(json.dumps(agg) + "\n")
    print(json.dumps(dict(what="H2: reference (heap-address backup order) vs first-occurrence order, stub net with un-quantised values, "
                               "19x19, bs 16, puct 1.5, vl 1, eps 0.25 / alpha 0.03, persistent tree", runs=out)))


if __name__ == "__main__":
    main()
