"""Worker of tests/test_dist_gloo.py: the N>1 path of bench.py (rank/seed partition, barrier, max-time / sum-count
reduction, rank-0-only report) on CPU tensors over gloo."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    os.environ["ELF_BENCH_BACKEND"] = "gloo"
    rank, local_rank, world, dist = bench.init_dist(None)
    assert dist is not None and dist.get_world_size() == world
    dev = torch.device("cpu")
    boards = 64
    # every rank owns its own boards: seeds must be disjoint across ranks and timed steps
    mine = np.concatenate([bench.seeds_for(rank, boards, rep) for rep in range(3)])
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    flat = [x for g in gathered for x in g]
    assert len(set(flat)) == len(flat), "seed overlap between ranks"
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # ranks finish at different times: the report must use the slowest
    dist.barrier()
    dt = 0.05 * (rank + 1)
    my_units = 1000 + rank
    dt_max, total = bench.reduce_max_sum(dist, dev, dt, my_units)
    assert abs(dt_max - 0.05 * world) < 1e-9
    assert total == sum(1000 + r for r in range(world))
    if rank == 0:
        print(json.dumps({"world": world, "dt_max": dt_max, "total": total, "wall": time.perf_counter() - t0}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
