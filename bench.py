#!/usr/bin/env python
"""bench.py -- driver contract. One "step" = one pass of the hot path over one batch:
4096 concurrent 19x19 boards (per GPU) played from the empty board to game end by the config-2
policy (BASELINE.json configs[1]), whole games inside one k_playout launch.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  value = board steps/s summed over all ranks (weak scaling:
independent boards per GPU, no data-path collective -- SURVEY.md 8e).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEP_BYTES = {19: 8730, 9: 4450}  # reference state in + out + legal mask, per board step


def seeds_for(rank, boards, rep):
    # s_b = 0x9E3779B9*b + 1; b made unique per rank and per timed step
    b = np.arange(boards, dtype=np.uint64) + np.uint64((rank * 1000003 + rep) * boards)
    return b * np.uint64(0x9E3779B9) + np.uint64(1)


def cpu_baseline(n, budget_s=12.0):
    """Reference (oracle/_ref, the real ELF board engine) or port timed on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        from pyoracle import Port, Ref, playout_seeds
    except Exception as e:  # checker missing: report, never substitute
        return {"value": None, "unit": "board_steps/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    if Ref.available(n):
        R = Ref(n)
        t0 = time.time()
        tot, _ = R.playout(playout_seeds(cores * 2), threads=cores)  # calibration
        rate = tot / max(time.time() - t0, 1e-6)
        games = int(max(cores * 4, min(4096, rate * budget_s / 455.0)))
        games -= games % cores
        t0 = time.time()
        tot, _ = R.playout(playout_seeds(games), threads=cores)
        dt = time.time() - t0
        return {"value": tot / dt, "unit": "board_steps/s", "cores": cores, "kind": "reference",
                "per_core": tot / dt / cores,
                "sample": "%d of the same 19x19 config-2 games (%d board steps) on %d host threads, %.1f s" % (games, tot, cores, dt)}
    P = Port(n)
    t0, tot, games = time.time(), 0, 0
    while time.time() - t0 < budget_s:
        s = P.new()
        tot += len(P.playout_moves(s, int(playout_seeds(1, base=games)[0])))
        P.free(s)
        games += 1
    dt = time.time() - t0
    return {"value": tot / dt, "unit": "board_steps/s", "cores": 1, "kind": "port",
            "sample": "%d config-2 games (%d board steps), single thread, %.1f s" % (games, tot, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--board-size", type=int, default=19)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import elf_amd
    n, boards = args.board_size, args.boards
    eng = elf_amd.GoEngine(n, boards, local_rank)
    dev = eng.device
    out = torch.empty((boards, 4), dtype=torch.int32, device=dev)
    total = args.warmup + args.steps
    # inputs resident in HBM before the timed region
    seeds = [torch.from_numpy(seeds_for(rank, boards, r).view(np.int64)).to(dev) for r in range(total)]
    step_counts = torch.zeros(total, dtype=torch.int64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(total)]

    def one(r):
        eng.reset()                       # GoState::reset for every board
        ev[r][0].record()
        eng.playout(seeds[r], out=out)    # the dominant kernel, on torch's current stream
        ev[r][1].record()
        step_counts[r] = out[:, 3].to(torch.int64).sum()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for r in range(args.warmup):
        one(r)
    barrier()
    t0 = time.perf_counter()
    for r in range(args.warmup, total):
        one(r)
    barrier()
    dt = time.perf_counter() - t0

    counts = step_counts.cpu().numpy()
    my_steps = int(counts[args.warmup:].sum())
    kern_ms = [ev[r][0].elapsed_time(ev[r][1]) for r in range(args.warmup, total)]
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    s_all = torch.tensor([my_steps], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(s_all, op=dist.ReduceOp.SUM)
    dt_max, steps_all = float(t_all.item()), int(s_all.item())

    if rank == 0:
        avg_kernel_s = float(np.mean(kern_ms)) / 1e3
        steps_per_launch = my_steps / args.steps
        achieved = steps_per_launch * STEP_BYTES[n] / avg_kernel_s / 1e9
        res = {
            "metric": "board_steps_per_sec (19x19 GoState::forward + legal-move mask, random legal play to game end)"
            if n == 19 else "board_steps_per_sec (9x9)",
            "value": steps_all / dt_max,
            "unit": "board_steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d concurrent %dx%d boards per GPU, config-2 random legal non-eye play "
                                   "to game end, board-step kernel only (no net)" % (boards, n, n),
                       "boards_per_gpu": boards, "board_size": n, "board_steps_per_pass": steps_per_launch,
                       "parallelism": "independent boards per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_playout<%d>" % n, "avg_kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_step": STEP_BYTES[n],
                         "note": "algorithmic bytes = reference Board in+out + legal mask per step (SURVEY.md 8d); the kernel "
                                 "keeps the position in LDS for the whole game, so this is a rate against the HBM roof, not "
                                 "HBM traffic (PMC traffic in profiles/)"},
        }
        res["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(n)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
