"""CPU: bench.py's multi-process path (one process per GPU, independent units per rank, max-over-ranks time,
sum-over-ranks count) with world_size 2 over gloo.  The data path has no collective."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    rep = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert rep["world"] == 2 and rep["total"] == 2001 and abs(rep["dt_max"] - 0.1) < 1e-9
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]   # only rank 0 reports


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run with two workers
    (one per GPU on a real node; the CPU stub workload over gloo here) and rank 0 prints one JSON line with n_gpus = 2, the
    slowest rank's time and the sum over ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "5", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["steps"] == 5 and rep["scaling"] == "weak"
    assert rep["config"]["units"] == 5 * (1000 + 1001) and rep["config"]["per_rank_units"] == [5000.0, 5005.0]
    assert rep["ms_per_step"] >= 4.0        # rank 1 sleeps 4 ms per step: the report carries the slowest rank
    # what a multi-GPU line says about scaling (the headline and the measured games/s carry the same block)
    sr = rep["config"]["scaling_report"]
    assert sr["n_gpus"] == 2 and len(sr["per_rank"]) == 2 and abs(sr["sum_over_ranks"] - sum(sr["per_rank"])) < 1e-6
    assert set(sr) >= {"per_rank", "sum_over_ranks", "n1_reference", "per_gpu_fraction_of_n1", "measured_curve"}
    assert "no multi-GPU node" in sr["measured_curve"]


import pytest


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_on_the_real_workloads():
    """The world > 1 branches of run_mcts / run_games on real kernels: two ranks on the one GPU of this box (ELF_BENCH_SHARE_GPU=1,
    process group over gloo), a small net and few rollouts.  Each rank plays its own games (game_idx_base = rank x games), the line
    carries the slowest rank's time, the sum of the rollouts and of the finished games, and the per-rank values."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "both", "--games", "32", "--groups", "2",
                        "--rollouts", "64", "--steps", "6", "--warmup", "2", "--net-blocks", "2", "--net-dim", "32",
                        "--games-rollouts", "16", "--games-cutoff", "6", "--games-generations", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["steps"] == 6 and rep["scaling"] == "weak"
    cfg = rep["config"]
    assert cfg["games_per_gpu"] == 32 and cfg["rollouts_per_step"] == 32 * 16
    # value = rollouts of BOTH ranks / the slowest rank's time
    assert abs(rep["value"] - 2 * 32 * 16 * 6 / (rep["ms_per_step"] * 6 / 1e3)) < 1e-6 * rep["value"]
    sr = cfg["scaling_report"]
    assert len(sr["per_rank"]) == 2 and all(v > 0 for v in sr["per_rank"])
    assert sr["process_group"]["backend"] == "gloo" and sr["process_group"]["ranks_share_one_gpu"] is True
    assert rep["cpu_baseline"] is None                      # timed on rank 0 at N = 1 only
    gm = rep["selfplay_games"]
    assert gm["n_gpus"] == 2 and len(gm["per_rank_games_per_sec"]) == 2
    assert gm["games_finished"] >= 2 * 32                   # both ranks finished at least one generation of their games


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("board", ["--boards", "256", "--steps", "2", "--warmup", "1"]),
                                            ("train", ["--train-batch", "256", "--train-prefetch", "2", "--train-records", "128", "--steps", "3", "--warmup", "1"])])
def test_two_ranks_share_the_gpu_on_the_board_and_trainer_workloads(workload, extra):
    """The world > 1 branches of run_board / run_train (per-rank seeds and records, max-over-ranks time, sum of the units) with two
    ranks on this box's one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--no-cpu-baseline"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["value"] > 0
    if workload == "board":
        assert rep["parity_mismatches"] == 0 and rep["config"]["boards_per_gpu"] == 256
    else:
        assert rep["config"]["samples_per_launch"] == 512
