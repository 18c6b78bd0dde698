"""CPU: bench.py's multi-process path (one process per GPU, independent units per rank, max-over-ranks time,
sum-over-ranks count) with world_size 2 over gloo.  The data path has no collective."""
import json
import os
import socket
import subprocess
import sys

import pytest

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


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run with two workers
    (one per GPU on a real node; the CPU stub workload over gloo here) and rank 0 prints one JSON line with n_gpus = 2, the
    slowest rank's time and the sum over ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_BACKEND"] = "gloo"
    env["ELF_BENCH_FULL"] = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "5", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rep, full = one_line(r.stdout, env)
    assert rep["n_gpus"] == 2 and rep["steps"] == 5 and rep["scaling"] == "weak"
    assert full["config"]["units"] == 5 * (1000 + 1001) and full["config"]["per_rank_units"] == [5000.0, 5005.0]
    assert rep["ms_per_step"] >= 4.0        # rank 1 sleeps 4 ms per step: the report carries the slowest rank
    # what a multi-GPU line says about scaling comes from THIS run's ranks only (no stored N = 1 number)
    sr = full["scaling_report"]
    assert sr["n_gpus"] == 2 and len(sr["per_rank"]) == 2 and abs(sr["sum_over_ranks"] - sum(sr["per_rank"])) < 1e-6
    assert "n1_reference" not in sr and "per_gpu_fraction_of_n1" not in sr and 0 < sr["min_over_max"] <= 1
    assert rep["scaling_report"]["sum_over_ranks"] == pytest.approx(sr["sum_over_ranks"], rel=1e-4)
    assert "headline_n1" not in r.stdout


def _check_eight_ranks(full, rep):
    """what differs between the ranks of a node: 8 per-rank values, disjoint core slices (when the host has a core per rank), one MIOpen
    user database per rank, and a line that still fits the driver with 8 per_rank entries"""
    pg = full["scaling_report"]["process_group"]
    assert len(pg["ranks"]) == 8 and sorted(r["rank"] for r in pg["ranks"]) == list(range(8))
    dbs = [r["miopen_user_db"] for r in pg["ranks"]]
    assert len(set(dbs)) == 8, dbs
    host = len(os.sched_getaffinity(0))
    if host >= 8:
        seen = set()
        for r in pg["ranks"]:
            assert r["cores"] and not (seen & set(r["cores"])), pg["ranks"]
            seen |= set(r["cores"])
    assert rep["n_gpus"] == 8 and len(full["scaling_report"]["per_rank"]) == 8 and all(v > 0 for v in full["scaling_report"]["per_rank"])


def test_bench_spawns_eight_ranks(tmp_path):
    """BASELINE configs[3] is an 8-GPU run: `python bench.py --gpus 8` spawns 8 workers (ports, rendezvous, per-rank MIOpen databases
    and core slices, the reductions of the report) -- the CPU stub workload over gloo here; the line stays under 4 KB."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_BACKEND"] = "gloo"
    env["ELF_BENCH_FULL"] = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "stub", "--steps", "3", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rep, full = one_line(r.stdout, env)
    _check_eight_ranks(full, rep)
    assert full["config"]["units"] == 3 * sum(1000 + k for k in range(8))


def strict_loads(line):
    def bad(c):
        raise ValueError("non-finite constant %r in the bench line" % c)
    return json.loads(line, parse_constant=bad)


def one_line(stdout, env):
    """The driver's view of a bench run: exactly ONE JSON line on stdout, strict JSON, < 4 KB; the full report is a side file."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])
    rep = strict_loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "full_report"):
        assert k in rep, k
    assert "workload" in rep["config"]
    full = strict_loads(open(env["ELF_BENCH_FULL"]).read())
    return rep, full


def test_the_line_of_a_full_default_run_fits_the_driver(tmp_path):
    """The round-4 driver could not parse a 27 KB line.  compact_line() of a stored FULL default report (every sub-result, all notes)
    must stay under 4 KB, be strict JSON, and carry the contract keys plus roofline / cpu_baseline / the sub-results' numbers."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "history", "r04z_bench_n1.json")))
    assert len(json.dumps(full)) > 20000
    full["roofline"]["frac"] = float("nan")           # a non-finite number must come out as null, not as NaN
    line = bench.compact_line(bench._clean(full), "bench_full.json")
    assert len(line) < 4096
    rep = strict_loads(line)
    assert rep["roofline"]["frac"] is None and rep["roofline"]["bound"] == "hbm" and rep["roofline"]["kernel"]
    assert rep["cpu_baseline"]["kind"] == "reference" and rep["cpu_baseline"]["cores"] > 0 and rep["cpu_baseline"]["value"] > 0
    assert rep["config"]["games_per_gpu"] == 256 and rep["config"]["rollouts_per_step"] == 4096
    assert rep["sub"]["board_step"]["parity"] == {"checked": 4096, "mismatches": 0}
    assert rep["sub"]["feature_extract"]["f32"]["frac"] > 0.5
    # a report that would not fit loses sub-results, never validity
    full["config"]["workload_short"] = "x" * 5000
    for k in list(full):
        if isinstance(full[k], dict) and k not in ("config", "roofline", "cpu_baseline"):
            full[k]["metric"] = "y" * 300
    line = bench.compact_line(bench._clean(full), "bench_full.json")
    assert len(line) < 4096 and strict_loads(line)["value"] == pytest.approx(full["value"], rel=1e-5)


def test_the_line_of_this_rounds_full_report_fits_too():
    """the same bound on the round-5 full report (more sub-results: search_only with two rooflines, leg timings, played data point)"""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06q_driver_command_bench_full.json")))
    line = bench.compact_line(bench._clean(full), "bench_full.json")
    assert len(line) < 4096
    rep = strict_loads(line)
    assert rep["steps"] == 20 and rep["warmup"] == 5 and rep["config"]["games_per_gpu"] == 2048 and rep["config"]["groups"] == 1
    assert rep["roofline"]["bound"] == "hbm" and 0.2 < rep["roofline"]["frac"] < 1 and rep["roofline"]["traffic"] > 0
    assert rep["cpu_baseline"]["kind"] == "reference" and rep["parity"]["mismatches"] == 0 and rep["parity"]["checked"] > 4000
    assert rep["sub"]["search_only"]["value"] > 6e7 and rep["sub"]["boundary"]["pinned_host"] > 3e4
    assert rep["sub"]["selfplay_games"]["derived"] is True and rep["sub"]["selfplay_games"]["played_moves_per_sec"] > 4


def test_stub_line(tmp_path):
    env = dict(os.environ, ELF_BENCH_FULL=str(tmp_path / "full.json"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--steps", "3"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rep, full = one_line(r.stdout, env)
    assert rep["n_gpus"] == 1 and rep["steps"] == 3 and full["metric"] == rep["metric"]


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_on_the_real_workloads(tmp_path):
    """The world > 1 branches of run_mcts / run_games on real kernels: two ranks on the one GPU of this box (ELF_BENCH_SHARE_GPU=1,
    process group over gloo), a small net and few rollouts.  Each rank plays its own games (game_idx_base = rank x games), the line
    carries the slowest rank's time, the sum of the rollouts and of the finished games, and the per-rank values."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_SHARE_GPU"] = "1"
    env["ELF_BENCH_FULL"] = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "both", "--games", "32", "--groups", "2",
                        "--rollouts", "64", "--steps", "6", "--warmup", "2", "--net-blocks", "2", "--net-dim", "32",
                        "--games-rollouts", "16", "--games-cutoff", "6", "--games-generations", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rep, full = one_line(r.stdout, env)
    assert rep["n_gpus"] == 2 and rep["steps"] == 6 and rep["scaling"] == "weak"
    cfg = rep["config"]
    assert cfg["games_per_gpu"] == 32 and cfg["rollouts_per_step"] == 32 * 16 and len(cfg["per_rank"]) == 2
    assert "1234 + 1000 r" in full["config"]["seed_rule"]
    # value = rollouts of BOTH ranks / the slowest rank's time
    assert abs(full["value"] - 2 * 32 * 16 * 6 / (full["ms_per_step"] * 6 / 1e3)) < 1e-6 * full["value"]
    assert rep["roofline"]["frac"] > 0 and rep["roofline"]["bound"] == "hbm"
    sr = full["scaling_report"]
    assert len(sr["per_rank"]) == 2 and all(v > 0 for v in sr["per_rank"]) and "n1_reference" not in sr
    assert sr["process_group"]["backend"] == "gloo" and sr["process_group"]["ranks_share_one_gpu"] is True
    assert rep["scaling_report"]["sum_over_ranks"] == pytest.approx(sum(sr["per_rank"]), rel=1e-4)
    assert rep["scaling_report"]["games_per_sec"] > 0        # the MEASURED games/s of the shortened configuration rides with an N > 1 line
    assert rep["cpu_baseline"] is None                      # timed on rank 0 at N = 1 only
    gm = full["selfplay_games"]
    assert gm["n_gpus"] == 2 and len(gm["per_rank_games_per_sec"]) == 2
    assert gm["games_finished"] >= 2 * 32                   # both ranks finished at least one generation of their games


@pytest.mark.gpu
def test_eight_ranks_share_the_gpu_on_the_headline_workload(tmp_path):
    """`bench.py --gpus 8` for real (README.rst:132-134: one client process per GPU): 8 ranks spawned by
    bench.py itself, all on this box's one GPU (ELF_BENCH_SHARE_GPU=1, gloo), each with its own games (seeds 1234 + 1000 r + i), MIOpen
    user database and core slice; the headline workload at toy size with a small conv net.  The compact line carries 8 per_rank entries
    and stays under 4 KB; cold MIOpen databases and all, the run takes well under 10 minutes."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_SHARE_GPU"] = "1"
    env["ELF_BENCH_FULL"] = str(tmp_path / "full.json")
    env["TMPDIR"] = str(tmp_path)                   # cold per-rank MIOpen user databases
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "mcts", "--games", "16", "--groups", "1",
                        "--rollouts", "64", "--steps", "6", "--warmup", "2", "--net-blocks", "2", "--net-dim", "32", "--no-sub"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert time.time() - t0 < 600
    rep, full = one_line(r.stdout, env)
    _check_eight_ranks(full, rep)
    cfg = rep["config"]
    assert cfg["games_per_gpu"] == 16 and len(cfg["per_rank"]) == 8 and rep["scaling"] == "weak"
    assert abs(full["value"] - 8 * 16 * 16 * 6 / (full["ms_per_step"] * 6 / 1e3)) < 1e-6 * full["value"]
    assert full["scaling_report"]["process_group"]["ranks_share_one_gpu"] is True
    for r_ in full["scaling_report"]["process_group"]["ranks"]:
        assert os.path.isdir(r_["miopen_user_db"])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("board", ["--boards", "256", "--steps", "2", "--warmup", "1"]),
                                            ("train", ["--train-batch", "256", "--train-prefetch", "2", "--train-records", "128", "--steps", "3", "--warmup", "1"])])
def test_two_ranks_share_the_gpu_on_the_board_and_trainer_workloads(workload, extra, tmp_path):
    """The world > 1 branches of run_board / run_train (per-rank seeds and records, max-over-ranks time, sum of the units) with two
    ranks on this box's one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ELF_BENCH_SHARE_GPU"] = "1"
    env["ELF_BENCH_FULL"] = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--no-cpu-baseline"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rep, full = one_line(r.stdout, env)
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["value"] > 0
    if workload == "board":
        assert full["parity_mismatches"] == 0 and full["config"]["boards_per_gpu"] == 256
        assert rep["parity"] == {"checked": 256, "mismatches": 0}
    else:
        assert full["config"]["samples_per_launch"] == 512
