#!/usr/bin/env python
"""bench.py -- driver contract.

  python bench.py --gpus N --steps K --warmup W [--workload mcts|board|train|feature|boundary|games|both]

--gpus N > 1 without a launcher environment: the script re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`, one process per GPU (under the driver's own torchrun launch RANK / WORLD_SIZE are
already set and are used as they are).

OUTPUT: rank 0 prints ONE compact strict-JSON line (< 4 KB, the last line of stdout): the contract keys, `roofline` and
`cpu_baseline` of the headline, a parity count, a few numbers per sub-result (compact_line()).  The FULL report -- every sub-result
with its notes, sweeps and per-phase detail -- is written to bench_full.json next to this file (and under gpurun_out/ where that
exists; ELF_BENCH_FULL names another path); the line's "full_report" says where.

Default workload = BASELINE.json configs[2], the configuration the headline metric is quoted on: MCTS self-play, 16 rollouts per
batch per game, 8192 rollouts per move, puct 1.5, virtual loss 1, Dirichlet 0.25/0.03, random-init 20-block/256-channel
policy/value net on PyTorch-ROCm (fp16, channels_last, called in 2048-row slices), 2048 games per GPU stepped together as one
group (node pools of 1.5 x rollouts ids per game = 167 GB; --groups 2 pipelines two half-size groups against the net).  One
"step" = one batch of the reference's batch interface for every game: G*16 rollouts (select -> leaf states -> leaf features -> net
-> expand -> backup).  value = rollouts/s summed over ranks.  The trees are grown in an UNTIMED prologue (cheap pseudo-random replies
instead of the conv net) to the point where the timed steps run at the depth of a search in progress and CROSS A MOVE BOUNDARY
(root statistics down, move choice, forward, treeAdvance, Dirichlet draws, next search).

At N = 1 the full report carries sub-results measured in the same run (each with its own roofline; the line has their numbers):
  search_only     the search kernels without the conv net: 9216 games in three groups on a shared node pool of 4096 ids per game
                  (250 GB; fewer where less HBM is free), 2048-rollout moves; kernel durations from the same games as one group
  board_step      configs[1]: 4096 boards 19x19 played to the end by the config-2 policy in one k_playout launch, EVERY final
                  (hash, ply, steps) compared with the reference (parity_checked_boards)
  board_step_9x9  configs[4]: 65 536 boards 9x9, same protocol, same check
  feature_extract extractAGZ rows per second, fp32 NCHW and fp16 NHWC, GB/s against the HBM roof
  train_loader    SURVEY.md 8f-1 trainer input pipeline
  boundary        the pybind11 drop-in boundary (_elf / _elfgames_go): rollouts/s with the batch tensors in pinned host memory
                  (the reference's Allocator) and device-resident, serial wait()/step() loop
  selfplay_games  games/s of the headline configuration, DERIVED (measured moves/s by game phase / measured game length), beside a
                  PLAYED data point at the headline's rollout count and the shortened configuration played end to end
  client_config   the reference's start_client.sh search settings (8 search threads x 200 rollouts)
Weak scaling: every rank owns independent games/boards, no data-path collective.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def chunked_forward(net, s, chunk_rows=2048):
    from elf_amd.net import chunked_forward as cf
    return cf(net, s, chunk_rows)

CK_INTERVAL = 16       # elf_amd/csrc/train.cuh: a record's state is checkpointed after every 16th move
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the same guide: 256 CUs x 4 SIMD-32, a wave64 VALU instruction issues over 2 cycles, 2.4 GHz max clock
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0   # G wave-instructions / s = 1228.8
# The scalar ALU: ONE per CU, shared by its four SIMDs.  Measured on this part (tools/issue_probe.hip, profiles/r04b_issue_probe.json):
# 1.00 scalar instruction per clock per CU for every SALU class tried (s_add / s_and_b64 / s_mul_i32 / s_lshl_b64 / s_bcnt1 / s_cselect),
# reached from 4 waves per CU on; a VALU stream issues beside it at full rate.  The same probe puts the VALU peak at 1.82 per clock and
# CU for plain VOP2 (v_add_u32; the 2-cycle wave64 rate would be 2.0) but at 1.0 per clock and CU -- one per 4 cycles per SIMD -- for
# 3-source VOP3 (v_and_or_b32), vector compares that write an SGPR pair and v_readlane, which the board kernels are full of; and a single
# wave issues at most one instruction per 4.1 - 4.2 cycles whatever its kind.
SALU_PEAK_GINST = 256 * 1.0 * 2.4       # G scalar instructions / s = 614.4
STEP_BYTES = {19: 8730, 9: 4450}  # reference state in + out + legal mask, per board step
FEAT_BYTES = {("f32", 19): 26728, ("f32", 9): 6008, ("f16", 19): 13732, ("f16", 9): 3092}   # row written + 16 history bit-planes read
# SURVEY.md 8d, algorithmic bytes of one MCTS rollout: per visited node 362 x 20 B edge read + 12 B vloss write + 12 B backup
# write; per expansion 8368 B state copy + 7240 B edge init + 26728 B features + 1468 B reply read + 8730 B legality
ROLLOUT_NODE_BYTES = 362 * 20 + 12 + 12
ROLLOUT_EXPAND_BYTES = 8368 + 7240 + 26728 + 1468 + 8730


def seeds_for(rank, boards, rep):
    # s_b = 0x9E3779B9*b + 1; b made unique per rank and per timed step
    b = np.arange(boards, dtype=np.uint64) + np.uint64((rank * 1000003 + rep) * boards)
    return b * np.uint64(0x9E3779B9) + np.uint64(1)


PARITY_NOTE = ("bit-exactness of this configuration rests on: mcts_threads = 1; reference fixtures with a stub net whose value head is on the "
               "1/256 grid (tests/golden/mcts_*.npz); the real 20x256 fp32 net against the real reference stack on this GPU "
               "(tests/test_gpu_mcts.py::test_config3_real_net_against_the_real_reference_stack; profiles/r03a_, r03m_ and r04w_config3_real_net_parity_*.json: "
               "208 searches, 204 bit-equal, 4 with one edge off by 1 ulp of its reward sum, no visit count / move differs); hazard H2 (backup order "
               "of a batch: heap-address order in the reference, first occurrence here) measured reference-vs-restatement with an "
               "un-quantised value head on 454 searches: reward sums differ in their last bits, no decision differs "
               "(profiles/r03_h2_divergence_stub_cpu.json); serial-loop equality of the pipelined / graph-replayed / fp16 path "
               "(test_pipelined_graph_fp16_groups_equal_the_serial_fp32_loop); mcts_threads > 1 and two-group pipelining are pinned on "
               "the CPU restatement / the engine itself (the reference races there)")


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def pmc_source_match(name):
    """True when profiles/<name> was measured on the kernel sources that are in the tree now (elf_amd._lib.kernel_source_hash)."""
    try:
        from elf_amd._lib import kernel_source_hash
        return load_profile_json(name).get("_source", {}).get("kernel_source_hash") == kernel_source_hash()
    except Exception:
        return False


def load_traffic(kernel):
    """HBM bytes per launch measured with rocprofv3 PMC passes (tools/profile_all.sh), committed under profiles/; None when the
    kernels have changed since the passes were run."""
    if not pmc_source_match("pmc_traffic.json"):
        return None
    return load_profile_json("pmc_traffic.json").get(kernel, {}).get("hbm_bytes_per_launch")


def issue_roof(kernel, units_per_launch, kernel_s):
    """Instruction-issue roof of a kernel that lives in LDS/registers (board engine): VALU wave-instructions per second against
    what 1024 SIMD-32 can issue (one wave64 VALU op per 2 cycles at 2.4 GHz).  The per-unit VALU count comes from the committed
    PMC pass (profiles/pmc_issue.json: SQ_INSTS_VALU / units); HBM bytes per launch are reported as `traffic`.  The counts belong
    to one version of the kernels: pmc_source_match says whether that is the version in the tree, and no fraction is printed
    when it is not."""
    per = load_profile_json("pmc_issue.json").get(kernel, {})
    valu = per.get("valu_per_unit")
    if not valu:
        return None
    match = pmc_source_match("pmc_issue.json")
    if not match:
        return {"bound": "issue", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s (VALU)", "frac": None,
                "traffic": None, "kernel": kernel, "avg_kernel_ms": kernel_s * 1e3, "pmc_source_match": False,
                "note": "profiles/pmc_issue.json was measured on other kernel sources than the tree holds (kernel_source_hash differs): "
                        "no issue-roof fraction is derived from stale instruction counts; re-run tools/profile_all.sh + tools/update_issue.py"}
    ach = units_per_launch * valu / kernel_s / 1e9
    salu = per.get("salu_per_unit")
    salu_ach = units_per_launch * salu / kernel_s / 1e9 if salu else None
    salu_frac = salu_ach / SALU_PEAK_GINST if salu_ach else None
    # class-weighted VALU roof: only plain VOP1/VOP2 issue at (nearly) the 2-cycle rate; VOP3 / VOPC / DPP / SDWA / lane access /
    # 64-bit shifts / multiplies were measured at one per 4 cycles per SIMD (profiles/r04b_issue_probe.json).  The kernel's static mix
    # (tools/valu_mix.py -> profiles/valu_mix.json) prices its instructions accordingly.
    mixes = load_profile_json("valu_mix.json")
    mix = None
    if mixes.get("_source", {}).get("kernel_source_hash") == load_profile_json("pmc_issue.json").get("_source", {}).get("kernel_source_hash"):
        base, size = kernel.split("<")[0], kernel.split("<")[1].rstrip(">")
        for k, v in mixes.items():
            if k.startswith(base + "<" + size) and not k.endswith("true>"):
                mix = v
    w_peak = mix["valu_peak_ginst"] if mix else None
    w_frac = ach / w_peak if w_peak else None
    binding = "salu" if (salu_frac is not None and salu_frac > (w_frac or ach / VALU_PEAK_GINST)) else "valu"
    return {"bound": "issue", "achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s (VALU)", "frac": ach / VALU_PEAK_GINST,
            "salu_issue": {"achieved": salu_ach, "peak": SALU_PEAK_GINST, "unit": "G scalar instructions/s", "frac": salu_frac,
                           "note": "one scalar ALU per CU at a measured 1.00 instruction per clock (profiles/r04b_issue_probe.json): "
                                   "SQ_INSTS_SALU per unit x units/s / (256 CUs x 2.4 GHz)"},
            "valu_class_weighted": {"peak": w_peak, "frac": w_frac, "quarter_rate_share_static": mix["quarter_rate_share"] if mix else None,
                                    "note": "VALU peak for this kernel's static instruction mix: plain VOP1/VOP2 at the measured 1.82 per clock and "
                                            "CU, everything else at the measured 1.0 (profiles/valu_mix.json, r04b_issue_probe.json)"},
            "salu_issue_frac": salu_frac, "binding_issue_roof": binding, "binding_frac": max(w_frac or ach / VALU_PEAK_GINST, salu_frac or 0.0),
            "traffic": load_traffic(kernel), "kernel": kernel, "avg_kernel_ms": kernel_s * 1e3, "valu_per_unit": valu, "pmc_source_match": True,
            "pmc_source_hash": load_profile_json("pmc_issue.json").get("_source", {}).get("kernel_source_hash"),
            "salu_per_unit": per.get("salu_per_unit"), "lds_per_unit": per.get("lds_per_unit"),
            "lds_bank_conflict_frac": per.get("lds_bank_conflict_frac"), "lds_active_cycles_per_unit": per.get("lds_active_cycles_per_unit"),
            "wave_issue_frac": per.get("wave_issue_frac"), "wave_wait_frac": per.get("wave_wait_frac"), "profile": per.get("profile"),
            "note": "the position never leaves LDS, so HBM is not the roof: frac = VALU wave-instructions/s (SQ_INSTS_VALU per unit from "
                    "profiles/pmc_issue.json x units/s) / (1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 op), i.e. the VALU-pipe busy "
                    "fraction at the maximum clock; traffic = PMC HBM bytes per launch; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / "
                    "SQ_LDS_IDX_ACTIVE (share of LDS-pipe cycles lost to bank conflicts), wave_issue_frac / wave_wait_frac = share of "
                    "wave cycles in an issue state / parked on s_waitcnt, from the same committed PMC passes"}


# ------------------------------------------------------------------------------------------------------------------- CPU baselines
def _oracle():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import pyoracle
    return pyoracle


def cpu_baseline_board(n, budget_s=12.0):
    """Reference (oracle/_ref, the real ELF board engine) or port timed on the host cores, bounded sample."""
    try:
        po = _oracle()
    except Exception as e:  # checker missing: report, never substitute
        return {"value": None, "unit": "board_steps/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    if po.Ref.available(n):
        R = po.Ref(n)
        t0 = time.time()
        tot, _ = R.playout(po.playout_seeds(cores * 8), threads=cores)  # calibration
        rate = tot / max(time.time() - t0, 1e-6)
        games = int(max(cores * 4, min(65536, rate * budget_s / 455.0)))
        games -= games % cores
        t0 = time.time()
        tot, _ = R.playout(po.playout_seeds(games), threads=cores)
        dt = time.time() - t0
        return {"value": tot / dt, "unit": "board_steps/s", "cores": cores, "kind": "reference", "per_core": tot / dt / cores,
                "sample": "%d of the same %dx%d config-2 games (%d board steps) on %d host threads, %.1f s" % (games, n, n, tot, cores, dt)}
    P = po.Port(n)
    t0, tot, games = time.time(), 0, 0
    while time.time() - t0 < budget_s:
        s = P.new()
        tot += len(P.playout_moves(s, int(po.playout_seeds(1, base=games)[0])))
        P.free(s)
        games += 1
    dt = time.time() - t0
    return {"value": tot / dt, "unit": "board_steps/s", "cores": 1, "kind": "port",
            "sample": "%d config-2 games (%d board steps), single thread, %.1f s" % (games, tot, dt)}


def reference_playout(n, seeds):
    """(hash_lo, hash_hi, ply, steps) of every board from the CPU checker: the real reference where its prebuilt library is
    present, its C restatement otherwise.  -> (uint32 [boards, 4], kind)"""
    po = _oracle()
    if po.Ref.available(n):
        _, out = po.Ref(n).playout(seeds, threads=max(1, min(len(os.sched_getaffinity(0)), 64)))
        return out, "reference"
    P = po.Port(n)
    out = np.zeros((len(seeds), 4), np.uint32)
    for i, sd in enumerate(seeds):
        s = P.new()
        mv = P.playout_moves(s, int(sd))
        h = P.hash(s)
        out[i] = (h & 0xFFFFFFFF, h >> 32, int(P.info(s)[0]), len(mv))
        P.free(s)
    return out, "port"


def cpu_baseline_mcts_stub(n, rollouts_per_batch):
    """The REAL reference self-play stack (oracle/_ref/libelfsp: Context batcher + GoGameSelfPlay + MCTSGoAI) on the host cores,
    one game thread + one search thread per core, net replaced by the stub: the host-side ceiling of the reference (its net time
    is excluded, ours is included)."""
    try:
        po = _oracle()
    except Exception as e:
        return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    if not po.RefSelfPlay.available(n):
        try:
            rollouts, moves = 1024, 3
            t0 = time.time()
            r = po.PortSelfPlay(n).run(num_games=1, mcts_threads=1, rollouts_per_thread=rollouts, rollouts_per_batch=rollouts_per_batch,
                                       batchsize=rollouts_per_batch, max_searches=moves, seed=1234)
            dt = time.time() - t0
            return {"value": len(r["search"]) * rollouts / dt, "unit": "rollouts/s", "cores": 1, "kind": "port",
                    "sample": "%d searches of %d rollouts (bs %d) by the single-threaded CPU restatement, stub net included, %.1f s"
                              % (len(r["search"]), rollouts, rollouts_per_batch, dt)}
        except Exception as e:
            return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    cores = max(1, min(len(os.sched_getaffinity(0)) // 2, 32))
    rollouts, moves = 2048, 2
    r = po.RefSelfPlay(n).run(num_games=cores, mcts_threads=1, rollouts_per_thread=rollouts, rollouts_per_batch=rollouts_per_batch,
                              batchsize=rollouts_per_batch, max_searches=cores * moves, seed=1234)
    dt = r["usec"] / 1e6
    done = len(r["search"]) * rollouts
    return {"value": done / dt, "unit": "rollouts/s", "cores": cores * 2, "kind": "reference",
            "sample": "%d searches of %d rollouts (bs %d) by %d reference game threads + %d search threads, stub net (net time "
                      "excluded), %.1f s" % (len(r["search"]), rollouts, rollouts_per_batch, cores, cores, dt)}


def cpu_baseline_mcts_with_net(n, rollouts_per_batch, net, dev, dtype, budget_s=18.0):
    """SURVEY.md 8d / BASELINE.md: the reference stack (its TreeSearchT + batcher, oracle/_ref/libelfsp) driving the SAME
    PyTorch-ROCm net through its batch interface (`refsp_net_fn` plays GCWrapper's part: pinned-host rows -> GPU -> net -> host),
    mcts_threads = 2, batchsize = 16 as in start_selfplay.sh, game threads sized to the host cores.  Bounded sample."""
    try:
        po = _oracle()
    except Exception as e:
        return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    if not po.RefSelfPlay.available(n) or net is None:
        return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": "oracle/_ref/libelfsp%d.so not built" % n}
    host = len(os.sched_getaffinity(0))
    calls = [0, 0]

    def net_fn(s):
        with torch.no_grad():
            x = torch.from_numpy(s).to(dev, non_blocking=True)
            out = net({"s": x.to(dtype).contiguous(memory_format=torch.channels_last)})
            calls[0] += 1
            calls[1] += s.shape[0]
            return out["pi"].float().cpu().numpy(), out["V"].float().cpu().numpy()

    R = po.RefSelfPlay(n)
    rollouts = 256
    # "num_games tuned to saturate host cores" -- swept, the best setting reported (one game thread + two search threads
    # per game; start_client.sh:21 runs 32)
    settings = [g for g in (64, 128, 256) if 3 * g <= 4 * host] or [max(1, min(host // 3, 32))]
    sweep, best = [], None
    for games in settings:
        budget = budget_s / len(settings)
        t0 = time.time()
        # calibrate on one search per game, then size the sample to the budget
        r = R.run(net=net_fn, num_games=games, mcts_threads=2, rollouts_per_thread=rollouts // 2, rollouts_per_batch=rollouts_per_batch,
                  batchsize=rollouts_per_batch, max_searches=games, seed=1234, timeout_usec=10)
        dt0 = max(r["usec"] / 1e6, 1e-3)
        per = dt0 / max(len(r["search"]), 1)
        searches = int(max(games, min(64 * games, (budget - (time.time() - t0)) / max(per, 1e-4))))
        calls[0] = calls[1] = 0
        r = R.run(net=net_fn, num_games=games, mcts_threads=2, rollouts_per_thread=rollouts // 2, rollouts_per_batch=rollouts_per_batch,
                  batchsize=rollouts_per_batch, max_searches=searches, seed=1234, timeout_usec=10)
        dt = r["usec"] / 1e6
        done = len(r["search"]) * rollouts       # 2 search threads x rollouts/2 each (tree_search.h:472-476)
        row = {"num_games": games, "rollouts_per_sec": done / dt, "searches": len(r["search"]), "seconds": dt, "net_calls": calls[0],
               "mean_rows_per_call": calls[1] / max(calls[0], 1)}
        sweep.append(row)
        if best is None or row["rollouts_per_sec"] > best["rollouts_per_sec"]:
            best = row
    return {"value": best["rollouts_per_sec"], "unit": "rollouts/s", "cores": min(host, best["num_games"] * 3), "kind": "reference",
            "num_games_sweep": sweep,
            "sample": "best of num_games in %s: %d reference game threads, %d searches of %d rollouts (2 search threads x %d, bs %d), the same "
                      "%s net on the GPU through the reference's batch interface (%d net calls, mean %.1f rows: at batchsize 16 the reference "
                      "is bound by the latency of its 16-row net calls, more games do not widen them), net time INCLUDED, %.1f s; "
                      "host has %d cores"
                      % ([r_["num_games"] for r_ in sweep], best["num_games"], best["searches"], rollouts, rollouts // 2, rollouts_per_batch,
                         str(dtype).replace("torch.", ""), best["net_calls"], best["mean_rows_per_call"], best["seconds"], host)}


# ------------------------------------------------------------------------------------------------------------------- distributed
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(n):
    """python bench.py --gpus N (no launcher environment): become `torch.distributed.run` with N workers, one per GPU."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


DIST_INFO = {}


def init_dist(args):
    """-> (rank, device index, world, dist).  One process per GPU (independent games per GPU, no data-path collective);
    the process group only carries the barrier and the max-time / sum-count reduction of the report.  ELF_BENCH_SHARE_GPU=1 maps
    every rank onto the visible GPUs round-robin (two ranks on the one GPU of a development box: the world > 1 code path on real
    kernels) and moves the process group to gloo (RCCL refuses two ranks on one device)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    share = os.environ.get("ELF_BENCH_SHARE_GPU", "0") == "1"
    device = local_rank
    if share and torch.cuda.is_available():
        device = local_rank % torch.cuda.device_count()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # every rank tunes and caches its convolutions in its own MIOpen user database: N concurrent finds on one sqlite file
        # serialise on its lock (and have been seen to corrupt it)
        # (a database path inherited from the environment is shared by all ranks of the launch: every rank takes its own sub-directory
        # of it -- found by the 8-rank test when a profiling script exported one path for the whole job)
        base = os.environ.get("MIOPEN_USER_DB_PATH") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "elf_amd_miopen")
        mi = os.path.join(base, "rank%d" % rank)
        os.makedirs(mi, exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = mi
        os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = mi
        # the host side of a rank (boundary threads, torch's intra-op pool) gets its own slice of the cores, as one client process
        # per GPU would be pinned on a node (README.rst:132-134)
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // max(1, local_world))
        mine = cores[(local_rank % local_world) * per:(local_rank % local_world + 1) * per] or cores
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            mine = cores
        os.environ.setdefault("ELF_AMD_HOST_THREADS", str(max(1, min(16, len(mine)))))
        torch.set_num_threads(max(1, len(mine)))
        backend = os.environ.get("ELF_BENCH_BACKEND", "gloo" if share else "nccl")   # "gloo": the CPU test of this path / shared GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
        if torch.cuda.is_available():
            torch.cuda.set_device(device)
        # what differs between the ranks of a node, gathered once (full report: scaling_report.process_group.ranks): the core slice,
        # the MIOpen user database and the device of every rank -- slices and databases must be disjoint
        ranks_info = [None] * world
        try:
            dist.all_gather_object(ranks_info, {"rank": rank, "device": device, "cores": [int(c) for c in mine], "miopen_user_db": os.environ["MIOPEN_USER_DB_PATH"]})
        except Exception as e:      # a report detail, never a reason to lose the run
            ranks_info = "unavailable: %r" % (e,)
        DIST_INFO.update(backend=backend, host_cores_per_rank=len(mine), ranks_share_one_gpu=bool(share),
                         miopen_user_db="per rank (%s)" % os.path.dirname(mi), ranks=ranks_info)
    if torch.cuda.is_available():
        torch.cuda.set_device(device)
    return rank, device, world, dist


def _reduce_dev(dist, dev):
    """gloo reduces host tensors (its device support is optional in a ROCm build); RCCL reduces on the rank's GPU"""
    return torch.device("cpu") if (dist is not None and dist.get_backend() != "nccl") else dev


def reduce_max_sum(dist, dev, dt, count):
    dev = _reduce_dev(dist, dev)
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    s_all = torch.tensor([count], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(s_all, op=dist.ReduceOp.SUM)
    return float(t_all.item()), int(s_all.item())


def gather_per_rank(dist, dev, value, world):
    if dist is None:
        return [value]
    t = torch.zeros(world, dtype=torch.float64, device=_reduce_dev(dist, dev))
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def scaling_report(world, value, per_rank, key):
    """What one `bench.py --gpus N` line says about scaling (BASELINE configs[3]: independent games per GPU), from THIS run's ranks
    only: the per-rank values, their sum, and the slowest rank's share of the fastest (balance).  The driver computes efficiency
    itself from its N = 1, 2, 4, 8 runs; nothing here refers to a stored N = 1 number."""
    pr = [float(v) for v in (per_rank or [])]
    return {"n_gpus": world, "metric": key, "per_rank": pr, "sum_over_ranks": float(sum(pr)) if pr else None,
            "min_over_max": (min(pr) / max(pr)) if pr and max(pr) > 0 else None,
            "value_over_sum": (value / sum(pr)) if pr and sum(pr) > 0 else None,
            "process_group": dict(DIST_INFO)}


def make_barrier(dist):
    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    return barrier


def run_stub(args, rank, local_rank, world, dist, steps, warmup):
    """CPU stand-in workload for the N > 1 plumbing test (tests/test_dist_gloo.py): no GPU, no library -- each rank "processes"
    1000 + rank units per step; the report must carry the slowest rank's time and the sum of the units."""
    dev = torch.device("cpu")
    barrier = make_barrier(dist)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        time.sleep(0.002 * (rank + 1))
    barrier()
    dt = time.perf_counter() - t0
    dt_max, total = reduce_max_sum(dist, dev, dt, (1000 + rank) * steps)
    per_rank = gather_per_rank(dist, dev, float((1000 + rank) * steps), world)
    if rank != 0:
        return None
    return {"metric": "stub_units_per_sec", "value": total / dt_max, "unit": "units/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
            "data": "synthetic", "config": {"workload": "CPU stub (plumbing test of the N-rank path)", "per_rank_units": per_rank, "units": total},
            "scaling_report": scaling_report(world, total / dt_max, [u / dt_max for u in per_rank], "stub_units_per_sec")}


# ------------------------------------------------------------------------------------------------------------------- board step
def run_board(args, rank, local_rank, world, dist, steps, warmup, with_cpu, n=None, boards=None):
    import elf_amd
    n = n or args.board_size
    boards = boards or args.boards
    eng = elf_amd.GoEngine(n, boards, local_rank)
    dev = eng.device
    out = torch.empty((boards, 4), dtype=torch.int32, device=dev)
    total = warmup + steps
    host_seeds = [seeds_for(rank, boards, r) for r in range(total)]
    seeds = [torch.from_numpy(s.view(np.int64)).to(dev) for s in host_seeds]  # resident in HBM
    step_counts = torch.zeros(total, dtype=torch.int64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(total)]
    keep = torch.empty((boards, 4), dtype=torch.int32, device=dev)   # results of the first timed pass, for the parity check

    def one(r):
        eng.reset()                       # GoState::reset for every board
        ev[r][0].record()
        eng.playout(seeds[r], out=out)    # the dominant kernel, on torch's current stream
        ev[r][1].record()
        step_counts[r] = out[:, 3].to(torch.int64).sum()
        if r == warmup:
            keep.copy_(out)

    barrier = make_barrier(dist)
    for r in range(warmup):
        one(r)
    barrier()
    t0 = time.perf_counter()
    for r in range(warmup, total):
        one(r)
    barrier()
    dt = time.perf_counter() - t0
    counts = step_counts.cpu().numpy()
    my_steps = int(counts[warmup:].sum())
    kern_ms = [ev[r][0].elapsed_time(ev[r][1]) for r in range(warmup, total)]
    dt_max, steps_all = reduce_max_sum(dist, dev, dt, my_steps)
    got = keep.cpu().numpy().astype(np.uint32)
    eng.close()
    if rank != 0:
        return None
    avg_kernel_s = float(np.mean(kern_ms)) / 1e3
    steps_per_launch = my_steps / steps
    kname = "k_playout<%d>" % n
    # wave-lifetime utilisation: every board is one wave resident from t = 0; the launch lasts as long as the longest game
    lens = got[:, 3].astype(np.float64)
    life = float(lens.mean() / max(lens.max(), 1.0))
    res = {
        "metric": "board_steps_per_sec (%dx%d GoState::forward + legal-move mask, random legal play to game end)" % (n, n),
        "value": steps_all / dt_max, "unit": "board_steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: %d concurrent %dx%d boards per GPU, config-2 random legal non-eye play "
                               "to game end, board-step kernel only (no net)" % (1 if n == 19 else 4, boards, n, n),
                   "boards_per_gpu": boards, "board_size": n, "board_steps_per_pass": steps_per_launch,
                   "mean_game_length": float(lens.mean()), "longest_game": int(lens.max()),
                   "wave_lifetime_utilisation": life,
                   "wave_lifetime_note": "mean game length / longest game: one wave per board, all resident from the start, the launch "
                                         "ends with the longest game -- the share of wave-slot time that holds a live game",
                   "algorithmic_GBps": steps_per_launch * STEP_BYTES[n] / avg_kernel_s / 1e9,
                   "algorithmic_note": "SURVEY.md 8d bytes (reference Board in+out + legal mask = %d B per step) x steps/s: the data "
                                       "movement the reference's formulation would need; NOT this kernel's roof" % STEP_BYTES[n],
                   "parallelism": "independent boards per GPU, no collective"},
    }
    roof = issue_roof(kname, steps_per_launch, avg_kernel_s)
    if roof is None:
        roof = {"bound": "issue", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s (VALU)", "frac": None,
                "traffic": load_traffic(kname), "kernel": kname, "avg_kernel_ms": avg_kernel_s * 1e3,
                "note": "profiles/pmc_issue.json has no VALU count for this kernel"}
    res["roofline"] = roof
    # parity: EVERY board of the first timed pass against the CPU checker (final hash + ply of every board)
    mismatch = None
    try:
        t0 = time.time()
        want, kind = reference_playout(n, host_seeds[warmup])
        ok = int(np.sum(np.all(got == want, axis=1)))
        res["parity_checked_boards"] = int(boards)
        res["parity_mismatches"] = int(boards) - ok
        res["parity_checker"] = "%s, %.1f s" % (kind, time.time() - t0)
        if ok != boards:
            mismatch = "bench: %d of %d boards differ from the %s -- results are invalid" % (boards - ok, boards, kind)
    except Exception as e:
        res["parity_checked_boards"] = 0
        res["parity_checker"] = "unavailable: %r" % (e,)
    if mismatch:
        raise SystemExit(mismatch)
    res["cpu_baseline"] = cpu_baseline_board(n) if with_cpu else None
    return res


# ------------------------------------------------------------------------------------------------------------------- features
def run_feature(args, rank, local_rank, world, dist, steps, warmup):
    """extractAGZ alone (north_star: HBM GB/s on feature extraction): rows boards in mid-game positions, random D4 codes, one
    launch writes every row.  fp32 NCHW (the reference's "s" layout) and fp16 NHWC."""
    import elf_amd
    n, rows = args.board_size, args.feature_rows
    dev = torch.device("cuda", local_rank)
    eng = elf_amd.GoEngine(n, rows, local_rank)
    # positions: 60 plies of the config-2 policy (8 history planes filled, stones on the board)
    out4 = torch.empty((rows, 4), dtype=torch.int32, device=dev)
    eng.playout(torch.from_numpy(seeds_for(rank, rows, 777).view(np.int64)).to(dev), max_steps=60, out=out4)
    d4 = torch.randint(0, 8, (rows,), device=dev, dtype=torch.int32)
    res = {}
    for fmt, key in (("f32_nchw", "f32"), ("f16_nhwc", "f16")):
        if key not in args.feature_formats.split(","):
            continue
        dst = None
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for _ in range(warmup):
            dst = eng.extract_agz(d4=d4, out=dst, fmt=fmt)
        torch.cuda.synchronize()
        for i in range(steps):
            ev[i][0].record()
            dst = eng.extract_agz(d4=d4, out=dst, fmt=fmt)
            ev[i][1].record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        per = FEAT_BYTES[(key, n)]
        gbs = rows * per / (ms / 1e3) / 1e9
        res[key] = {"rows_per_sec": rows / (ms / 1e3), "avg_kernel_ms": ms,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                 "traffic": load_traffic("k_extract_agz<%d>:%s" % (n, key)), "kernel": "k_extract_agz<%d> (%s)" % (n, fmt),
                                 "algorithmic_bytes_per_row": per}}
    eng.close()
    if rank != 0:
        return None
    return {"metric": "feature_rows_per_sec (extractAGZ 18 planes, random D4)", "rows": rows, "board_size": n, **res,
            "note": "one wave per row (4 rows in flight per workgroup, grid-stride); fp32 row = 25 992 B written + 736 B of history "
                    "bit-planes read (SURVEY.md 8d).  The kernel is a pure streaming write: a fill of the same bytes (tools/fill_probe.py, "
                    "torch fill_) runs at 6.6 TB/s on this part, a float4 copy at 6.3 TB/s (MI355X_MICROARCH.md); frac is against the "
                    "8 TB/s spec peak"}


# ------------------------------------------------------------------------------------------------------------------- MCTS
def mcts_parity_check(dev_index):
    """The headline's kernels against the REAL reference inside the bench run (checker only, outside every timed region): the searches of
    two reference fixtures (what the reference's self-play stack did with the oracle's stub net) are replayed by this library on the GPU
    -- tests/golden/mcts_19_r128_fresh.npz (opening positions) and mcts_19_sgf_p180.npz (ladder-suite game 406844.sgf preloaded to ply
    180: ~200 legal moves, pass edges, Tromp-Taylor leaves inside the tree); edge order, visit counts, reward sums (bit patterns) and
    the move must be equal.  -> {"checked": statistics compared, "mismatches": n, "what": ...}"""
    try:
        po = _oracle()
        import elf_amd
        checked = bad = 0
        what = []
        for name, n_max in (("mcts_19_r128_fresh", 4), ("mcts_19_sgf_p180", 2)):
            g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            cfg = dict(zip([str(k) for k in g["cfg_keys"]], g["cfg_vals"]))
            n_s = int(min(n_max, len(g["n_edges"])))
            sp = elf_amd.SelfPlay(board_size=19, num_games=1, device=dev_index, mcts_rollout_per_thread=int(cfg["rollouts_per_thread"]),
                                  mcts_rollout_per_batch=int(cfg["rollouts_per_batch"]), mcts_puct=float(np.float32(cfg["c_puct"])),
                                  mcts_virtual_loss=int(cfg["virtual_loss"]), mcts_persistent_tree=bool(cfg["persistent_tree"]),
                                  mcts_epsilon=float(np.float32(cfg["root_epsilon"])), mcts_alpha=float(np.float32(cfg["root_alpha"])),
                                  mcts_unexplored_q_zero=bool(cfg["unexplored_q_zero"]), komi=float(np.float32(cfg["komi"])),
                                  ply_pass_enabled=int(cfg["ply_pass_enabled"]), policy_distri_cutoff=int(cfg["policy_distri_cutoff"]),
                                  seed=int(cfg["seed"]), log_searches=n_s)
            if "preload_moves" in g.files:     # GameOptions.preload_sgf / preload_sgf_move_to (game_selfplay.cc:202-219)
                sp.preload(g["preload_moves"], int(g["preload_move_to"]))
            while sp.stats()["logged"] < n_s:
                rows = sp.begin_step()
                pi, v = po.stub_net(19, sp.s[:rows].cpu().numpy(), int(cfg["net_salt"]), int(cfg["net_tie_levels"]))
                sp.end_step(torch.from_numpy(pi).to(sp.device), torch.from_numpy(v).to(sp.device))
            rec, coord, visits, prior, reward = sp.search_log()
            sp.close()
            for i in range(n_s):
                ne = int(g["n_edges"][i])
                checked += 3 * ne + 1
                bad += int(rec[i].n_edges != ne)
                bad += int(np.sum(coord[i, :ne] != g["coord"][i, :ne].astype(np.int32)))
                bad += int(np.sum(visits[i, :ne] != g["visits"][i, :ne]))
                bad += int(np.sum(reward[i, :ne].view(np.uint32) != g["reward"][i, :ne].view(np.uint32)))
                bad += int(rec[i].move_played != int(g["move_played"][i]))
            what.append("%d of %s" % (n_s, name))
        return {"checked": checked, "mismatches": bad,
                "what": "searches of reference fixtures replayed on the GPU (%s; the second from a dense ply-180 SGF position): edge order, "
                        "visit counts, reward bits, move" % ", ".join(what)}
    except Exception as e:   # the checker is absent (no oracle library on this box): say so, never substitute
        return {"checked": 0, "mismatches": None, "what": "unavailable: %r" % (e,)}


class RandomReplies:
    """Pseudo-random policy/value replies (no conv net): a peaky policy grows deep, narrow trees like a trained net does.  Used
    for --net random and for the untimed tree-growing prologue of the headline.  A pool of replies is drawn once on the GPU and
    cycled through, so that a search-only measurement times the search kernels and not torch's random-number kernels."""

    def __init__(self, rows, na, dev, seed, pool=8, flat16=False):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        if flat16:
            # what the benchmark's random-init fp16 net answers (measured: priors 0.0021 .. 0.0034, ~240 distinct fp16 values among the
            # 362 of a row): a near-uniform policy on the fp16 grid, i.e. EQUAL priors among the candidates of nearly every row -- the
            # expansion's exact std::sort replay (go/mcts/mcts.h:292-297) runs for every row, as it does in the headline
            self.pool = [(torch.softmax(0.08 * torch.randn((rows, na), device=dev, generator=gen), dim=1).half().float(),
                          0.01 * torch.randn((rows,), device=dev, generator=gen)) for _ in range(pool)]
        else:
            self.pool = [(torch.softmax(4.0 * torch.randn((rows, na), device=dev, generator=gen), dim=1),
                          torch.tanh(0.5 * torch.randn((rows,), device=dev, generator=gen))) for _ in range(pool)]
        self.i = 0

    def __call__(self, s=None, rows=None):
        self.i = (self.i + 1) % len(self.pool)
        return self.pool[self.i]


def build_net(args, n, dev):
    from elf_amd.net import make_net
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.net_dtype]
    if args.net != "resnet":
        return None, dtype
    torch.backends.cudnn.benchmark = True
    net = make_net(n, args.net_blocks, args.net_dim, dev, dtype, channels_last=True, seed=0, fold_bn=not args.no_fold_bn)
    if args.no_fold_bn or dtype == torch.float32:
        args.net_impl = "eager"   # the fused epilogue is an fp16/bf16, BN-folded inference path
    if args.net_impl != "eager":
        from elf_amd.net import FusedInferenceNet
        net = FusedInferenceNet(net)
    return net, dtype


def net_flops_per_position(args, n):
    d = n * n
    return 2.0 * d * 9 * (18 * args.net_dim + 2 * args.net_blocks * args.net_dim * args.net_dim) \
        + 2.0 * d * args.net_dim * 3 + 2.0 * (2 * d * (d + 1) + d * 256 + 256)


def run_mcts(args, rank, local_rank, world, dist, steps, warmup, with_cpu):
    import elf_amd  # noqa: F401
    from elf_amd.pipeline import PipelinedSelfPlay
    n, G, K = args.board_size, args.games, args.rollouts_per_batch
    dev = torch.device("cuda", local_rank)
    net, dtype = build_net(args, n, dev)
    feat_fmt = "f16_nhwc" if (args.features == "f16" or (args.features == "auto" and net is not None and dtype == torch.float16)) else "f32_nchw"
    groups = max(1, args.groups)
    import elf_amd as _ea0
    npg_arg = args.nodes_per_game
    if npg_arg is None and args.rollouts * max(1, args.mcts_threads) >= 4096:
        # node ids per game: the library's default is 4 x rollouts + 1024 (trees of peaky nets keep large subtrees across moves: 2.2 x
        # rollouts live nodes were measured with the pregrow's random peaky replies, profiles/history/r05c_node_usage_random_replies.json).  With
        # the benchmark's random-init net the tree of a move is almost all new (peak 8918 live nodes over 10 moves of 64 games at 8192
        # rollouts, profiles/history/r05q_node_usage_resnet_10_moves.json; the pregrow stays inside the first move): 1.5 x rollouts leaves 38 % head room and
        # lets 2048 games per GPU (two waves per SIMD in the per-game kernels) fit in 167 GB.  A pool that runs out is an error, not a
        # silent truncation (ELFMCTS_E_POOL).
        npg_arg = (3 * args.rollouts * max(1, args.mcts_threads) // 2 + 63) // 64 * 64
    if torch.cuda.is_available():
        # never ask for more games than the free HBM holds (trees + feature rows + ~25 % head room for the net's activations)
        free_b, _tot = _ea0.mem_info(local_rank)
        npg_fit = npg_arg if npg_arg is not None else (4 * args.rollouts * max(1, args.mcts_threads) + 1024)
        per_game = _ea0.tree_bytes_per_game(n, npg_fit) + 2 * K * max(1, args.mcts_threads) * 18 * n * n * 4
        fit = int((0.75 if net is not None else 0.92) * free_b // per_game) // (64 * groups) * (64 * groups)
        if fit < G and fit > 0:
            sys.stderr.write("bench: %d games per GPU do not fit the free HBM (%.0f GB): running %d\n" % (G, free_b / 1e9, fit))
            G = fit
    Gg = G // groups
    G = Gg * groups
    # process g of the job is seeded 1234 + 1000 g.  The reference then gives every game thread of a process the
    # same seed (common/game_base.h:32-38: identical games under a deterministic net); here game i of the rank plays with seed
    # 1234 + 1000 rank + i (DESIGN.md "Seeds"), so the ranks' games are distinct for up to 1000 games per rank
    sp = PipelinedSelfPlay(groups=groups, seed=1234 + 1000 * rank, game_idx_base=0, wait_rows=bool(args.wait_rows), net_streams=args.net_streams,
                           board_size=n, num_games=Gg,
                           device=local_rank, mcts_rollout_per_thread=args.rollouts, mcts_rollout_per_batch=K, mcts_puct=1.5,
                           mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5,
                           ply_pass_enabled=0, policy_distri_cutoff=30, nodes_per_game=npg_arg, feature_format=feat_fmt,
                           mcts_threads=args.mcts_threads)
    na = n * n + 1
    rows_max = sp.groups[0].max_rows
    rnd = RandomReplies(rows_max, na, dev, 99 + rank)
    rnd16 = RandomReplies(rows_max, na, dev, 199 + rank, flat16=True) if args.net == "random16" else None
    uni_pi = torch.full((rows_max, na), 1.0 / na, dtype=torch.float32, device=dev)
    zero_v = torch.zeros(rows_max, dtype=torch.float32, device=dev)
    if net is not None:
        # initialisation, not a step: the first call of each convolution shape runs MIOpen's find (tens of seconds on a fresh box).
        with torch.no_grad():
            chunked_forward(net, sp.groups[0].s)
        torch.cuda.synchronize()
    graphs = {}
    if net is not None and args.net_graph:
        # one HIP graph per game group (PyTorch's CUDAGraph on ROCm): the ~85 kernel launches of a net call become one graph launch;
        # the graph reads the group's own "s" tensor, where the select kernel writes the leaf features
        from elf_amd.net import GraphedNet
        try:
            for g in sp.groups:
                graphs[g.s.data_ptr()] = GraphedNet(net, g.s)
        except Exception as e:   # capture is a launch mechanism, not a compute path: the same kernels run eagerly instead
            sys.stderr.write("bench: HIP graph capture of the net call failed (%s); launching eagerly\n" % repr(e)[:200])
            graphs = {}
            args.net_graph = 0
        torch.cuda.synchronize()

    def net_fn(s, rows):
        if net is not None:
            gn = graphs.get(s.data_ptr())
            if gn is not None:
                out = gn()
                return out["pi"], out["V"]
            with torch.no_grad():
                out = chunked_forward(net, s)   # fixed shape [Gg*K, 18, N, N]: rows beyond the count are stale and ignored
            return out["pi"], out["V"]
        if args.net == "random":
            return rnd()
        if args.net == "random16":
            return rnd16()
        return uni_pi, zero_v

    barrier = make_barrier(dist)
    # ---- untimed prologue: grow every tree with cheap pseudo-random replies so that the timed window is a search IN PROGRESS that
    # ends its move inside the window (the move boundary falls in the middle of the timed steps)
    spm = sp.groups[0].stats()["steps_per_move"]
    if args.pregrow < 0:
        pregrow = max(0, spm - warmup - max(1, steps // 2)) if steps + warmup < spm else 0
    else:
        pregrow = args.pregrow
    for _ in range(pregrow):
        sp.step(lambda s, rows: rnd())
    sp.synchronize()
    for i in range(warmup):
        sp.step(net_fn)
    sp.synchronize()
    barrier()
    st0 = sp.stats()
    sp.timing = True
    prof = None
    if os.environ.get("ELF_BENCH_CPROFILE"):      # diagnostic: where the HOST spends the timed loop (python + ctypes calls), to stderr
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    t_host = 0.0
    for i in range(steps):
        sp.step(net_fn)   # every group: net -> expand -> backup -> select -> leaf features
    t_host = time.perf_counter() - t0     # the host has ENQUEUED every step: if this is the wall time, the launches are the bottleneck
    sp.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
    sp.timing = False
    st1 = sp.stats()
    T = max(1, args.mcts_threads)
    my_rollouts = G * K * T * steps
    d = {k: st1[k] - st0[k] for k in ("moves", "games", "rollouts", "rows", "steps", "node_visits", "boundary_ns", "boundaries", "boundary_wait_ns")}
    my_rows = d["rows"]
    # select chain: the HIP events bracket begin_step, which right after a MOVE BOUNDARY also holds the next search's set-up (root
    # state, Dirichlet noise, the upload of the D4 draw windows): those steps are left out of the kernels' average as well
    sel_all = np.array([a.elapsed_time(b) for a, b in sp.t_select], np.float64)
    sel_med = float(np.median(sel_all)) if len(sel_all) else 0.0
    sel_plain = sel_all[sel_all <= 3.0 * sel_med] if len(sel_all) else sel_all
    sel_ms = float(sel_plain.mean()) * (len(sel_all) / max(steps, 1)) if len(sel_plain) else 0.0
    sel_ms_with_boundaries = float(sel_all.sum()) / steps if len(sel_all) else 0.0
    # expand + backup: the HIP events bracket end_step, which at a MOVE BOUNDARY also holds the boundary's host work and kernels
    # (reported on their own as move_boundary_ms): those steps are left out of the kernels' average launch duration
    exp_all = np.array([a.elapsed_time(b) for a, b in sp.t_expand], np.float64)
    exp_med = float(np.median(exp_all)) if len(exp_all) else 0.0
    exp_plain = exp_all[exp_all <= 3.0 * exp_med] if len(exp_all) else exp_all
    exp_ms = float(exp_plain.mean()) * (len(exp_all) / max(steps, 1)) if len(exp_plain) else 0.0
    exp_ms_with_boundaries = float(exp_all.sum()) / steps if len(exp_all) else 0.0
    net_ms = float(np.mean([a.elapsed_time(b) for a, b in sp.t_net])) if sp.t_net else 0.0   # per net call (one group)
    dt_max, roll_all = reduce_max_sum(dist, dev, dt, my_rollouts)
    per_rank = gather_per_rank(dist, dev, my_rollouts / dt, world)
    npg = int(sp.groups[0].opt.nodes_per_game)
    pinfo = [g_.pool_info() for g_ in sp.groups]
    sp.close()
    if rank != 0:
        return None
    import elf_amd as _ea
    tree_gb = G * _ea.tree_bytes_per_game(n, npg) / 1e9
    # the context's shared node pool after the run (round 6: one pool per game group, any game may hold any id): ids in trees now, the
    # largest tree, and what G fixed per-game pools would have had to provide (the sum over games of each game's own maximum)
    pool_stats = {"ids_total": sum(p_["small_total"] for p_ in pinfo), "ids_live_now": sum(p_["live"] for p_ in pinfo),
                  "largest_tree_now": max(p_["live_max_game"] for p_ in pinfo), "largest_tree_ever": max(p_["peak_max_game"] for p_ in pinfo),
                  "sum_of_per_game_maxima": sum(p_["peak_sum_games"] for p_ in pinfo),
                  "note": "ONE pool per game group shared by its games (the reference's nodes live on the heap, tree_search_node.h:439-467): "
                          "nodes_per_game sizes the pool for the MEAN tree, a game whose kept subtree is large holds more"}
    step_ms = dt_max / steps * 1e3
    depth = d["node_visits"] / max(d["rollouts"], 1)   # measured mean descent depth over the timed window
    bytes_per_step = (d["node_visits"] * ROLLOUT_NODE_BYTES + my_rows * ROLLOUT_EXPAND_BYTES) / steps
    search_s = (sel_ms + exp_ms) / 1e3
    achieved = bytes_per_step / search_s / 1e9 if search_s > 0 else None
    # the select kernel alone: per visited node the header + edge statistics + child ids + coords it really loads, per new node
    # the parent's board slot in and the child's out
    # (round 5 layout: header + the first 64 entries {prior, coord, orig} + <= 16 touched-edge records; compact boards of 2624 B)
    sel_bytes = (d["node_visits"] * (64 + 64 * 8 + 256) + my_rows * (2 * 2624 + 64)) / steps if n == 19 else None
    net_desc = ("random-init %d-block/%d-ch net on PyTorch-ROCm (%s, channels_last%s%s; leaf features %s)"
                % (args.net_blocks, args.net_dim, args.net_dtype, "" if args.no_fold_bn else ", eval BatchNorm folded into the convs",
                   {"eager": "", "fused": ", conv epilogue = one elfnet_bias_act_f16 pass"}[args.net_impl]
                   + (", one HIP graph per net call" if args.net_graph else ""), feat_fmt)
                if net is not None else "NO conv net (--net %s: search kernels only)" % args.net)
    res = {
        "metric": "mcts_rollouts_per_sec (self-play, %dx%d Go, %d rollouts/move, bs %d)" % (n, n, args.rollouts * T, K),
        "value": roll_all / dt_max, "unit": "rollouts/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload_short": "BASELINE configs[%s]: MCTS self-play bs=%d, %d rollouts/move, %s, %d games/GPU in %d group(s), window crosses a move boundary"
                                     % ("2" if world == 1 else "3, one process per GPU", K, args.rollouts * T,
                                        ("20x256 %s net on PyTorch-ROCm" % args.net_dtype) if (net is not None and args.net_blocks == 20 and args.net_dim == 256)
                                        else ("%dx%d %s net" % (args.net_blocks, args.net_dim, args.net_dtype) if net is not None else "no conv net (--net %s)" % args.net),
                                        G, groups),
                   "seed_rule": "game i of rank r: 1234 + 1000 r + i (SURVEY 8d config 4 seeds process g with 1234 + 1000 g; %d games per rank instead of "
                                "the client script's 32: one wave per game wants many games in flight)" % G,
                   "workload": "BASELINE configs[2]: MCTS self-play bs=%d, %d rollouts/move, puct 1.5, vloss 1, Dirichlet 0.25/0.03, "
                               "persistent tree, %s, %d games per GPU in %d lock-step group(s) pipelined against the net; trees grown for %d "
                               "untimed steps so that the timed window is a search in progress that crosses a move boundary"
                               % (K, args.rollouts * T, net_desc, G, groups, pregrow),
                   "search_dtype": "f32 edge statistics (the reference's float), u16 board labels", "net_dtype": args.net_dtype if net is not None else None,
                   "games_per_gpu": G, "board_size": n, "mcts_threads": T, "nodes_per_game": npg,
                   "node_bytes": _ea.tree_bytes_per_game(n, npg) // npg, "tree_pool_GB": tree_gb,
                   "node_pool": pool_stats,
                   "mcts_threads_note": None if T == 1 else ("mcts_threads = %d: the reference's search threads race on the shared tree (it is nondeterministic "
                                                             "there, SURVEY H8); this engine runs ONE deterministic interleaving of them -- thread t's K descents see "
                                                             "the virtual losses of threads < t -- which is pinned on the REAL reference under that forced "
                                                             "schedule (its turnstile build, oracle/Makefile libelfsp*_ts.so: fixtures mcts_*_T{2,3,4,8}*, "
                                                             "mcts_19_sgf_T2_p140)" % T),
                   "rollouts_per_step": G * K * T, "net_rows_per_step": my_rows / steps,
                   "search_ms_per_step": sel_ms + exp_ms, "select_ms": sel_ms, "expand_backup_ms": exp_ms,
                   "expand_backup_ms_incl_move_boundaries": exp_ms_with_boundaries, "end_step_events_with_a_boundary": int(len(exp_all) - len(exp_plain)),
                   "select_ms_incl_move_boundaries": sel_ms_with_boundaries, "begin_step_events_with_a_boundary": int(len(sel_all) - len(sel_plain)),
                   "step_minus_search_ms": step_ms - sel_ms - exp_ms, "groups": groups, "pregrow_steps": pregrow,
                   "host_wait_per_step": bool(args.wait_rows), "host_enqueue_ms_per_step": t_host / steps * 1e3,
                   "mean_depth": depth, "moves_in_window": d["moves"], "games_finished_in_window": d["games"],
                   "move_boundaries_in_window": d["boundaries"],
                   "move_boundary_ms": (d["boundary_ns"] / 1e6 / d["boundaries"]) if d["boundaries"] else None,
                   "move_boundary_queue_drain_ms": (d["boundary_wait_ns"] / 1e6 / d["boundaries"]) if d["boundaries"] else None,
                   "move_boundary_note": "move_boundary_ms = host + kernel time of one group's move boundary after its stream has drained "
                                         "(chooseAction / move sampling / resign check for its %d games, forward on the game boards, treeAdvance, "
                                         "game ends, Dirichlet + D4 draws and root set-up of the next search); queue_drain = how long the boundary "
                                         "first waited for the steps the host had queued ahead (pipeline depth; the GPU is busy meanwhile). Both "
                                         "are inside ms_per_step" % Gg,
                   "moves_per_sec": roll_all / dt_max / (args.rollouts * T),
                   "games_per_sec_note": "games/s of this configuration is MEASURED in the sub-result selfplay_games: moves/s at plies 0 / 60 / 120 / "
                                         "180 of recorded games over the measured length of games played to their natural end (no assumed game "
                                         "length; round 3 divided by an assumed 250 moves)",
                   "per_rank_rollouts_per_sec": per_rank,
                   "parity_note": PARITY_NOTE,
                   "pregrow_note": "the untimed tree-growing steps answer the leaves with random priors (not the net): the window's mean depth "
                                   "is that of trees shaped by noise priors plus the timed net steps; the search kernels are ~2 %% of the step either way",
                   "parallelism": "independent games per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "traffic": ((load_profile_json("pmc_traffic.json").get("k_mcts_search<%d>" % n, {}).get("hbm_bytes_per_rollout") or 0) * G * K * T or None)
                                if pmc_source_match("pmc_traffic.json") else None,
                     "pmc_source_match": pmc_source_match("pmc_traffic.json"),
                     "traffic_note": "PMC HBM bytes per rollout of the four search kernels (profiles/pmc_traffic.json, search-only run at depth 6.3) x "
                                     "rollouts per step; withheld (null) when the kernel sources have changed since the PMC passes",
                     "kernel": "k_mcts_select+k_mcts_leafstate+k_mcts_leafindex+k_mcts_features+k_mcts_expand+k_mcts_backup", "avg_kernel_ms": sel_ms + exp_ms,
                     "algorithmic_bytes_per_rollout": bytes_per_step / (G * K * T),
                     "select_kernel": {"achieved_GBps": (sel_bytes / (sel_ms / 1e3) / 1e9) if (sel_bytes and sel_ms > 0) else None,
                                       "bytes_per_visited_node": 64 + 64 * 8 + 256, "bytes_per_new_node": 2 * 2624 + 64,
                                       "note": "select + leaf-feature launches (HIP events around begin_step); bytes = what the kernel loads "
                                               "per visited node (header, 64 x {prior, coord, orig}, <= 16 touched-edge records) and moves per new node (parent's compact board in, child's out)"},
                     "note": "search kernels of this library (HIP events around begin_step / end_step on the groups' launch streams); bytes per "
                             "rollout = measured depth %.2f x (362 x 20 B edge read + 24 B writes) + per expansion 52534 B (SURVEY.md 8d). The "
                             "descent is a dependent pointer chase (one memory round trip per level, one wave per game): latency shares the roof "
                             "with bandwidth -- DESIGN.md section 3" % depth,
                     "mean_depth": depth},
        "selfplay_stats_window": d,
        "scaling_report": scaling_report(world, roll_all / dt_max, per_rank, "mcts_rollouts_per_sec"),
    }
    # ---- the roofs that BIND (round 6).  The SURVEY 8(d) byte model above prices the reference's formulation (362 x 20 B per visited
    # node, 52 KB per expansion); this layout moves a third of that, so `frac` is a speed-up over the model, not a bandwidth fraction.
    # Beside it: the bandwidth the kernels really draw (PMC bytes), the dependent-load chain of the descents, and the instruction
    # issue of the expansion -- each as the time it would take ALONE, and which of them is the longest.
    rf = res["roofline"]
    avg_ms = sel_ms + exp_ms
    bind = {}
    if rf.get("traffic") and avg_ms > 0:
        rf["frac_pmc"] = rf["traffic"] / (avg_ms / 1e3) / 1e9 / HBM_PEAK_GBS
        bind["hbm (PMC bytes at 8 TB/s)"] = rf["traffic"] / (HBM_PEAK_GBS * 1e9) * 1e3
    else:
        rf["frac_pmc"] = None
    chase = load_profile_json("chase.json") or {}
    hop = None
    for row in chase.get("rows", []):     # the row measured at (or above) this pool's footprint and this many chasing waves
        if row["footprint_GB"] >= min(tree_gb, 160.0) - 1e-6 and row["waves"] >= min(Gg, 4608) and (hop is None or row["ns_per_hop"] < hop["ns_per_hop"]):
            hop = row
    if hop is not None:
        # one wave per game runs K x T descents of `depth` levels one after the other, every level one dependent load; the waves of a
        # launch run side by side (85 VGPRs: 5 per SIMD, 5120 on the chip), so the chain of ONE game is the launch's floor
        waves_resident = 5 * 4 * 256
        rounds_of_waves = -(-Gg // waves_resident)
        rf["latency_chain_ms"] = K * T * depth * hop["ns_per_hop"] * 1e-6 * rounds_of_waves
        rf["latency_chain_note"] = ("%d descents x %.2f levels x %.0f ns per dependent load (tools/chase.hip: %d waves chasing through %.0f GB) per "
                                    "resident wave; against select_ms %.3f (which also holds the leaf states and the feature rows)"
                                    % (K * T, depth, hop["ns_per_hop"], hop["waves"], hop["footprint_GB"], sel_ms))
        bind["latency chain of the descents"] = rf["latency_chain_ms"]
    else:
        rf["latency_chain_ms"] = None
    iss = load_profile_json("pmc_issue.json") or {}
    if pmc_source_match("pmc_issue.json") and iss.get("k_mcts_expand<%d>" % n):
        e_ = iss["k_mcts_expand<%d>" % n]
        roll_step = G * K * T / max(1, groups)       # per launch of one group
        t_valu = e_["valu_per_unit"] * roll_step / (VALU_PEAK_GINST * 1e9) * 1e3
        t_salu = e_["salu_per_unit"] * roll_step / (SALU_PEAK_GINST * 1e9) * 1e3
        rf["expand_issue"] = {"valu_per_rollout": e_["valu_per_unit"], "salu_per_rollout": e_["salu_per_unit"],
                              "valu_frac": t_valu / exp_ms if exp_ms > 0 else None, "salu_frac": t_salu / exp_ms if exp_ms > 0 else None,
                              "note": "SQ_INSTS_VALU / SALU per rollout of k_mcts_expand (search-only PMC pass, no prior ties) x rollouts per launch "
                                      "/ the nominal issue peaks, over expand_backup_ms; with prior ties in every row (the random-init fp16 net) "
                                      "the kernel issues about twice as many"}
        bind["instruction issue of k_mcts_expand (VALU)"] = t_valu
    if bind:
        b = max(bind, key=bind.get)
        rf["binding"] = b
        rf["binding_ms"] = bind[b]
        rf["binding_frac"] = bind[b] / avg_ms if avg_ms > 0 else None
        rf["bounds_ms"] = bind
    else:
        rf["binding"] = None
    if net is not None:
        # the kernel that dominates the timed region is not this library's: PyTorch-ROCm's convolution (north_star leaves the
        # net on PyTorch).  Reported for transparency: algorithmic flops of the 20x256 net per position / measured call time.
        flops_pos = net_flops_per_position(args, n)
        rows_call = Gg * K * T
        ach = flops_pos * rows_call / (net_ms / 1e3) / 1e12 if net_ms > 0 else None
        res["net_roofline"] = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": (ach / 2500.0) if ach else None,
                               "traffic": None, "kernel": "PyTorch-ROCm net call (MIOpen CK implicit-GEMM 3x3 conv x41 + elfnet_bias_act_f16 epilogues)",
                               "avg_call_ms": net_ms, "rows_per_call": rows_call, "flops_per_position": flops_pos,
                               "note": "not a kernel of this library; dense fp16/bf16 MFMA peak from MI355X_MICROARCH.md"}
    if net is not None and with_cpu and args.net_dtype == "fp16" and not args.no_sub and args.net_variants:
        # the levers that stay inside "the net is a PyTorch-ROCm module", measured beside the headline: the same
        # call in bf16 and at twice the rows.  Reports only -- the headline stays fp16 (what the reference times), 2048 rows per call.
        try:
            import copy
            var = {}
            a2 = copy.copy(args)
            a2.net_dtype = "bf16"
            net_b, dt_b = build_net(a2, n, dev)
            rows_v = min(rows_call, 2048)      # the conv shape every net call of this bench has (chunked_forward)
            for name, nn_, dt_, rows in (("bf16_2048_rows", net_b, dt_b, rows_v), ("fp16_4096_rows", net, dtype, 2 * rows_v)):
                x = torch.zeros((rows, n, n, 18), dtype=dt_, device=dev).permute(0, 3, 1, 2)   # channels_last [rows,18,N,N]
                with torch.no_grad():
                    for _ in range(2):
                        nn_({"s": x})
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        nn_({"s": x})
                    e1.record()
                    torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                var[name] = {"avg_call_ms": ms, "TFLOP/s": flops_pos * rows / (ms / 1e3) / 1e12, "ms_per_2048_rows": ms * 2048 / rows}
            del net_b
            res["net_roofline"]["variants"] = var
            res["net_roofline"]["variants_note"] = "eager launches (no HIP graph), same fused epilogue; compare ms_per_2048_rows with avg_call_ms"
        except Exception as e:
            res["net_roofline"]["variants"] = "unavailable: %r" % (e,)
    if world == 1 and n == 19:
        res["parity"] = mcts_parity_check(local_rank)
        if res["parity"]["mismatches"]:
            raise SystemExit("bench: the search kernels differ from the reference fixture (%r) -- results are invalid" % (res["parity"],))
    if with_cpu:
        base = cpu_baseline_mcts_with_net(n, K, net, dev, dtype) if net is not None else cpu_baseline_mcts_stub(n, K)
        if base.get("value") is None and net is not None:
            base = cpu_baseline_mcts_stub(n, K)
        res["cpu_baseline"] = base
        if net is not None:
            res["cpu_baseline_stub_net"] = cpu_baseline_mcts_stub(n, K)
    else:
        res["cpu_baseline"] = None
    return res


def run_games(args, rank, local_rank, world, dist):
    """Whole self-play games per second, MEASURED, on a shortened configuration of the headline (same net, same kernels, same host
    loop): few rollouts per move and a move cutoff, so that several generations of games finish inside the run -- exercises what
    the headline window cannot hold: game ends (Tromp-Taylor scoring), restarts, Record assembly."""
    from elf_amd.pipeline import PipelinedSelfPlay
    n, K = args.board_size, args.rollouts_per_batch
    dev = torch.device("cuda", local_rank)
    net, dtype = build_net(args, n, dev)
    G, groups = args.games, max(1, args.groups)
    Gg = G // groups
    feat_fmt = "f16_nhwc" if (net is not None and dtype == torch.float16) else "f32_nchw"
    roll, cutoff = args.games_rollouts, args.games_cutoff
    sp = PipelinedSelfPlay(groups=groups, seed=4321, game_idx_base=rank * G, wait_rows=False, board_size=n, num_games=Gg, device=local_rank,
                           mcts_rollout_per_thread=roll, mcts_rollout_per_batch=K, mcts_puct=1.5, mcts_virtual_loss=1, mcts_persistent_tree=True,
                           mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, policy_distri_cutoff=30, move_cutoff=cutoff, resign_thres=0.05,
                           nodes_per_game=4096, feature_format=feat_fmt, keep_records=8)
    rnd = RandomReplies(sp.groups[0].max_rows, n * n + 1, dev, 5 + rank)
    graphs = {}
    if net is not None:
        with torch.no_grad():
            chunked_forward(net, sp.groups[0].s)
        from elf_amd.net import GraphedNet
        try:
            for g in sp.groups:
                graphs[g.s.data_ptr()] = GraphedNet(net, g.s)
        except Exception:
            graphs = {}

    def net_fn(s, rows):
        if net is None:
            return rnd()
        gn = graphs.get(s.data_ptr())
        if gn is not None:
            o = gn()
        else:
            with torch.no_grad():
                o = chunked_forward(net, s)
        return o["pi"], o["V"]

    barrier = make_barrier(dist)
    spm = sp.groups[0].stats()["steps_per_move"]
    want_moves = (cutoff - 1) * args.games_generations     # a game ends when ply reaches the cutoff
    barrier()
    t0 = time.perf_counter()
    for _ in range(want_moves * spm):
        sp.step(net_fn)
    sp.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    st = sp.stats()
    recs = sum(len(g.pop_records()) for g in sp.groups)
    dt_max, games_all = reduce_max_sum(dist, dev, dt, st["games"])
    per_rank = gather_per_rank(dist, dev, st["games"] / dt, world)
    sp.close()
    graphs.clear()

    def graphed(spx):
        """net_fn for a fresh PipelinedSelfPlay: one HIP graph per group's row tensor, as in the headline"""
        gr = {}
        if net is not None and args.net_graph:
            from elf_amd.net import GraphedNet
            try:
                for g in spx.groups:
                    gr[g.s.data_ptr()] = GraphedNet(net, g.s)
            except Exception:
                gr = {}
            torch.cuda.synchronize()

        def fn(s, rows):
            if net is None:
                return rnd()
            gn = gr.get(s.data_ptr())
            if gn is not None:
                o = gn()
            else:
                with torch.no_grad():
                    o = chunked_forward(net, s)
            return o["pi"], o["V"]
        return fn, gr

    # ---- moves/s of the HEADLINE configuration (BASELINE configs[2]: 8192 rollouts/move) by game phase: the games of each group
    # are put on a recorded 19x19 game (GameOptions.preload_sgf: two of the ladder-suite SGFs with >= 199 moves per phase, one per
    # group, tests/golden/ladder_suite.npz) at ply 0 / 60 / 120 / 180 and search there for a fixed window.  moves/s = measured
    # rollouts/s / rollouts per move: what a full-length game costs per move at that stage (legal moves thin out, trees deepen).
    phases, by_phase = (0, 60, 120, 180), {}
    lad = np.load(os.path.join(ROOT, "tests", "golden", "ladder_suite.npz")) if n == 19 else None
    T = max(1, args.mcts_threads)
    if lad is not None and args.phase_steps > 0:
        lens = np.diff(lad["offsets"])
        long_ = [i for i in range(len(lens)) if lens[i] >= 199]
        for pi_, ply in enumerate(phases):
            spx = PipelinedSelfPlay(groups=groups, seed=1234 + ply, game_idx_base=rank * G, wait_rows=False, net_streams=args.net_streams, board_size=n,
                                    num_games=Gg, device=local_rank, mcts_rollout_per_thread=args.rollouts, mcts_rollout_per_batch=K, mcts_puct=1.5,
                                    mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=0,
                                    policy_distri_cutoff=30, nodes_per_game=args.nodes_per_game, feature_format=feat_fmt, mcts_threads=args.mcts_threads)
            for gi, g in enumerate(spx.groups):
                k = long_[(pi_ * groups + gi) % len(long_)]
                g.preload(lad["moves"][lad["offsets"][k]:lad["offsets"][k + 1]].astype(np.uint16), ply)
            fn, gr = graphed(spx)
            for _ in range(max(2, args.phase_steps // 4)):
                spx.step(fn)
            spx.synchronize()
            barrier()
            tp = time.perf_counter()
            for _ in range(args.phase_steps):
                spx.step(fn)
            spx.synchronize()
            barrier()
            dtp = time.perf_counter() - tp
            dtp_max, roll_p = reduce_max_sum(dist, dev, dtp, Gg * groups * K * T * args.phase_steps)
            stp = spx.stats()
            by_phase[str(ply)] = {"rollouts_per_sec": roll_p / dtp_max, "moves_per_sec": roll_p / dtp_max / (args.rollouts * T),
                                  "ms_per_step": dtp_max / args.phase_steps * 1e3, "net_rows_per_step": stp["rows"] / max(stp["steps"], 1) * groups}
            spx.close()
            gr.clear()
    # ---- how long a game is when nothing cuts it short: the same loop with 16 rollouts per move and few games, no move cutoff, played
    # to the natural end (two passes, the move limit 2 N^2, or a resignation at resign_thres 0.05) -- MEASURED with the random-init
    # net this benchmark has (a trained net ends its games much earlier; the number is what this run's games really are)
    length = None
    if args.length_games > 0:
        Gl = max(1, args.length_games // groups)
        spl = PipelinedSelfPlay(groups=groups, seed=97, game_idx_base=rank * Gl * groups, wait_rows=False, board_size=n, num_games=Gl, device=local_rank,
                                mcts_rollout_per_thread=K, mcts_rollout_per_batch=K, mcts_puct=1.5, mcts_virtual_loss=1, mcts_persistent_tree=True,
                                mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, policy_distri_cutoff=30, resign_thres=0.05, nodes_per_game=2048,
                                feature_format=feat_fmt, keep_records=4 * Gl)
        fn, gr = graphed(spl)
        limit = 2 * n * n + 8
        tl = time.perf_counter()
        for _ in range(limit):
            spl.step(fn)
            if _ % 16 == 15 and spl.stats()["games"] >= Gl * groups:
                break
        spl.synchronize()
        stl = spl.stats()
        # the length of every FINISHED game from its record (Record.result.num_move), not moves / games (games still running would count)
        lens = [json.loads(r)["result"]["num_move"] for g in spl.groups for r in g.pop_records()]
        length = {"games_finished": len(lens), "moves_played": stl["moves"], "seconds": time.perf_counter() - tl,
                  "mean_game_length": float(np.mean(lens)) if lens else None, "min_max_game_length": [int(min(lens)), int(max(lens))] if lens else None,
                  "config": "%d games, %d rollouts/move (one step per move), no move cutoff, resign_thres 0.05, random-init net: games end by two "
                            "passes, the move limit %d, or resignation" % (Gl * groups, K, 2 * n * n)}
        spl.close()
        gr.clear()
    # ---- one PLAYED data point at the headline's rollout count: a small cohort plays whole moves from the empty board, end to end (every
    # search of 8192 rollouts, the move boundaries, Dirichlet draws, tree advance); moves/s = moves counted by the engine / wall time.
    # It confirms the per-phase moves/s above (derived from rollouts/s of a window) by an end-to-end count.
    played = None
    if args.played_moves > 0 and args.played_games > 0:
        Gp = max(1, args.played_games // groups)
        spp = PipelinedSelfPlay(groups=groups, seed=555 + 1000 * rank, game_idx_base=0, wait_rows=False, net_streams=args.net_streams, board_size=n,
                                num_games=Gp, device=local_rank, mcts_rollout_per_thread=args.rollouts, mcts_rollout_per_batch=K, mcts_puct=1.5,
                                mcts_virtual_loss=1, mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=0,
                                policy_distri_cutoff=30, nodes_per_game=args.nodes_per_game, feature_format=feat_fmt, mcts_threads=args.mcts_threads)
        fn, gr = graphed(spp)
        spm_p = spp.groups[0].stats()["steps_per_move"]
        for _ in range(2):
            spp.step(fn)
        spp.synchronize()
        barrier()
        m0 = spp.stats()["moves"]
        tpl = time.perf_counter()
        # the two warm-up steps belong to the first move: run to the end of move `played_moves`
        for _ in range(args.played_moves * spm_p - 2 + 1):
            spp.step(fn)
        spp.synchronize()
        barrier()
        dpl = time.perf_counter() - tpl
        stp = spp.stats()
        dpl_max, moves_p = reduce_max_sum(dist, dev, dpl, stp["moves"] - m0)
        played = {"games": Gp * groups * world, "moves_played": moves_p, "seconds": dpl_max, "moves_per_sec": moves_p / dpl_max,
                  "rollouts_per_move": args.rollouts * T, "rollouts_per_sec": moves_p * args.rollouts * T / dpl_max,
                  "what": "%d games (%d groups) play %d whole moves each from the empty board at %d rollouts/move with the same net: moves counted by "
                          "the engine / wall time, move boundaries included" % (Gp * groups, groups, args.played_moves, args.rollouts * T)}
        p0 = (by_phase.get("0") or {}).get("moves_per_sec")
        if p0:
            played["ratio_to_phase0_derived"] = played["moves_per_sec"] / p0
        spp.close()
        gr.clear()
    games_from_phases = None
    if by_phase and length and length["mean_game_length"]:
        # seconds per game = sum over the plies of a game of 1 / moves_per_sec(phase of that ply): plies [0,60) at the rate measured at
        # ply 0, [60,120) at ply 60, [120,180) at ply 120, the rest at ply 180; x games in flight = the whole job's games/s
        Lm, sec, edges = length["mean_game_length"], 0.0, list(phases) + [1e9]
        for i, ply in enumerate(phases):
            span = max(0.0, min(Lm, edges[i + 1]) - ply)
            sec += span / by_phase[str(ply)]["moves_per_sec"]
        games_from_phases = 1.0 / sec if sec > 0 else None    # moves/s is the job's aggregate over its games in flight: games/s = 1 / sum over a game's plies of 1 / (moves/s)
    if rank != 0:
        return None
    derived = games_from_phases is not None
    return {"metric": "selfplay_games_per_sec (derived: measured moves/s by game phase / measured game length)" if derived
                      else "selfplay_games_per_sec (played: shortened configuration)",
            "value": games_from_phases if derived else games_all / dt_max, "unit": "games/s",
            "n_gpus": world, "derived": bool(derived), "played_moves_per_sec": (played or {}).get("moves_per_sec"),
            "derived_moves_per_sec_phase0": (by_phase.get("0") or {}).get("moves_per_sec"), "played_point": played,
            "value_note": ("headline configuration (%d rollouts/move): measured moves/s by game phase (moves_per_sec_by_phase) over the MEASURED mean "
                           "game length of this run's games (game_length); no assumed length enters" % (args.rollouts * T)) if games_from_phases is not None
                          else "measured on the shortened configuration (shortened_run); the phase / length legs were switched off",
            "moves_per_sec_by_phase": by_phase, "game_length": length,
            "shortened_run": {"games_per_sec": games_all / dt_max, "games_finished": games_all, "seconds": dt_max, "moves": st["moves"], "records_kept": recs,
                              "rollouts_per_move": roll, "move_cutoff": cutoff, "games_per_gpu": G,
                              "workload": "the headline's self-play loop with %d rollouts/move (bs %d) and move_cutoff %d: %d games per GPU, %d generations "
                                          "of games played to the cutoff, scored, recorded and restarted" % (roll, K, cutoff, G, args.games_generations)},
            "games_finished": games_all, "seconds": dt_max, "moves": st["moves"], "records_kept": recs,
            "per_rank_games_per_sec": per_rank, "scaling_report": scaling_report(world, games_all / dt_max, per_rank, "selfplay_games_per_sec"),
            "config": {"workload": "games/s of BASELINE configs[2]: measured moves/s at plies 0/60/120/180 of recorded games (%d steps each) and the "
                                   "measured length of games played to their natural end; beside it the shortened configuration played end to end "
                                   "(%d rollouts/move, move_cutoff %d, %d generations)" % (args.phase_steps, roll, cutoff, args.games_generations),
                       "rollouts_per_move": args.rollouts * T, "games_per_gpu": G, "net": "resnet" if net is not None else args.net}}


def run_single_game(args, rank, local_rank, world, dist):
    """Latency of ONE game (what README.rst:147 and the GTP front-end are about: `--mcts_threads 2 --mcts_rollout_per_thread 8192
    --batchsize 16`): 1 game, bs 16, 8192 rollouts per search thread and move, the 20 x 256 fp16 net.  Three runs of the headline's own
    loop: T = 1 with the net call replayed as one HIP graph, T = 1 with eager launches (~85 kernels per call), T = 2 (graph; 16 384
    rollouts per move).  A step is one batch: 16 T descents, the net on 16 T rows, expansion + backup; a move is 512 steps.  moves/s =
    1 / (512 x ms_per_step); the split says where a step goes: search kernels, net call, and what is left (launches, host, stream waits)."""
    import copy
    out = {"metric": "single-game search latency (1 game, bs 16, 8192 rollouts per thread and move, 20x256 fp16 net)", "unit": "moves/s", "runs": {}}
    for name, T, graph in (("T1_graph", 1, 1), ("T1_eager", 1, 0), ("T2_graph", 2, 1)):
        a = copy.copy(args)
        a.games, a.groups, a.rollouts, a.mcts_threads, a.net_graph, a.pregrow, a.nodes_per_game, a.net = 1, 1, 8192, T, graph, 0, None, "resnet"
        r = run_mcts(a, rank, local_rank, world, dist, 96, 16, False)
        if rank != 0 or not isinstance(r, dict):
            continue
        c = r["config"]
        spm = 8192 // c["rollouts_per_step"] * T if c.get("rollouts_per_step") else 512
        spm = 8192 * T // max(1, c["rollouts_per_step"])
        net_ms = (r.get("net_roofline") or {}).get("avg_call_ms")
        step = r["ms_per_step"]
        search = (c.get("select_ms") or 0.0) + (c.get("expand_backup_ms") or 0.0)
        out["runs"][name] = {"mcts_threads": T, "net_graph": bool(graph), "rollouts_per_move": 8192 * T, "steps_per_move": spm, "ms_per_step": step,
                             "moves_per_sec": 1e3 / (step * spm), "seconds_per_move": step * spm / 1e3, "rollouts_per_sec": r["value"],
                             "search_kernels_ms": search, "net_call_ms": net_ms, "launch_host_wait_ms": step - search - (net_ms or 0.0),
                             "mean_depth": c.get("mean_depth")}
    if rank != 0:
        return None
    g = out["runs"].get("T1_graph") or {}
    out.update(value=g.get("moves_per_sec"), ms_per_step=g.get("ms_per_step"), moves_per_sec=g.get("moves_per_sec"),
               moves_per_sec_T2=(out["runs"].get("T2_graph") or {}).get("moves_per_sec"),
               moves_per_sec_eager=(out["runs"].get("T1_eager") or {}).get("moves_per_sec"),
               search_kernels_ms=g.get("search_kernels_ms"), net_call_ms=g.get("net_call_ms"), launch_host_wait_ms=g.get("launch_host_wait_ms"),
               note="one game cannot fill the chip (one wave descends, 16 waves expand): a step is launch / latency bound -- ~12 kernel launches "
                    "of the search + the net call at 16 rows; the headline's 2048 games amortise exactly this")
    return out


def run_client(args, rank, local_rank, world, dist):
    """The reference's canonical self-play CLIENT configuration (scripts/elfgames/go/start_client.sh:11-30): puct 0.85, virtual loss 5,
    Dirichlet 0.25 / 0.03, 8 search threads x 200 rollouts with 1 rollout per batch (the option's default), persistent tree,
    ply_pass_enabled 160, policy_distri_cutoff 30, fp16 net -- here with 256 games per GPU (the script runs 32 per client process) in
    two pipelined groups.  A move is 200 steps of 8 rollouts per game; the window crosses a move boundary.  The 8 search threads are
    the deterministic interleaving of DESIGN.md (the reference races there)."""
    from elf_amd.pipeline import PipelinedSelfPlay
    n = args.board_size
    dev = torch.device("cuda", local_rank)
    net, dtype = build_net(args, n, dev)
    G, groups, T, K, roll = args.games, max(1, args.groups), 8, 1, 200
    Gg = G // groups
    feat_fmt = "f16_nhwc" if (net is not None and dtype == torch.float16) else "f32_nchw"
    sp = PipelinedSelfPlay(groups=groups, seed=2468, game_idx_base=rank * G, wait_rows=False, board_size=n, num_games=Gg, device=local_rank,
                           mcts_rollout_per_thread=roll, mcts_rollout_per_batch=K, mcts_threads=T, mcts_puct=0.85, mcts_virtual_loss=5,
                           mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=160,
                           policy_distri_cutoff=30, policy_distri_training_for_all=True, keep_records=4, nodes_per_game=8192,
                           feature_format=feat_fmt)
    rnd = RandomReplies(sp.groups[0].max_rows, n * n + 1, dev, 17 + rank)
    graphs = {}
    if net is not None:
        with torch.no_grad():
            chunked_forward(net, sp.groups[0].s)
        from elf_amd.net import GraphedNet
        try:
            for g in sp.groups:
                graphs[g.s.data_ptr()] = GraphedNet(net, g.s)
        except Exception:
            graphs = {}

    def net_fn(s, rows):
        if net is None:
            return rnd()
        gn = graphs.get(s.data_ptr())
        if gn is not None:
            o = gn()
        else:
            with torch.no_grad():
                o = chunked_forward(net, s)
        return o["pi"], o["V"]

    barrier = make_barrier(dist)
    spm = sp.groups[0].stats()["steps_per_move"]
    steps, warm = 40, 4
    for _ in range(spm - warm - steps // 2):          # untimed: grow the trees with cheap replies up to shortly before the move ends
        sp.step(lambda s, rows: rnd())
    for _ in range(warm):
        sp.step(net_fn)
    sp.synchronize()
    s0 = sp.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sp.step(net_fn)
    sp.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    s1 = sp.stats()
    d = {k: s1[k] - s0[k] for k in ("moves", "rollouts", "rows", "boundaries", "boundary_ns", "node_visits")}
    dt_max, roll_all = reduce_max_sum(dist, dev, dt, d["rollouts"])
    sp.close()
    if rank != 0:
        return None
    return {"metric": "mcts_rollouts_per_sec (start_client.sh configuration)", "value": roll_all / dt_max, "unit": "rollouts/s", "n_gpus": world,
            "ms_per_step": dt_max / steps * 1e3, "steps": steps, "moves_in_window": d["moves"], "moves_per_sec": roll_all / dt_max / (roll * T),
            "net_rows_per_step": d["rows"] / steps, "mean_depth": d["node_visits"] / max(d["rollouts"], 1),
            "move_boundary_ms": (d["boundary_ns"] / 1e6 / d["boundaries"]) if d["boundaries"] else None,
            "config": {"workload": "scripts/elfgames/go/start_client.sh: puct 0.85, vloss 5, Dirichlet 0.25/0.03, mcts_threads 8 x 200 rollouts, "
                                   "1 rollout per batch, persistent tree, ply_pass_enabled 160, policy_distri_cutoff 30, %d games per GPU in %d "
                                   "pipelined groups, %s" % (G, groups, "20x256 fp16 net" if net is not None else "no conv net"),
                       "rollouts_per_move": roll * T, "steps_per_move": spm, "games_per_gpu": G}}


def run_boundary(args, rank, local_rank, world, dist, steps, warmup):
    """The pybind11 drop-in boundary (_elf / _elfgames_go) in a serial wait()/step() loop, as src_py/elf/utils_elf.py drives it:
    once with the batch tensors in pinned host memory (what the reference's Allocator makes: s rows D2H, replies H2D, every step),
    once device-resident.  One SharedMem buffer holds a whole device step (batchsize = games x 16)."""
    from elf_amd import compat
    compat.install_reference_module_names()
    import _elfgames_go as go
    n, K = args.board_size, args.rollouts_per_batch
    dev = torch.device("cuda", local_rank)
    net, dtype = build_net(args, n, dev)
    G = args.boundary_games
    B = G * K
    na = n * n + 1
    rnd = RandomReplies(B, na, dev, 17)
    out = {}
    for mode in ("pinned_host", "device_resident"):
        co, opt = go.ContextOptions(), go.GameOptions()
        co.num_games, co.batchsize, co.job_id = G, B, "bench"
        ts = co.mcts_options
        ts.num_threads, ts.num_rollouts_per_thread, ts.num_rollouts_per_batch = 1, args.rollouts, K
        ts.virtual_loss, ts.persistent_tree, ts.root_epsilon, ts.root_alpha = 1, True, 0.25, 0.03
        ts.alg_opt.c_puct = 1.5
        opt.mode, opt.seed, opt.board_size, opt.gpu, opt.policy_distri_cutoff = "selfplay", 1234, n, local_rank, 30
        opt.nodes_per_game = 4 * K * (steps + warmup + 8) + 1024
        GC = go.GameContext(co, opt)
        ctx = GC.ctx()
        o = ctx.createSharedMemOptions("actor_black", B)
        sm = ctx.allocateSharedMem(o, ["s", "pi", "V", "rv"])
        tens = {}
        for key in ("s", "pi", "V", "rv"):
            f = sm[key].field()
            dt_ = {"float": torch.float32, "int64_t": torch.int64}[f.type_name()]
            if mode == "pinned_host":
                t = torch.zeros(tuple(f.sz().vec()), dtype=dt_).pin_memory()
            else:
                t = torch.zeros(tuple(f.sz().vec()), dtype=dt_, device=dev)
            sm[key].set(t.data_ptr(), [s_ * t.element_size() for s_ in t.stride()])
            tens[key] = t
        ctx.start()
        GC.getClient().setRequest(0, -1, 0.0, -1)

        def one():
            smem = ctx.wait()
            k = smem.effective_batchsize()
            s = tens["s"][:k]
            if mode == "pinned_host":
                s = s.to(dev, non_blocking=True)               # Batch.cpu2gpu (utils_elf.py:243-250)
            if net is not None:
                with torch.no_grad():
                    r = net({"s": s.to(dtype).contiguous(memory_format=torch.channels_last)})
                pi, v = r["pi"], r["V"]
            else:
                pi, v = rnd()
                pi, v = pi[:k], v[:k]
            tens["pi"][:k].copy_(pi)                           # Batch.copy_from (utils_elf.py:184-222): D2H when pinned
            tens["V"][:k].copy_(v)
            tens["rv"][:k].zero_()
            torch.cuda.synchronize()
            ctx.step()
            return k

        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows = 0
        for _ in range(steps):
            rows += one()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = {"rollouts_per_sec": G * K * steps / dt, "ms_per_step": dt / steps * 1e3, "rows_per_step": rows / steps}
        ctx.stop()
        del sm, ctx, GC
    if rank != 0:
        return None
    pcie = G * K * (18 * n * n * 4 + na * 4 + 4 + 8)
    out["note"] = ("serial loop, no pipelining, %d games x %d rollouts per step, %s; pinned_host moves %d bytes per step over PCIe (s rows down, "
                   "pi/V/rv up); the difference between the two modes is the cost of the reference's memory contract"
                   % (G, K, "same net as the headline" if net is not None else "--net %s" % args.net, pcie))
    out["pcie_bytes_per_step"] = pcie
    return out


# ------------------------------------------------------------------------------------------------------------------- trainer
def synth_games(n, games, plies, dev, local_rank, seed):
    """Random legal play on the product board engine (legal mask -> torch.multinomial -> forward), `games` games of up to `plies`
    plies; pass when nothing is legal.  -> int64 tensor [games, plies] of reference Coords."""
    import elf_amd
    eng = elf_amd.GoEngine(n, games, local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    S = n + 2
    out = torch.zeros((games, plies), dtype=torch.int64, device=dev)
    for t in range(plies):
        m = eng.legal_mask().float()
        m[:, n * n] = (m[:, : n * n].sum(1) == 0).float()          # pass only when no point is legal
        a = torch.multinomial(m, 1, generator=gen).reshape(-1)
        x, y = a // n, a % n
        c = torch.where(a == n * n, torch.zeros_like(a), (y + 1) * S + (x + 1))
        eng.forward(None, c.to(torch.int32))
        out[:, t] = c
    eng.close()
    return out


def cpu_baseline_train(n, records_json, nfa):
    """The reference's per-sample trainer work (GoGameTrain::act: fromRecord, switchRandomMove, generateD4Code + every "train"
    extractor; oracle/_ref, the real reference) on the host cores, same records."""
    try:
        po = _oracle()
    except Exception as e:
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    if not po.RefSelfPlay.available(n):
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "unavailable", "sample": "oracle/_ref/libelfsp%d.so not built" % n}
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    R = po.RefSelfPlay(n)
    samples = 4000 * cores
    steps, sec = R.train_bench(records_json, samples, cores, nfa)
    return {"value": samples / sec, "unit": "samples/s", "cores": cores, "kind": "reference", "board_steps_per_sec": steps / sec,
            "sample": "%d samples of the same records (%d replayed board steps) on %d host threads, %.1f s" % (samples, steps, cores, sec)}


def run_train(args, rank, local_rank, world, dist, steps, warmup, with_cpu):
    """the trainer's input pipeline.  One step = --train-prefetch "train" batches of --train-batch samples: draw
    (record, ply, D4) like GoGameTrain::act, replay each record to its ply and extract every field, in ONE k_replay_extract launch."""
    import ctypes as C
    import elf_amd
    from elf_amd.selfplay import MctsOptions, SpOptions
    n, B, R, nfa = args.board_size, args.train_batch, args.train_records, 1
    dev = torch.device("cuda", local_rank)
    plies = 320 if n == 19 else 70
    moves = synth_games(n, R, plies, dev, local_rank, 4242 + rank).cpu().numpy().astype(np.uint16)
    rng = np.random.default_rng(7 + rank)
    P = (n + 2) ** 2
    KB = max(1, args.train_prefetch)    # train batches drawn + extracted per launch (the trainer prefetches; start_server.sh:11-12 runs 2048 loader threads)
    fmt = "f16_nhwc" if args.features in ("auto", "f16") else "f32_nchw"
    queues = args.train_sampler == "queues"
    if queues:
        # the reference's replay buffer and draws: ReaderQueues filled with InsertWithParity, GoGameTrain::act's draws by B / 64 game
        # threads per train batch (elf_amd.ReplayBuffer; bit-exact against the real GoGameTrain path, tests/test_gpu_train.py)
        # (queue_min_size 10 as in the reference's defaults, less when --train-records is too small to fill every queue that far)
        rb = elf_amd.ReplayBuffer(board_size=n, num_reader=args.train_readers, queue_min_size=max(1, min(10, R // (4 * args.train_readers))),
                                  queue_max_size=1000, batchsize=B,
                                  batches_per_launch=KB, insert_seed=77 + rank, seed=1234 + 1000 * rank, device=local_rank,
                                  num_future_actions=nfa, feature_format=fmt)
        ld = rb.loader
    else:
        ld = elf_amd.ReplayLoader(board_size=n, capacity=R, batchsize=B, device=local_rank, num_future_actions=nfa, seed=1234 + 1000 * rank,
                                  feature_format=fmt, batches_per_launch=KB)
    recs_json = []
    L = elf_amd.lib()
    opt = SpOptions(n, 1, 1024, 1600, 1, 0.25, 0.03, 0, 30, -1, 0.0, 0.0, 0, 1, 1, 0, 0, 0, MctsOptions(16, 1, 1, 0, 0, 1.5, 7.5, 0, 1, 1, 1, 0, -1))
    for r in range(R):
        pol = np.zeros((plies, P), np.uint8)                       # policy_distri_training_for_all: one policy per ply
        idx = rng.integers(0, P, size=(plies, 24))
        np.put_along_axis(pol, idx, rng.integers(1, 256, size=(plies, 24)).astype(np.uint8), axis=1)
        pol[np.arange(plies), moves[r]] = 255
        val = np.tanh(rng.standard_normal(plies)).astype(np.float32)
        rec = dict(moves=moves[r], reward=float(rng.choice([-1.0, 1.0])), black_ver=0, policies=pol, values=val)
        if queues:
            rb.insert(rec)
        else:
            ld.put(r, rec)
        if with_cpu and r < 32:                                    # the CPU baseline reads the same records as Record JSON
            args_ = (C.byref(opt), moves[r].ctypes.data, plies, pol.ctypes.data, plies, val.ctypes.data, plies, C.c_float(rec["reward"]), 0, 2, 0, 0)
            k = L.elfrec_record_to_json(*args_, None, 0)
            buf = C.create_string_buffer(k + 1)
            L.elfrec_record_to_json(*args_, buf, k + 1)
            recs_json.append(buf.raw[:k].decode())
    B1, B = B, B * KB                   # below, B = samples per launch
    out = ld._alloc(B)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    mt_sum = torch.zeros((), dtype=torch.int64, device=dev)
    fwd_sum = torch.zeros((), dtype=torch.int64, device=dev)     # plies forwarded from the checkpoints (move_idx mod CK_INTERVAL)
    ck_sum = torch.zeros((), dtype=torch.int64, device=dev)      # samples that load a checkpoint (move_idx >= CK_INTERVAL)
    barrier = make_barrier(dist)
    d = ld._draw
    for i in range(warmup):
        (rb.sample(B // 64, out=out) if queues else ld.sample(B, out=out))
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        if queues:
            rb.sample(B // 64, out=out, events=ev[i])               # the events bracket the kernel, not the index copies
        else:
            check_rc = ld.L.elftrain_draw(ld._h, B, nfa, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()), C.c_void_p(d[2].data_ptr()), ld._stream())
            assert check_rc == 0
            ev[i][0].record()
            ld.extract(d[0, :B], d[1, :B], d[2, :B], out=out)          # the dominant kernel, on torch's current stream
            ev[i][1].record()
        mt_sum += out["move_idx"].sum()
        fwd_sum += (out["move_idx"] % CK_INTERVAL).sum()
        ck_sum += (out["move_idx"] >= CK_INTERVAL).sum()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    replayed = int(mt_sum.item())
    dt_max, samples_all = reduce_max_sum(dist, dev, dt, B * steps)
    (rb.close() if queues else ld.close())
    if rank != 0:
        return None
    per_sample = replayed / (B * steps) * STEP_BYTES[n] + (26728 if n == 19 else 6008) + P + 4 * (n * n + 1) + 40
    kname = "k_replay_extract<%d>" % n
    # what THIS kernel has to move per sample: the record's checkpoint slot in (3840 B at 19x19, none below ply
    # 16), the <= 15 moves it forwards, the quantised policy row in; the feature row, the normalised policy, offline_a and 36 B of scalars out
    slot_b = 3840 if n == 19 else 1024
    row_b = 18 * n * n * (2 if ld.f16 else 4)
    fwd_mean = float(fwd_sum.item()) / (B * steps)
    ck_share = float(ck_sum.item()) / (B * steps)
    own_bytes = ck_share * slot_b + 2 * fwd_mean + P + row_b + 4 * (n * n + 1) + 8 * nfa + 36
    res = {
        "metric": "train_samples_per_sec (%dx%d replay to a random ply + every field of the reference's train batch)" % (n, n),
        "value": samples_all / dt_max, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
        "data": "synthetic",
        "config": {"workload": "SURVEY.md 8f-1 trainer input pipeline: train batches of %d samples, %d batches drawn + extracted per launch (one "
                               "step = one launch: the trainer prefetches), %d records of %d plies (random legal play on the device engine), one "
                               "MCTS policy per ply, num_future_actions %d, s rows %s; every sample starts from its record's checkpoint (the state "
                               "after every 16th move, written once per record) and forwards the rest"
                               % (B1, KB, R, plies, nfa, ld.f16 and "f16_nhwc" or "f32_nchw"),
                   "batch": B1, "batches_per_launch": KB, "samples_per_launch": B, "records": R, "board_size": n,
                   "sampler": ("reference replay buffer: %d ReaderQueues (q_min 10, q_max 1000) filled with InsertWithParity, draws of "
                               "GoGameTrain::act by %d game threads taking turns (64 states per act)" % (args.train_readers, B1 // 64)) if queues
                   else "uniform over the records (elftrain_draw)",
                   "mean_replayed_plies": replayed / (B * steps),
                   "mean_forwarded_plies": fwd_mean,
                   "replay_note": "mean_replayed_plies = the forwards the reference's switchBeforeMove makes per sample (reset + forward x "
                                  "move_to); mean_forwarded_plies = what k_replay_extract forwards from the checkpoint below move_to",
                   "replayed_board_steps_per_sec": replayed / dt,
                   "algorithmic_GBps": B * per_sample / (kern_ms / 1e3) / 1e9,
                   "algorithmic_note": "SURVEY.md 8d bytes per sample (replayed plies x 8730 B + features + policy row + scores): the data "
                                       "movement of the reference's formulation; NOT this kernel's roof",
                   "host_side": "value = samples / wall time of the timed loop, which includes the host's draws (ReaderQueues + mt19937) and "
                                "index uploads; roofline.avg_kernel_ms is the kernel alone",
                   "parallelism": "independent samples per GPU, no collective"},
    }
    ach = B * own_bytes / (kern_ms / 1e3) / 1e9
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": load_traffic(kname), "kernel": kname, "avg_kernel_ms": kern_ms, "bytes_per_sample": own_bytes,
            "samples_per_sec_kernel_only": B / (kern_ms / 1e3),
            "note": "with <= 15 forwards per sample the kernel is an HBM stream: per sample %.0f B = checkpoint slot in (%d B x %.2f of the "
                    "samples) + moves + policy row in, feature row (%d B) + normalised policy + offline_a + scalars out; achieved = samples "
                    "per launch x that / the kernel's mean HIP-event time" % (own_bytes, slot_b, ck_share, row_b)}
    issue = issue_roof(kname, fwd_sum.item() / steps, kern_ms / 1e3)   # unit = one forwarded board step
    if issue is not None:
        roof["issue"] = issue
    res["roofline"] = roof
    res["cpu_baseline"] = cpu_baseline_train(n, "[" + ",".join(recs_json) + "]", nfa) if with_cpu else None
    return res


# ------------------------------------------------------------------------------------------------------------------- the report
LINE_LIMIT = 4096   # the driver keeps a bounded tail of stdout: the ONE line it parses must fit well inside it


def _clean(x):
    """strict JSON: NaN / +-inf -> null, numpy scalars -> Python numbers"""
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if x == x and abs(x) != float("inf") else None
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def _num(x, digits=5):
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def _pick(d, keys, digits=5):
    return {k: _num(d.get(k), digits) for k in keys if isinstance(d, dict) and k in d}


ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_kernel_ms")


def _roof(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ROOF_KEYS)
    if isinstance(out.get("kernel"), str):
        out["kernel"] = out["kernel"][:56]
    if isinstance(out.get("unit"), str):
        out["unit"] = out["unit"][:40]
    for k in ("binding_issue_roof", "binding_frac", "lds_bank_conflict_frac", "frac_pmc", "latency_chain_ms", "binding_ms"):
        if r.get(k) is not None:
            out[k] = _num(r[k], 4)
    if isinstance(r.get("binding"), str):      # which roof binds the search kernels: PMC bandwidth, the descents' latency chain, or issue
        out["binding"] = r["binding"][:48]
    if isinstance(r.get("expand_issue"), dict):
        out["expand_valu_frac"] = _num(r["expand_issue"].get("valu_frac"), 4)
        out["expand_salu_frac"] = _num(r["expand_issue"].get("salu_frac"), 4)
    return out


def _sub_summary(name, d):
    """one small object per sub-result: value, unit, roofline fraction, parity -- the notes stay in the full report"""
    if not isinstance(d, dict):
        return {"status": str(d)[:60]}
    if name == "feature_extract":
        out = {}
        for key in ("f32", "f16"):
            if isinstance(d.get(key), dict):
                r = d[key].get("roofline") or {}
                out[key] = {"rows_per_sec": _num(d[key].get("rows_per_sec")), "GBps": _num(r.get("achieved")), "frac": _num(r.get("frac"), 4),
                            "traffic": _num(r.get("traffic")), "avg_kernel_ms": _num(d[key].get("avg_kernel_ms"))}
        return out
    if name == "boundary":
        return {k: _num((d.get(k) or {}).get("rollouts_per_sec")) for k in ("pinned_host", "device_resident") if isinstance(d.get(k), dict)}
    out = _pick(d, ("value", "unit", "ms_per_step"))
    if isinstance(out.get("unit"), str):
        out["unit"] = out["unit"][:24]
    r = d.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = {k: v for k, v in _roof(r).items() if k in ("bound", "frac", "achieved", "peak", "binding_frac", "lds_bank_conflict_frac", "avg_kernel_ms", "frac_pmc")}
    if "parity_checked_boards" in d:
        out["parity"] = {"checked": d.get("parity_checked_boards"), "mismatches": d.get("parity_mismatches")}
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        out["cpu_baseline"] = _pick(cb, ("value", "cores", "kind"))
    if name == "search_only":
        out["rollouts_per_move"] = _num(d.get("rollouts_per_move"))     # NOT the headline's 8192: shorter moves on smaller pools
        out["nodes_per_game"] = _num(d.get("nodes_per_game"))
    for k in ("games_per_gpu", "groups", "mean_depth", "select_ms", "expand_backup_ms", "rollouts_per_step", "pinned_host", "device_resident",
              "moves_per_sec", "played_moves_per_sec", "derived_moves_per_sec_phase0", "derived", "moves_per_sec_T2", "moves_per_sec_eager",
              "search_kernels_ms", "net_call_ms", "launch_host_wait_ms"):
        if k in d and not isinstance(d[k], (dict, list, str)):
            out[k] = _num(d[k])
    return out


SUB_NAMES = ("search_only", "board_step", "board_step_9x9", "feature_extract", "train_loader", "boundary", "selfplay_games", "client_config",
             "single_game")


def compact_line(res, full_path=None):
    """The ONE line the driver parses: every key of the bench contract, `roofline` and `cpu_baseline` of the headline, a parity count,
    and a few numbers per sub-result; < LINE_LIMIT bytes, strict JSON.  Notes, sweeps and per-phase detail are in the full report."""
    cfg = res.get("config") or {}
    line = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data"), 7)
    if isinstance(line.get("metric"), str):
        line["metric"] = line["metric"][:100]
    c = _pick(cfg, ("games_per_gpu", "groups", "rollouts_per_step", "mean_depth", "board_size", "mcts_threads", "net_dtype",
                    "select_ms", "expand_backup_ms", "moves_in_window", "move_boundary_ms",
                    "boards_per_gpu", "board_steps_per_pass", "nodes_per_game", "node_bytes", "tree_pool_GB", "mean_forwarded_plies",
                    "mean_replayed_plies", "samples_per_launch", "batch"))
    c = {"workload": str(cfg.get("workload_short") or cfg.get("workload") or "")[:164], **c}
    if isinstance(cfg.get("node_pool"), dict):     # the shared node pool: ids in trees now / the largest tree ever
        c["live_nodes"] = _num(cfg["node_pool"].get("ids_live_now"))
        c["largest_tree"] = _num(cfg["node_pool"].get("largest_tree_ever"))
    if cfg.get("per_rank_rollouts_per_sec") is not None and res.get("n_gpus", 1) > 1:
        c["per_rank"] = [_num(v) for v in cfg["per_rank_rollouts_per_sec"]]
    line["config"] = c
    line["roofline"] = _roof(res.get("roofline"))
    if isinstance(res.get("roofline"), dict):
        for k in ("mean_depth", "algorithmic_bytes_per_rollout"):
            if res["roofline"].get(k) is not None:
                line["roofline"][k] = _num(res["roofline"][k])
    nr = res.get("net_roofline")
    if isinstance(nr, dict):
        line["net_roofline"] = _pick(nr, ("bound", "achieved", "peak", "unit", "frac", "avg_call_ms", "rows_per_call"))
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample") or "")[:84]
    else:
        line["cpu_baseline"] = None
    par = res.get("parity")
    if isinstance(par, dict):
        line["parity"] = _pick(par, ("checked", "mismatches", "what"))
        if isinstance(line["parity"].get("what"), str):
            line["parity"]["what"] = line["parity"]["what"][:64]
    elif "parity_checked_boards" in res:
        line["parity"] = {"checked": res.get("parity_checked_boards"), "mismatches": res.get("parity_mismatches")}
    sub = {k: _sub_summary(k, res[k]) for k in SUB_NAMES if k in res}
    if sub:
        line["sub"] = sub
    if isinstance(res.get("scaling_report"), dict) and res.get("n_gpus", 1) > 1:
        line["scaling_report"] = _pick(res["scaling_report"], ("n_gpus", "sum_over_ranks", "min_over_max"))
        gm = res.get("selfplay_games")
        if isinstance(gm, dict):     # the MEASURED games/s of the shortened configuration (all ranks), for the games/sec curve
            line["scaling_report"]["games_per_sec"] = _num((gm.get("shortened_run") or {}).get("games_per_sec"))
            line["scaling_report"]["games_per_sec_per_rank"] = [_num(v) for v in (gm.get("per_rank_games_per_sec") or [])]
    line["full_report"] = full_path
    line = _clean(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # never exceed the limit: drop the sub-results one by one (last first), then the optional blocks
    drop = [("sub", k) for k in reversed(SUB_NAMES)] + [("net_roofline", None), ("scaling_report", None)]
    while len(text) >= LINE_LIMIT - 64 and drop:
        a, b = drop.pop(0)
        if b is None:
            line.pop(a, None)
        elif isinstance(line.get(a), dict):
            line[a].pop(b, None)
        line["truncated"] = True
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return text


def emit(res, args):
    """Full report -> bench_full.json (next to bench.py, and under gpurun_out/ where that exists); ONE compact line -> stdout."""
    full = _clean(res)
    paths = []
    names = [os.environ.get("ELF_BENCH_FULL") or os.path.join(ROOT, "bench_full.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        names.append(os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    for p in names:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1, allow_nan=False)
            paths.append(os.path.relpath(p, ROOT))
        except Exception:
            pass
    line = compact_line(full, paths[0] if paths else None)
    assert len(line) < LINE_LIMIT and "\n" not in line
    json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # strict: no NaN / Infinity
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["mcts", "board", "train", "feature", "boundary", "games", "client", "single", "both", "stub"], default="both",
                    help="both (default) = mcts headline + every sub-result at N = 1")
    ap.add_argument("--train-batch", type=int, default=2048)
    ap.add_argument("--train-prefetch", type=int, default=16, help="train batches drawn + extracted per launch (1 = one batch per launch)")
    ap.add_argument("--train-records", type=int, default=256)
    ap.add_argument("--train-sampler", choices=["queues", "uniform"], default="queues",
                    help="queues = the reference's ReaderQueues + GoGameTrain::act draws (elf_amd.ReplayBuffer); uniform = elftrain_draw")
    ap.add_argument("--train-readers", type=int, default=4, help="ReaderQueues.num_reader for --train-sampler queues (reference default 50)")
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--boards9", type=int, default=65536)
    ap.add_argument("--board-size", type=int, default=19)
    ap.add_argument("--feature-rows", type=int, default=16384)
    ap.add_argument("--feature-formats", default="f32,f16", help="row formats the feature workload times (one per PMC pass: both use the same kernel)")
    ap.add_argument("--games", type=int, default=2048, help="games per GPU of the headline (split over --groups): one wave per game in the "
                    "search kernels, so 2048 games = two waves per SIMD per launch of a 1024-game group")
    ap.add_argument("--sub-games", type=int, default=256, help="games per GPU of the sub-results that search with the conv net (game phases, "
                    "client configuration): they measure moves/s, which does not depend on the games in flight")
    ap.add_argument("--groups", type=int, default=1, help="lock-step game groups pipelined against the net (1 = serial: select -> net -> expand; the "
                    "net is 99 %% of a step, so pipelining buys < 0.5 %% of throughput, and with one group the HIP-event durations of the "
                    "search kernels are kernel durations, not times shared with the convolutions of another group)")
    ap.add_argument("--net-streams", type=int, default=1, help="net streams of the pipeline (1 = the groups' net calls queue on one stream; "
                    "= --groups: every group's call on its own stream)")
    ap.add_argument("--rollouts", type=int, default=8192, help="TSOptions.num_rollouts_per_thread")
    ap.add_argument("--rollouts-per-batch", type=int, default=16)
    ap.add_argument("--mcts-threads", type=int, default=1, help="TSOptions.num_threads (search threads per game)")
    ap.add_argument("--nodes-per-game", type=int, default=None)
    ap.add_argument("--pregrow", type=int, default=-1, help="untimed tree-growing steps before the warm-up (-1: so that the move ends mid-window)")
    ap.add_argument("--net", choices=["resnet", "random", "random16", "null"], default="resnet",
                    help="random: peaky pseudo-random replies (deep trees, no prior ties); random16: near-uniform replies on the fp16 grid like the "
                         "random-init fp16 net's (prior ties in every row: the expansion's exact std::sort replay); null: uniform")
    ap.add_argument("--no-fold-bn", action="store_true")
    ap.add_argument("--net-variants", action="store_true", help="also time the net call in bf16 and at 4096 rows (reports only; a second MIOpen "
                    "find of ~25 s; round 2-4 numbers: DESIGN.md section 3)")
    ap.add_argument("--net-blocks", type=int, default=20)
    ap.add_argument("--net-dim", type=int, default=256)
    ap.add_argument("--net-dtype", choices=["fp16", "bf16", "fp32"], default="fp16")
    ap.add_argument("--net-impl", choices=["eager", "fused"], default="fused",
                    help="eager: plain PyTorch modules; fused: PyTorch convs + one HIP epilogue pass per conv (elfnet_bias_act_f16)")
    ap.add_argument("--net-graph", type=int, default=1, help="1: replay each group's net call as one HIP graph; 0: eager launches")
    ap.add_argument("--features", choices=["auto", "f32", "f16"], default="auto",
                    help="leaf feature rows: f32 NCHW (reference layout) or f16 channels_last (auto: f16 when the net is fp16)")
    ap.add_argument("--wait-rows", type=int, default=0, help="1: the host waits for the row count of every step (drop-in path)")
    ap.add_argument("--boundary-games", type=int, default=128)
    ap.add_argument("--games-rollouts", type=int, default=32)
    ap.add_argument("--games-cutoff", type=int, default=40)
    ap.add_argument("--games-generations", type=int, default=2)
    ap.add_argument("--search-only-games", type=int, default=9216, help="games per GPU of the search-only sub-result (no conv net); 0 = off")
    ap.add_argument("--phase-steps", type=int, default=16, help="timed steps per game phase (plies 0/60/120/180) of the games/s leg; 0 = off")
    ap.add_argument("--played-games", type=int, default=64, help="cohort of the PLAYED data point at the headline's rollout count; 0 = off")
    ap.add_argument("--played-moves", type=int, default=2, help="whole moves the cohort plays end to end; 0 = off")
    ap.add_argument("--length-games", type=int, default=32, help="games played to their natural end to measure the game length; 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="headline only (no sub-results)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)      # does not return

    rank, local_rank, world, dist = init_dist(args)
    legs, t_leg = {}, [time.time()]

    def leg(name):     # wall seconds per leg of the run (full report): what a default run spends where
        now = time.time()
        legs[name] = round(now - t_leg[0], 2)
        t_leg[0] = now
    with_cpu = (not args.no_cpu_baseline) and world == 1
    sub = args.workload == "both" and world == 1 and not args.no_sub
    res = None
    if args.workload == "stub":
        res = run_stub(args, rank, local_rank, world, dist, args.steps or 5, args.warmup or 0)
    if args.workload in ("mcts", "both"):
        steps = args.steps if args.steps is not None else 40
        warmup = args.warmup if args.warmup is not None else 8
        res = run_mcts(args, rank, local_rank, world, dist, steps, warmup, with_cpu)
        leg("headline")

    if torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.empty_cache()     # the headline's net graphs / activations go back to the device before the next leg sizes itself
    import copy as _copy
    sargs = _copy.copy(args)       # the sub-results that search with the conv net
    if args.workload == "both":
        sargs.games = max(args.groups, min(args.games, args.sub_games))

    def own(name, dflt):      # --steps/--warmup belong to the headline unless a single sub-workload was asked for
        v = getattr(args, name)
        return v if (v is not None and args.workload != "both") else dflt

    if sub and args.board_size == 19 and args.search_only_games > 0:
        # the search kernels without the conv net, with as many games in flight as the free HBM holds (up to --search-only-games): moves of
        # 2048 rollouts on a SHARED node pool of 2 x 2048 ids per game -- round 6: the 250 GB that held 4608 fixed 8192-id pools hold 9216
        # games now (a tree of these moves peaks near 2.5 k nodes) = 9 waves per SIMD in the per-game kernels, three pipelined groups
        try:
            import copy
            import elf_amd
            a2 = copy.copy(args)
            nodes = 4096
            free, _ = elf_amd.mem_info(local_rank)
            per_game = elf_amd.tree_bytes_per_game(19, nodes) + 2 * 16 * 18 * 361 + (1 << 16)    # + its feature rows and slack
            fit = int(0.85 * free // per_game) // 192 * 192
            so_games = max(192, min(args.search_only_games, fit))
            a2.net, a2.features, a2.games, a2.groups, a2.nodes_per_game, a2.rollouts, a2.pregrow = "random", "f16", so_games, 3, nodes, 2048, 0
            so = run_mcts(a2, rank, local_rank, world, dist, 32, 88, False)
            # the same games as ONE group: no launch of another group shares the GPU, so the HIP-event durations are the kernels' own
            # (the roofline of the kernels); the two-group run above is the throughput figure (its phases overlap)
            a1 = copy.copy(a2)
            a1.groups = 1
            so1 = run_mcts(a1, rank, local_rank, world, dist, 32, 88, False)
            if rank == 0:
                c = so["config"]
                so["roofline_pipelined"] = so["roofline"]
                so["roofline"] = dict(so1["roofline"], measured_on="the same games as one group (kernel durations without another group's "
                                      "launches beside them): %.1f M rollouts/s, select %.3f ms + expand/backup %.3f ms per %d rollouts"
                                      % (so1["value"] / 1e6, so1["config"]["select_ms"], so1["config"]["expand_backup_ms"], so1["config"]["rollouts_per_step"]))
                res["search_only"] = {"metric": "mcts_rollouts_per_sec, search kernels only (random replies instead of the conv net)", "value": so["value"],
                                      "unit": "rollouts/s", "ms_per_step": so["ms_per_step"], "games_per_gpu": c["games_per_gpu"], "groups": c["groups"],
                                      "rollouts_per_step": c["rollouts_per_step"], "mean_depth": c["mean_depth"], "select_ms": c["select_ms"],
                                      "expand_backup_ms": c["expand_backup_ms"], "roofline": so["roofline"], "roofline_pipelined": so["roofline_pipelined"], "nodes_per_game": nodes,
                                      "rollouts_per_move": a2.rollouts, "node_pool": c.get("node_pool"),
                                      "tree_pool_GB": so_games * elf_amd.tree_bytes_per_game(19, nodes) / 1e9, "games_that_fit_free_hbm": fit,
                                      "note": "same kernels, same tree shape (2048 rollouts per move, depth ~6.3) as the headline's search, on a shared "
                                              "node pool of 4096 ids per game; the per-game kernels are latency chains, more resident waves per "
                                              "SIMD hide them (4608 games on fixed 8192-id pools, the round-5 configuration: 72 M rollouts/s)"}
        except Exception as e:   # e.g. not enough free HBM beside another process
            if rank == 0:
                res["search_only"] = "unavailable: %r" % (e,)
            import gc
            gc.collect()                     # a half-built context frees its node pools in its finaliser
            torch.cuda.empty_cache()
    leg("search_only")
    if args.workload == "board" or sub:
        b = run_board(args, rank, local_rank, world, dist, own("steps", 20), own("warmup", 3), with_cpu)
        if args.workload == "board":
            res = b
        elif rank == 0:
            res["board_step"] = b
    leg("board_step")
    if sub and args.board_size == 19:
        b9 = run_board(args, rank, local_rank, world, dist, 8, 2, with_cpu, n=9, boards=args.boards9)
        if rank == 0:
            res["board_step_9x9"] = b9
    leg("board_step_9x9")
    if args.workload == "feature" or sub:
        f = run_feature(args, rank, local_rank, world, dist, own("steps", 20), own("warmup", 3))
        if args.workload == "feature":
            res = f
        elif rank == 0:
            res["feature_extract"] = f
    leg("feature_extract")
    if args.workload == "train" or sub:
        t = run_train(args, rank, local_rank, world, dist, own("steps", 20), own("warmup", 3), with_cpu)
        if args.workload == "train":
            res = t
        elif rank == 0:
            res["train_loader"] = t
    leg("train_loader")
    if args.workload == "boundary" or sub:
        bd = run_boundary(args, rank, local_rank, world, dist, own("steps", 10), own("warmup", 3))
        if args.workload == "boundary":
            res = bd
        elif rank == 0:
            res["boundary"] = bd
    leg("boundary")
    if args.workload == "games" or sub or (args.workload == "both" and world > 1 and not args.no_sub):
        # N > 1: the measured games/s of the shortened configuration rides with the headline, so that one `bench.py --gpus N` line
        # per N yields the games/sec scaling curve BASELINE.json names (measured, not the 250-moves-per-game estimate)
        gm = run_games(sargs, rank, local_rank, world, dist)
        if args.workload == "games":
            res = gm
        elif rank == 0:
            res["selfplay_games"] = gm
            # north_star: games/sec "as absolute and as fraction of HBM roofline": the SURVEY 8(d) bytes of one game = moves x rollouts per
            # move x bytes per rollout (at the headline's measured depth) x games/s, against the HBM peak (x N GPUs)
            try:
                bpr = res["roofline"]["algorithmic_bytes_per_rollout"]
                L_ = (gm.get("game_length") or {}).get("mean_game_length")
                if bpr and L_ and gm.get("value"):
                    gbs = gm["value"] * L_ * args.rollouts * max(1, args.mcts_threads) * bpr / 1e9
                    gm["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": gbs / (HBM_PEAK_GBS * world),
                                      "traffic": None, "bytes_per_game": L_ * args.rollouts * max(1, args.mcts_threads) * bpr,
                                      "note": "games/s x (measured game length x rollouts per move x SURVEY 8(d) bytes per rollout at the measured depth); the "
                                              "step is 99 % convolution (net_roofline), so this fraction is small by construction"}
            except Exception:
                pass
    leg("selfplay_games")
    if args.workload == "client" or sub:
        cl = run_client(sargs, rank, local_rank, world, dist)
        if args.workload == "client":
            res = cl
        elif rank == 0:
            res["client_config"] = cl
    leg("client_config")
    if args.workload == "single" or (sub and args.board_size == 19 and args.net == "resnet"):
        try:
            sg = run_single_game(sargs, rank, local_rank, world, dist)
        except Exception as e:
            sg = "unavailable: %r" % (e,)
        if args.workload == "single":
            res = sg
        elif rank == 0:
            res["single_game"] = sg
    leg("single_game")
    if rank == 0:
        if isinstance(res, dict):
            res["leg_seconds"] = legs
        emit(res, args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
