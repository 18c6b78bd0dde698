#!/usr/bin/env python
"""bench.py -- driver contract.

  python bench.py --gpus N --steps K --warmup W [--workload mcts|board]
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Default workload "mcts" = BASELINE.json configs[2], the configuration the headline metric is quoted on:
MCTS self-play, 16 rollouts per batch per game, 8192 rollouts per move, puct 1.5, virtual loss 1, Dirichlet
0.25/0.03, random-init 20-block/256-channel policy/value net on PyTorch-ROCm (fp16, channels_last), G games per
GPU in lock-step.  One "step" = one batch of the reference's batch interface for every game: G*16 rollouts
(select -> leaf features -> net -> expand -> backup).  value = rollouts/s summed over ranks.

Workload "board" = configs[1]: 4096 concurrent 19x19 boards per GPU played to game end by the config-2 policy,
whole games inside one k_playout launch; value = board steps/s.  The default run also measures it and reports
it under "board_step" in the same JSON line.

Weak scaling: every rank owns independent games/boards, no data-path collective.
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
# SURVEY.md 8d, algorithmic bytes of one MCTS rollout: per visited node 362 x 20 B edge read + 12 B vloss write + 12 B
# backup write; per expansion 8368 B state copy + 7240 B edge init + 26728 B features + 1468 B reply read + 8730 B legality
ROLLOUT_NODE_BYTES = 362 * 20 + 12 + 12
ROLLOUT_EXPAND_BYTES = 8368 + 7240 + 26728 + 1468 + 8730


def seeds_for(rank, boards, rep):
    # s_b = 0x9E3779B9*b + 1; b made unique per rank and per timed step
    b = np.arange(boards, dtype=np.uint64) + np.uint64((rank * 1000003 + rep) * boards)
    return b * np.uint64(0x9E3779B9) + np.uint64(1)


def cpu_baseline_board(n, budget_s=12.0):
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
        tot, _ = R.playout(playout_seeds(cores * 8), threads=cores)  # calibration
        rate = tot / max(time.time() - t0, 1e-6)
        games = int(max(cores * 4, min(65536, rate * budget_s / 455.0)))
        games -= games % cores
        t0 = time.time()
        tot, _ = R.playout(playout_seeds(games), threads=cores)
        dt = time.time() - t0
        return {"value": tot / dt, "unit": "board_steps/s", "cores": cores, "kind": "reference",
                "per_core": tot / dt / cores,
                "sample": "%d of the same %dx%d config-2 games (%d board steps) on %d host threads, %.1f s" % (games, n, n, tot, cores, dt)}
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


def cpu_baseline_mcts(n, rollouts_per_batch):
    """The REAL reference self-play stack (oracle/_ref/libelfsp: Context batcher + GoGameSelfPlay + MCTSGoAI) on the host
    cores, one game thread + one search thread per core, net replaced by the stub (host-side ceiling of the reference:
    its net time is excluded, ours is included)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        from pyoracle import RefSelfPlay
    except Exception as e:
        return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    if not RefSelfPlay.available(n):
        # no prebuilt reference here (fresh clone): time the CPU restatement (oracle/mcts_oracle.cc) instead, one core
        try:
            from pyoracle import PortSelfPlay
            rollouts, moves = 1024, 3
            t0 = time.time()
            r = PortSelfPlay(n).run(num_games=1, mcts_threads=1, rollouts_per_thread=rollouts, rollouts_per_batch=rollouts_per_batch,
                                    batchsize=rollouts_per_batch, max_searches=moves, seed=1234)
            dt = time.time() - t0
            return {"value": len(r["search"]) * rollouts / dt, "unit": "rollouts/s", "cores": 1, "kind": "port",
                    "sample": "%d searches of %d rollouts (bs %d) by the single-threaded CPU restatement, stub net included, %.1f s"
                              % (len(r["search"]), rollouts, rollouts_per_batch, dt)}
        except Exception as e:
            return {"value": None, "unit": "rollouts/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    cores = max(1, min(len(os.sched_getaffinity(0)) // 2, 32))
    rollouts, moves = 2048, 2
    r = RefSelfPlay(n).run(num_games=cores, mcts_threads=1, rollouts_per_thread=rollouts, rollouts_per_batch=rollouts_per_batch,
                           batchsize=rollouts_per_batch, max_searches=cores * moves, seed=1234)
    dt = r["usec"] / 1e6
    done = len(r["search"]) * rollouts
    return {"value": done / dt, "unit": "rollouts/s", "cores": cores * 2, "kind": "reference",
            "sample": "%d searches of %d rollouts (bs %d) by %d reference game threads + %d search threads, stub net (net time "
                      "excluded), %.1f s" % (len(r["search"]), rollouts, rollouts_per_batch, cores, cores, dt)}


def init_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("ELF_BENCH_BACKEND", "nccl")   # "gloo" only in the CPU test of this path
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1:
        # one process per GPU on one node: keep the host-side thread pools of the ranks from oversubscribing the cores
        torch.set_num_threads(max(1, len(os.sched_getaffinity(0)) // world))
    return rank, local_rank, world, dist


def reduce_max_sum(dist, dev, dt, count):
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    s_all = torch.tensor([count], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(s_all, op=dist.ReduceOp.SUM)
    return float(t_all.item()), int(s_all.item())


def run_board(args, rank, local_rank, world, dist, steps, warmup, with_cpu):
    import elf_amd
    n, boards = args.board_size, args.boards
    eng = elf_amd.GoEngine(n, boards, local_rank)
    dev = eng.device
    out = torch.empty((boards, 4), dtype=torch.int32, device=dev)
    total = warmup + steps
    seeds = [torch.from_numpy(seeds_for(rank, boards, r).view(np.int64)).to(dev) for r in range(total)]  # resident in HBM
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
    eng.close()
    if rank != 0:
        return None
    avg_kernel_s = float(np.mean(kern_ms)) / 1e3
    steps_per_launch = my_steps / steps
    achieved = steps_per_launch * STEP_BYTES[n] / avg_kernel_s / 1e9
    traffic = load_traffic("k_playout<%d>" % n)
    res = {
        "metric": "board_steps_per_sec (%dx%d GoState::forward + legal-move mask, random legal play to game end)" % (n, n),
        "value": steps_all / dt_max, "unit": "board_steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: %d concurrent %dx%d boards per GPU, config-2 random legal non-eye play "
                               "to game end, board-step kernel only (no net)" % (boards, n, n),
                   "boards_per_gpu": boards, "board_size": n, "board_steps_per_pass": steps_per_launch,
                   "parallelism": "independent boards per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "k_playout<%d>" % n, "avg_kernel_ms": avg_kernel_s * 1e3,
                     "algorithmic_bytes_per_step": STEP_BYTES[n],
                     "note": "algorithmic bytes = reference Board in+out + legal mask per step (SURVEY.md 8d); the kernel "
                             "keeps the position in LDS for the whole game, so this is a rate against the HBM roof; "
                             "traffic = PMC HBM bytes per launch from profiles/ (FETCH_SIZE x2 + WRITE_SIZE, KiB)"},
    }
    res["cpu_baseline"] = cpu_baseline_board(n) if with_cpu else None
    return res


def load_traffic(kernel):
    """HBM bytes per launch measured with rocprofv3 PMC passes (tools/gpu_round.sh), committed under profiles/."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(p)).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def mcts_traffic(n, rollouts_per_step):
    """PMC HBM bytes of the four search kernels per step, scaled per rollout from the profiled run (profiles/pmc_traffic.json)."""
    try:
        per = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["k_mcts_search<%d>" % n]["hbm_bytes_per_rollout"]
        return per * rollouts_per_step
    except Exception:
        return None


def run_mcts(args, rank, local_rank, world, dist, steps, warmup, with_cpu):
    import elf_amd
    from elf_amd.net import make_net
    n, G, K = args.board_size, args.games, args.rollouts_per_batch
    dev = torch.device("cuda", local_rank)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.net_dtype]
    net = None
    if args.net == "resnet":
        torch.backends.cudnn.benchmark = True
        net = make_net(n, args.net_blocks, args.net_dim, dev, dtype, channels_last=True, seed=0, fold_bn=not args.no_fold_bn)
        if args.no_fold_bn or dtype == torch.float32:
            args.net_impl = "eager"   # the fused epilogue is an fp16, BN-folded inference path
        if args.net_impl != "eager":
            from elf_amd.net import FusedInferenceNet
            net = FusedInferenceNet(net)
    feat_fmt = "f16_nhwc" if (args.features == "f16" or (args.features == "auto" and net is not None and dtype == torch.float16)) else "f32_nchw"
    from elf_amd.pipeline import PipelinedSelfPlay
    groups = max(1, args.groups)
    Gg = G // groups
    G = Gg * groups
    sp = PipelinedSelfPlay(groups=groups, seed=1234, game_idx_base=rank * G, wait_rows=bool(args.wait_rows), board_size=n, num_games=Gg,
                           device=local_rank,
                           mcts_rollout_per_thread=args.rollouts, mcts_rollout_per_batch=K, mcts_puct=1.5, mcts_virtual_loss=1,
                           mcts_persistent_tree=True, mcts_epsilon=0.25, mcts_alpha=0.03, komi=7.5, ply_pass_enabled=0,
                           policy_distri_cutoff=30, nodes_per_game=args.nodes_per_game, feature_format=feat_fmt)
    na = n * n + 1
    rows_max = sp.groups[0].max_rows
    # --net random: a peaky pseudo-random policy and a random value drawn on the GPU by torch (no conv net): isolates the
    # search kernels while still growing deep, narrow trees like a trained net does.  --net null: uniform prior, V = 0.
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    uni_pi = torch.full((rows_max, na), 1.0 / na, dtype=torch.float32, device=dev)
    zero_v = torch.zeros(rows_max, dtype=torch.float32, device=dev)
    rows_log = []
    if net is not None:
        # initialisation, not a step: the first call of each convolution shape runs MIOpen's find (tens of seconds on a fresh box).
        # Done here so that even --warmup 0 times steady-state steps only.
        with torch.no_grad():
            net({"s": sp.groups[0].s})
        torch.cuda.synchronize()
    graphs = {}
    if net is not None and args.net_graph:
        # one HIP graph per game group (PyTorch's CUDAGraph on ROCm): the ~85 kernel launches of a net call (41 convolutions, 41
        # epilogue passes, heads) become one graph launch; the graph reads the group's own "s" tensor, which is where the select
        # kernel writes the leaf features, and its output tensors are what the expand kernel reads
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
                out = net({"s": s})   # fixed shape [Gg*K, 18, N, N]: rows >= `rows` are stale and ignored (no MIOpen re-tuning)
            return out["pi"], out["V"]
        if args.net == "random":
            return (torch.softmax(4.0 * torch.randn((rows_max, na), device=dev, generator=gen), dim=1),
                    torch.tanh(0.5 * torch.randn((rows_max,), device=dev, generator=gen)))
        return uni_pi, zero_v

    def one(i):
        rows_log.append(sp.step(net_fn))   # every group: select -> leaf features -> net -> expand -> backup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        one(i)
    barrier()
    sp.timing = True
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        one(i)
    barrier()
    dt = time.perf_counter() - t0
    sp.timing = False
    my_rollouts = G * K * steps
    st = sp.stats()
    my_rows = int(sum(rows_log[warmup:])) if args.wait_rows else int(st["rows"] * steps / max(st["steps"] / groups, 1))
    sel_ms = float(np.sum([a.elapsed_time(b) for a, b in sp.t_select])) / steps
    exp_ms = float(np.sum([a.elapsed_time(b) for a, b in sp.t_expand])) / steps
    net_ms = float(np.mean([a.elapsed_time(b) for a, b in sp.t_net])) if sp.t_net else 0.0   # per net call (one group)
    dt_max, roll_all = reduce_max_sum(dist, dev, dt, my_rollouts)
    sp.close()
    if rank != 0:
        return None
    step_ms = dt_max / steps * 1e3
    # roofline of the search kernels (select+features / expand+backup), SURVEY.md 8d bytes with a depth estimate
    depth = st["node_visits"] / max(st["rollouts"], 1)   # measured mean descent depth (select kernel counter)
    bytes_per_step = G * K * depth * ROLLOUT_NODE_BYTES + my_rows / steps * ROLLOUT_EXPAND_BYTES
    search_s = (sel_ms + exp_ms) / 1e3
    achieved = bytes_per_step / search_s / 1e9
    res = {
        "metric": "mcts_rollouts_per_sec (self-play, %dx%d Go, %d rollouts/move, bs %d)" % (n, n, args.rollouts, K),
        "value": roll_all / dt_max, "unit": "rollouts/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: MCTS self-play bs=%d, %d rollouts/move, puct 1.5, vloss 1, Dirichlet 0.25/0.03, "
                               "persistent tree, %s, %d games per GPU in %d lock-step group(s) pipelined against the net"
                               % (K, args.rollouts, "random-init %d-block/%d-ch net on PyTorch-ROCm (%s, channels_last%s%s; leaf features %s)"
                                  % (args.net_blocks, args.net_dim, args.net_dtype, "" if args.no_fold_bn else ", eval BatchNorm folded into the convs",
                                     {"eager": "", "fused": ", conv epilogue = one elfnet_bias_act_f16 pass"}[args.net_impl] + (", one HIP graph per net call" if args.net_graph else ""),
                                     feat_fmt)
                                  if net is not None else "NO conv net (--net %s: search kernels only)" % args.net, G, groups),
                   "search_dtype": "f32 edge statistics (the reference's float), u16 board labels", "net_dtype": args.net_dtype if net is not None else None,
                   "games_per_gpu": G, "board_size": n, "rollouts_per_step": G * K, "net_rows_per_step": my_rows / steps,
                   "search_ms_per_step": sel_ms + exp_ms, "select_ms": sel_ms, "expand_backup_ms": exp_ms,
                   "step_minus_search_ms": step_ms - sel_ms - exp_ms, "groups": groups,
                   "moves_per_sec": roll_all / dt_max / args.rollouts,
                   "games_per_sec_est": roll_all / dt_max / args.rollouts / 250.0,
                   "games_per_sec_note": "rollouts/s / (rollouts per move x 250 moves per game); a full game does not fit a bench run",
                   "parallelism": "independent games per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": mcts_traffic(n, G * K),
                     "kernel": "k_mcts_select+k_mcts_features+k_mcts_expand+k_mcts_backup", "avg_kernel_ms": sel_ms + exp_ms,
                     "algorithmic_bytes_per_rollout": bytes_per_step / (G * K),
                     "note": "search kernels only (HIP events around begin_step/end_step on the launch stream); bytes per rollout = "
                             "depth x (362x20 B edge read + 24 B writes) + per expansion 52534 B (SURVEY.md 8d) with depth %.1f; "
                             "the path is latency-bound (pointer chasing down the tree), not HBM-bound -- see DESIGN.md" % depth,
                     "mean_depth": depth},
        "selfplay_stats": st,
    }
    if net is not None:
        # the kernel that dominates the timed region is not this library's: PyTorch-ROCm's convolution (north_star leaves the
        # net on PyTorch).  Reported for transparency: algorithmic flops of the 20x256 net per position / measured call time.
        d = n * n
        flops_pos = 2.0 * d * 9 * (18 * args.net_dim + 2 * args.net_blocks * args.net_dim * args.net_dim) \
            + 2.0 * d * args.net_dim * 3 + 2.0 * (2 * d * (d + 1) + d * 256 + 256)
        rows_call = Gg * K
        ach = flops_pos * rows_call / (net_ms / 1e3) / 1e12 if net_ms > 0 else None
        res["net_roofline"] = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": (ach / 2500.0) if ach else None,
                               "traffic": None, "kernel": "PyTorch-ROCm net call (MIOpen CK implicit-GEMM 3x3 conv x41 + elfnet_bias_act_f16 epilogues)",
                               "avg_call_ms": net_ms, "rows_per_call": rows_call, "flops_per_position": flops_pos,
                               "note": "not a kernel of this library; dense fp16/bf16 MFMA peak from MI355X_MICROARCH.md"}
    res["cpu_baseline"] = cpu_baseline_mcts(n, K) if with_cpu else None
    return res


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
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        from pyoracle import RefSelfPlay
    except Exception as e:
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "unavailable", "sample": str(e)}
    if not RefSelfPlay.available(n):
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "unavailable", "sample": "oracle/_ref/libelfsp%d.so not built" % n}
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))
    R = RefSelfPlay(n)
    samples = 4000 * cores
    steps, sec = R.train_bench(records_json, samples, cores, nfa)
    return {"value": samples / sec, "unit": "samples/s", "cores": cores, "kind": "reference", "board_steps_per_sec": steps / sec,
            "sample": "%d samples of the same records (%d replayed board steps) on %d host threads, %.1f s" % (samples, steps, cores, sec)}


def run_train(args, rank, local_rank, world, dist, steps, warmup, with_cpu):
    """the trainer's input pipeline.  One step = one "train" batch of --train-batch samples: draw (record, ply,
    D4) like GoGameTrain::act, replay each record to its ply and extract every field, in ONE k_replay_extract launch."""
    import ctypes as C
    import elf_amd
    from elf_amd.selfplay import MctsOptions, SpOptions
    n, B, R, nfa = args.board_size, args.train_batch, args.train_records, 1
    dev = torch.device("cuda", local_rank)
    plies = 320 if n == 19 else 70
    moves = synth_games(n, R, plies, dev, local_rank, 4242 + rank).cpu().numpy().astype(np.uint16)
    rng = np.random.default_rng(7 + rank)
    P = (n + 2) ** 2
    ld = elf_amd.ReplayLoader(board_size=n, capacity=R, batchsize=B, device=local_rank, num_future_actions=nfa, seed=1234 + 1000 * rank,
                              feature_format="f16_nhwc" if args.features in ("auto", "f16") else "f32_nchw")
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
        ld.put(r, rec)
        if with_cpu and r < 32:                                    # the CPU baseline reads the same records as Record JSON
            args_ = (C.byref(opt), moves[r].ctypes.data, plies, pol.ctypes.data, plies, val.ctypes.data, plies, C.c_float(rec["reward"]), 0, 2, 0, 0)
            k = L.elfrec_record_to_json(*args_, None, 0)
            buf = C.create_string_buffer(k + 1)
            L.elfrec_record_to_json(*args_, buf, k + 1)
            recs_json.append(buf.raw[:k].decode())
    out = ld._alloc(B)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    mt_sum = torch.zeros((), dtype=torch.int64, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    d = ld._draw
    for i in range(warmup):
        ld.sample(B, out=out)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        check_rc = ld.L.elftrain_draw(ld._h, B, nfa, C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()), C.c_void_p(d[2].data_ptr()), ld._stream())
        assert check_rc == 0
        ev[i][0].record()
        ld.extract(d[0, :B], d[1, :B], d[2, :B], out=out)          # the dominant kernel, on torch's current stream
        ev[i][1].record()
        mt_sum += out["move_idx"].sum()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    replayed = int(mt_sum.item())
    dt_max, samples_all = reduce_max_sum(dist, dev, dt, B * steps)
    ld.close()
    if rank != 0:
        return None
    per_sample = replayed / (B * steps) * STEP_BYTES[n] + (26728 if n == 19 else 6008) + P + 4 * (n * n + 1) + 40
    achieved = B * per_sample / (kern_ms / 1e3) / 1e9
    res = {
        "metric": "train_samples_per_sec (%dx%d replay to a random ply + every field of the reference's train batch)" % (n, n),
        "value": samples_all / dt_max, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt_max / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16",
        "data": "synthetic",
        "config": {"workload": "SURVEY.md 8f-1 trainer input pipeline: batch %d, %d records of %d plies (random legal play on the device "
                               "engine), one MCTS policy per ply, num_future_actions %d, s rows %s" % (B, R, plies, nfa, ld.f16 and "f16_nhwc" or "f32_nchw"),
                   "batch": B, "records": R, "board_size": n, "mean_replayed_plies": replayed / (B * steps),
                   "replayed_board_steps_per_sec": replayed / dt, "parallelism": "independent samples per GPU, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": load_traffic("k_replay_extract<%d>" % n), "kernel": "k_replay_extract<%d>" % n, "avg_kernel_ms": kern_ms,
                     "algorithmic_bytes_per_sample": per_sample,
                     "note": "bytes per sample = replayed plies x 8730 B (reference Board in+out+mask per forward, SURVEY.md 8d) + 26728 B "
                             "features + 441 B policy row + 362 x 4 B scores + scalars; the replay itself runs in LDS, so like k_playout "
                             "this is a rate against the HBM roof, not HBM traffic"},
    }
    res["cpu_baseline"] = cpu_baseline_train(n, "[" + ",".join(recs_json) + "]", nfa) if with_cpu else None
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["mcts", "board", "train", "both"], default="both",
                    help="both (default) = mcts headline + board_step + train_loader sub-results")
    ap.add_argument("--train-batch", type=int, default=2048)
    ap.add_argument("--train-records", type=int, default=256)
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--board-size", type=int, default=19)
    ap.add_argument("--games", type=int, default=256, help="games per GPU (split over --groups)")
    ap.add_argument("--groups", type=int, default=2, help="lock-step game groups pipelined against the net (1 = serial)")
    ap.add_argument("--rollouts", type=int, default=8192)
    ap.add_argument("--rollouts-per-batch", type=int, default=16)
    ap.add_argument("--nodes-per-game", type=int, default=None)
    ap.add_argument("--net", choices=["resnet", "random", "null"], default="resnet")
    ap.add_argument("--no-fold-bn", action="store_true")
    ap.add_argument("--net-blocks", type=int, default=20)
    ap.add_argument("--net-dim", type=int, default=256)
    ap.add_argument("--net-dtype", choices=["fp16", "bf16", "fp32"], default="fp16")
    ap.add_argument("--net-impl", choices=["eager", "fused"], default="fused",
                    help="eager: plain PyTorch modules; fused: PyTorch convs + one HIP epilogue pass per conv (elfnet_bias_act_f16)")
    ap.add_argument("--net-graph", type=int, default=1, help="1: replay each group's net call as one HIP graph; 0: eager launches")
    ap.add_argument("--features", choices=["auto", "f32", "f16"], default="auto",
                    help="leaf feature rows: f32 NCHW (reference layout) or f16 channels_last (auto: f16 when the net is fp16)")
    ap.add_argument("--wait-rows", type=int, default=0, help="1: the host waits for the row count of every step (drop-in path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, local_rank, world, dist = init_dist(args)
    with_cpu = (not args.no_cpu_baseline) and world == 1
    res = None
    if args.workload in ("mcts", "both"):
        steps = args.steps if args.steps is not None else 40
        warmup = args.warmup if args.warmup is not None else 8
        res = run_mcts(args, rank, local_rank, world, dist, steps, warmup, with_cpu)
    if args.workload in ("board", "both"):
        bsteps = args.steps if (args.steps is not None and args.workload == "board") else 20
        bwarm = args.warmup if (args.warmup is not None and args.workload == "board") else 3
        b = run_board(args, rank, local_rank, world, dist, bsteps, bwarm, with_cpu)
        if args.workload == "board":
            res = b
        elif rank == 0:
            res["board_step"] = b
    if args.workload in ("train", "both"):
        tsteps = args.steps if (args.steps is not None and args.workload == "train") else 20
        twarm = args.warmup if (args.warmup is not None and args.workload == "train") else 3
        t = run_train(args, rank, local_rank, world, dist, tsteps, twarm, with_cpu)
        if args.workload == "train":
            res = t
        elif rank == 0:
            res["train_loader"] = t
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
