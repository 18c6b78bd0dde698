#!/bin/bash
# GPU-box tool: cycle attribution of k_mcts_expand by phase. Builds a PROFILE variant of the library in place (the committed
# build is restored afterwards), runs the search-only bench and prints the accumulated s_memtime ticks per phase.
set -e
make -C elf_amd/csrc clean >/dev/null
make -C elf_amd/csrc HIPCC="/opt/rocm/bin/hipcc -DELF_PROFILE_EXPAND" >/dev/null 2>&1
python - <<'PY'
import ctypes as C, json, os, subprocess, sys
sys.path.insert(0, ".")
# ELF_NET=random16: near-uniform fp16-grid replies like the headline's random-init net (prior ties in every row)
sys.argv = ["bench.py", "--workload", "mcts", "--net", os.environ.get("ELF_NET", "random"), "--games", "1024", "--groups", "1", "--nodes-per-game", "8192",
            "--rollouts", "2048", "--warmup", "24", "--steps", "32", "--no-cpu-baseline", "--no-sub", "--pregrow", "0", "--features", "f16"]
import bench, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.load(open("bench_full.json"))     # the full report beside the compact line (selfplay_stats_window lives there)
print("search-only (profile build)", d["value"], d["config"]["expand_backup_ms"])
import elf_amd
L = C.CDLL(elf_amd._lib.LIB_PATH)
out = (C.c_uint64 * 8)()
L.elfprof_expand_phases(out)
names = ["row map + parked legal mask", "introsort loop of the std::sort replay (ties)", "reply read / coords / validity / keys", "bitonic sort (512 slots)",
         "sorted rows to LDS + tie test (+ final stable sort and filter on ties)", "sequential fp32 normalisation", "unordered_map iteration order", "edge records to HBM"]
tot = sum(out)
rows = d["selfplay_stats_window"]["rows"]
print("  total %.0f ticks per row (%d rows)" % (tot / rows, rows))
for n, v in zip(names, out):
    print("  %-40s %6.2f %%  %8.0f ticks/row" % (n, 100.0 * v / tot, v / rows))
import numpy as np
mx = (C.c_uint64 * 65536)()
L.elfprof_expand_rowmax(mx)
a = np.array(mx[:16384], dtype=np.float64)
a = a[a > 0]
print("  longest row per block id (ticks): median %.0f  p90 %.0f  p99 %.0f  max %.0f  (mean row %.0f)" % (np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max(), tot / rows))
PY
make -C elf_amd/csrc clean >/dev/null
make -C elf_amd/csrc >/dev/null 2>&1
