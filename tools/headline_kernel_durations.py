#!/usr/bin/env python
"""GPU-box tool: per-dispatch durations of the search kernels inside the HEADLINE run (conv net between the steps) from a rocprofv3
kernel trace, split into the untimed tree-growing prologue (random replies, steps 1 ms apart) and the steps that follow a net call
(740 ms apart: caches and TLBs cold, every row of the reply carries prior ties).  Usage (from the repo root):
    rocprofv3 --kernel-trace -d gpurun_out/X -o t --output-format csv -- python bench.py --workload mcts --steps 8 --warmup 2 --no-cpu-baseline --no-sub
    python tools/headline_kernel_durations.py gpurun_out/X"""
import csv
import glob
import sys

import numpy as np

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["k_mcts_select", "k_mcts_leafstate", "k_mcts_leafindex", "k_mcts_rowbase", "k_mcts_features", "k_mcts_expand", "k_mcts_backup"]
# a dispatch "follows a net call" when a convolution kernel ran since the previous dispatch of the same search kernel
last_conv = -1
seen = {}
out = {n: {"prologue": [], "after_net": []} for n in names}
for i, r in enumerate(rows):
    nm = r["Kernel_Name"]
    if "kernel_grouped_conv" in nm or "igemm" in nm or "naive_conv" in nm:
        last_conv = i
        continue
    for n in names:
        if n + "<" in nm:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            out[n]["after_net" if last_conv > seen.get(n, -2) and last_conv >= 0 else "prologue"].append(d)
            seen[n] = i
print("%-18s %28s   %28s" % ("kernel", "prologue (random replies)", "after a net call"))
tot = [0.0, 0.0]
for n in names:
    a, b = np.array(out[n]["prologue"]), np.array(out[n]["after_net"])
    f = lambda x: "n=%4d mean %7.1f med %7.1f us" % (len(x), x.mean() if len(x) else 0, np.median(x) if len(x) else 0)
    print("%-18s %28s   %28s" % (n, f(a), f(b)))
    tot[0] += a.mean() if len(a) else 0
    tot[1] += b.mean() if len(b) else 0
print("sum of means: prologue %.1f us, after a net call %.1f us" % tuple(tot))
