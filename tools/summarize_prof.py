#!/usr/bin/env python
"""Condense rocprofv3 CSVs of one tools/profile_all.sh visit into a text summary (goes into profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
sfx = ("_" + tag) if tag else ""


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("# rocprofv3 --kernel-trace --stats (kernel_stats)")
for f in find("stats%s/**/*kernel_stats.csv" % sfx):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:16]:
        print("%-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

# the kernel TRACE, when it is still there (profile_all.sh deletes it after this script): the same table for the steady state only,
# i.e. for the dispatches that start after MIOpen's find is over (its last naive_conv_* benchmark kernel) -- what the timed region of
# bench.py looks like, without the one-off initialisation of a fresh process
for f in find("stats%s/**/*kernel_trace.csv" % sfx):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames or []
        c_name = next((c for c in cols if c.lower() in ("kernel_name", "name")), None)
        c_start = next((c for c in cols if "start" in c.lower()), None)
        c_end = next((c for c in cols if "end" in c.lower()), None)
        if not (c_name and c_start and c_end):
            continue
        rows = [(r[c_name], int(r[c_start]), int(r[c_end])) for r in rd]
    t0 = max([e for n, s, e in rows if n.startswith("naive_conv")] or [0])
    acc = defaultdict(lambda: [0, 0])
    for n, s_, e in rows:
        if s_ >= t0:
            acc[n][0] += 1
            acc[n][1] += e - s_
    tot = sum(v[1] for v in acc.values()) or 1
    span = (max(e for n, s_, e in rows) - max(t0, min(s_ for n, s_, e in rows))) or 1
    print("\n# steady state: dispatches after the last naive_conv_* kernel (MIOpen find) -- %d of %d dispatches, %.3f s of GPU timeline"
          % (sum(v[0] for v in acc.values()), len(rows), span / 1e9))
    for n, (c, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
        print("%-60s calls=%d total_ns=%d avg_ns=%.1f pct=%.2f" % (n[:60], c, ns, ns / c, 100.0 * ns / tot))

print("\n# PMC counters: per kernel, mean per dispatch")
for d in ("pmc_fetch", "pmc_write", "pmc_lds", "pmc_sq"):
    for f in find(d + sfx + "/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "memset" in k or "elementwise" in k or "reduce" in k:
                continue
            for c, v in sorted(cs.items()):
                print("%-48s %-24s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
