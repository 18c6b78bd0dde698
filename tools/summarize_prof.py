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
