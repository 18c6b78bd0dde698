#!/usr/bin/env python
"""Refresh profiles/pmc_traffic.json from the PMC summaries of one `tools/profile_all.sh <tag> pmc stats` visit.
usage: python tools/update_traffic.py gpurun_out/<tag> <tag>   (FETCH_SIZE doubled: MI355X_MICROARCH.md, gfx950 counts 128-B requests as 64 B)"""
import json, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def counters(path):
    acc = {}
    for line in open(path):
        m = re.match(r"(.{48}) (\S+)\s+n=(\d+) mean=(\S+)", line.rstrip("\n"))
        if m and m.group(2) in ("FETCH_SIZE", "WRITE_SIZE"):
            acc.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(4)), int(m.group(3)))
    return acc
def hbm(c):  # KiB counters -> bytes
    return (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0
p = os.path.join(root, "profiles", "pmc_traffic.json")
res = json.load(open(p)) if os.path.exists(p) else {}
b = counters(os.path.join(out, "summary_board.txt"))
for k, c in b.items():
    if "k_playout<19>" in k:
        res["k_playout<19>"] = {"hbm_bytes_per_launch": hbm(c), "fetch_size_kib": c["FETCH_SIZE"][0], "write_size_kib": c["WRITE_SIZE"][0],
                                "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over %d launches of bench.py --workload board "
                                        "(4096 boards, 1.865M board steps per launch); FETCH_SIZE doubled per MI355X_MICROARCH.md; source profiles/%s_board_rocprofv3.txt" % (c["FETCH_SIZE"][1], tag)}
m = counters(os.path.join(out, "summary_mcts.txt"))
per, n = {}, 0
for k, c in m.items():
    for name in ("k_mcts_select", "k_mcts_leafstate", "k_mcts_leafindex", "k_mcts_features", "k_mcts_expand", "k_mcts_backup"):
        if name in k:
            per[name] = hbm(c); n = c["FETCH_SIZE"][1]
if per:
    # profile_all.sh profiles the search-only run at 4096 games in two groups: one launch of a per-game kernel covers 2048 games x 16
    roll = int(os.environ.get("ELF_PMC_ROLLOUTS_PER_LAUNCH", "32768"))
    res["k_mcts_search<19>"] = {"hbm_bytes_per_rollout": sum(per.values()) / roll, "per_kernel_bytes_per_launch": per, "rollouts_per_launch": roll,
                                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over %d launches of bench.py --workload mcts --net random --games 4096 --groups 2 "
                                        "--rollouts 2048 (32768 rollouts per launch of a 2048-game group); FETCH_SIZE doubled per MI355X_MICROARCH.md; source profiles/%s_mcts_search_only_rocprofv3.txt" % (n, tag)}
t = os.path.join(out, "summary_train.txt")
if os.path.exists(t):
    for k, c in counters(t).items():
        if "k_replay_extract" in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res["k_replay_extract<19>"] = {"hbm_bytes_per_launch": hbm(c), "fetch_size_kib": c["FETCH_SIZE"][0], "write_size_kib": c["WRITE_SIZE"][0],
                                           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean over %d launches of bench.py --workload train "
                                                   "(32768 samples = 16 train batches of 2048 per launch, the bench default --train-prefetch 16; each sample loads its record's checkpoint and forwards <= 31 plies); FETCH_SIZE doubled per MI355X_MICROARCH.md; source profiles/%s_train_rocprofv3.txt" % (c["FETCH_SIZE"][1], tag)}
for key, algo in (("f32", 26728), ("f16", 13732)):
    t = os.path.join(out, "summary_feat%s.txt" % ("32" if key == "f32" else "16"))
    if not os.path.exists(t):
        continue
    for k, c in counters(t).items():
        if "k_extract_agz" in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res["k_extract_agz<19>:%s" % key] = {
                "hbm_bytes_per_launch": hbm(c), "fetch_size_kib": c["FETCH_SIZE"][0], "write_size_kib": c["WRITE_SIZE"][0],
                "note": "bench.py --workload feature --feature-formats %s: 16384 rows per launch (%d launches); algorithmic bytes %d per row; "
                        "source profiles/%s_feat%s_rocprofv3.txt" % (key, c["FETCH_SIZE"][1], algo, tag, "32" if key == "f32" else "16")}
sys.path.insert(0, root)
from elf_amd._lib import KERNEL_SOURCES, kernel_source_hash   # noqa: E402
res["_source"] = {"kernel_source_hash": kernel_source_hash(), "files": ["elf_amd/csrc/" + f for f in KERNEL_SOURCES], "visit": tag,
                  "note": "sha256[:16] over the kernel sources the PMC passes were run on; bench.py prints pmc_source_match and withholds "
                          "the issue-roof fraction / PMC traffic when the sources have changed since"}
res.pop("k_extract_agz<19>", None)   # the round-2 entry that averaged both row formats
json.dump(res, open(p, "w"), indent=1)
print(json.dumps(res, indent=1))
