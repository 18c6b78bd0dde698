#!/usr/bin/env python
"""Refresh profiles/pmc_issue.json (instructions per unit of work of the LDS-resident kernels) from the PMC summaries of one
tools/profile_all.sh visit.  usage: python tools/update_issue.py gpurun_out/<tag> <tag>
Units per launch come from the bench line that ran under the same rocprofv3 pass (stats_<workload>.log)."""
import json, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, kernel):
    acc = {}
    for line in open(path):
        m = re.match(r"(.{48}) (\S+)\s+n=(\d+) mean=(\S+)", line.rstrip("\n"))
        if m and kernel in m.group(1):
            acc[m.group(2)] = (float(m.group(4)), int(m.group(3)))
    return acc


def bench_line(path):
    for line in open(path):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


p = os.path.join(root, "profiles", "pmc_issue.json")
res = json.load(open(p)) if os.path.exists(p) else {}
for wl, kernel, key, unit, profile in (("board", "k_playout<19>", "k_playout<19>", "board step", "board"),
                                       ("board9", "k_playout<9>", "k_playout<9>", "board step", "board9"),
                                       ("train", "k_replay_extract<19>", "k_replay_extract<19>", "board step forwarded from the checkpoint", "train")):
    s = os.path.join(out, "summary_%s.txt" % wl)
    log = os.path.join(out, "stats_%s.log" % wl)
    if not (os.path.exists(s) and os.path.exists(log)):
        continue
    c = counters(s, kernel.split("<")[0])
    d = bench_line(log)
    if d is None or "SQ_INSTS_VALU" not in c:
        continue
    cfg = d["config"]
    if wl == "train" and "mean_replayed_plies" not in cfg:
        # the compact driver line of an older bench.py did not carry the trainer's per-launch figures: they are the same for the same
        # seeds, take them from a stored full report
        full = json.load(open(os.path.join(root, "profiles", os.environ.get("ELF_FULL_REPORT", "r06p_bench_full.json"))))
        cfg = dict(full["train_loader"]["config"], **cfg)
    units = cfg["board_steps_per_pass"] if wl.startswith("board") else cfg.get("mean_forwarded_plies", cfg["mean_replayed_plies"]) * cfg.get("samples_per_launch", cfg["batch"])
    res[key] = {"valu_per_unit": c["SQ_INSTS_VALU"][0] / units, "salu_per_unit": c["SQ_INSTS_SALU"][0] / units,
                "lds_per_unit": c.get("SQ_INSTS_LDS", (0.0, 0))[0] / units, "unit": unit,
                # north_star: LDS-bank utilisation of the board step.  SQ_LDS_IDX_ACTIVE = cycles the LDS index pipe is busy,
                # SQ_LDS_BANK_CONFLICT = cycles it stalls on a bank conflict, SQ_BUSY_CYCLES = SQ-busy cycles of the launch
                # (all summed over the chip's shader engines / CUs as rocprofv3 reports them)
                "lds_bank_conflict_frac": (c["SQ_LDS_BANK_CONFLICT"][0] / c["SQ_LDS_IDX_ACTIVE"][0]) if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"][0] else None,
                "lds_active_cycles_per_unit": (c["SQ_LDS_IDX_ACTIVE"][0] / units) if "SQ_LDS_IDX_ACTIVE" in c else None,
                "wave_issue_frac": (c["SQ_ACTIVE_INST_ANY"][0] / c["SQ_WAVE_CYCLES"][0]) if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c else None,
                "wave_wait_frac": (c["SQ_WAIT_ANY"][0] / c["SQ_WAVE_CYCLES"][0]) if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c else None,
                "profile": "profiles/%s_%s_rocprofv3.txt" % (tag, profile),
                "note": "rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS, mean over %d launches of the bench.py --workload %s run "
                        "of tools/profile_all.sh (%.0f units per launch)" % (c["SQ_INSTS_VALU"][1], wl, units)}
# round 6: the search kernels, per ROLLOUT of the search-only profile run (one launch = the rollouts of one game group)
s_m, log_m = os.path.join(out, "summary_mcts.txt"), os.path.join(out, "stats_mcts.log")
if os.path.exists(s_m) and os.path.exists(log_m):
    d = bench_line(log_m)
    if d is not None:
        per_launch = d["config"]["rollouts_per_step"] / max(1, d["config"].get("groups", 1))
        for kern in ("k_mcts_select", "k_mcts_leafstate", "k_mcts_features", "k_mcts_expand", "k_mcts_backup"):
            c = counters(s_m, kern)
            if "SQ_INSTS_VALU" not in c:
                continue
            res[kern + "<19>"] = {"valu_per_unit": c["SQ_INSTS_VALU"][0] / per_launch, "salu_per_unit": c["SQ_INSTS_SALU"][0] / per_launch,
                                  "lds_per_unit": c.get("SQ_INSTS_LDS", (0.0, 0))[0] / per_launch, "unit": "rollout",
                                  "wave_issue_frac": (c["SQ_ACTIVE_INST_ANY"][0] / c["SQ_WAVE_CYCLES"][0]) if "SQ_ACTIVE_INST_ANY" in c and "SQ_WAVE_CYCLES" in c else None,
                                  "wave_wait_frac": (c["SQ_WAIT_ANY"][0] / c["SQ_WAVE_CYCLES"][0]) if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c else None,
                                  "profile": "profiles/%s_mcts_search_only_rocprofv3.txt" % tag,
                                  "note": "rocprofv3 --pmc SQ_INSTS_*, mean over %d launches of the search-only profile run (%.0f rollouts per launch)"
                                          % (c["SQ_INSTS_VALU"][1], per_launch)}
sys.path.insert(0, root)
from elf_amd._lib import KERNEL_SOURCES, kernel_source_hash   # noqa: E402
res["_source"] = {"kernel_source_hash": kernel_source_hash(), "files": ["elf_amd/csrc/" + f for f in KERNEL_SOURCES], "visit": tag,
                  "note": "sha256[:16] over the kernel sources the PMC passes were run on; bench.py prints pmc_source_match and withholds "
                          "the issue-roof fraction / PMC traffic when the sources have changed since"}
json.dump(res, open(p, "w"), indent=1)
print(json.dumps(res, indent=1))
