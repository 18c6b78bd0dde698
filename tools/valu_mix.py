#!/usr/bin/env python
"""Static VALU class mix of the library's kernels (build container; hipcc cross-compiles): which share of a kernel's vector
instructions issues at the plain VOP1/VOP2 rate and which at the quarter rate that tools/issue_probe.hip measured for VOP3 /
VOPC / DPP / SDWA / lane-access / 64-bit-shift / 24-bit-multiply instructions (profiles/r04b_issue_probe.json: 1.82 vs 1.0 per
clock and CU).  Writes profiles/valu_mix.json, which bench.py uses for the class-weighted VALU issue roof.
The counts are STATIC (every instruction of the kernel's text once, cold paths included), so the share is an estimate of the
dynamic mix, not a measurement; the dynamic totals come from the PMC passes (profiles/pmc_issue.json)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = {"elf_amd.hip": ["k_playout", "k_extract_agz", "k_forward", "k_legal_mask"], "mcts_capi.hip": ["k_mcts_select", "k_mcts_leafstate", "k_mcts_expand", "k_mcts_backup", "k_mcts_features"],
       "train_capi.hip": ["k_replay_extract", "k_replay_checkpoint"]}
PLAIN_CYC, SLOW_CYC = 4.0 / 1.82, 4.0      # SIMD cycles per wave64 instruction (4 SIMDs per CU): measured at 8 waves per SIMD


def classify(op):
    if not op.startswith("v_"):
        return None
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "slow"
    if op.endswith(("_dpp", "_sdwa", "_e64")) or "_e64_dpp" in op:
        return "slow"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "slow"
    if re.match(r"v_(mul_|mad_|fma_|div_|rcp_|rsq_|sqrt_|exp_|log_|sin_|cos_|bfe_|bfi_|alignbit|alignbyte|lshl_add|add_lshl|lshl_or|and_or|or3|xad|add3|perm|lshlrev_b64|lshrrev_b64|ashrrev_i64|mbcnt|cndmask|med3|min3|max3|cvt_f64|cvt_f32_f64|mul_f64|add_f64|fma_f64|mov_b64|lshl_add_u64|pk_)", op):
        return "slow"
    if op.endswith("_e32"):
        return "plain"
    return "slow"       # VOP3-only opcodes print without a suffix


def main():
    from elf_amd._lib import kernel_source_hash
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for src, kernels in SRC.items():
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-c",
                            os.path.join(ROOT, "elf_amd", "csrc", src), "-o", os.path.join(td, "x.o"), "--save-temps=obj"],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
            asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
            text = open(os.path.join(td, asm)).read()
            os.remove(os.path.join(td, asm))
            # function bodies by label (a line scan: the text also holds out-of-line device functions, which have no kernel descriptor)
            kernel_names = set(re.findall(r"^\s*\.amdhsa_kernel (\S+)", text, re.M))
            bodies, cur = {}, None
            for line in text.splitlines():
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    cur = m.group(1)
                    bodies[cur] = []
                    continue
                if line.startswith(".Lfunc_end"):
                    cur = None
                    continue
                if cur is not None:
                    bodies[cur].append(line)
            for name in sorted(kernel_names & set(bodies)):
                body = "\n".join(bodies[name])
                dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
                short = re.sub(r"^void (elfgo::)?", "", dem).split("(")[0]
                if not any(short.startswith(k) for k in kernels):
                    continue
                c = collections.Counter()
                ops = collections.Counter()
                for line in body.splitlines():
                    line = line.strip()
                    if not line or line[0] in ".;/" or line.endswith(":"):
                        continue
                    op = line.split()[0]
                    k = classify(op)
                    if k:
                        c[k] += 1
                        ops[op] += 1
                    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm", "s_load", "s_barrier", "s_sleep", "s_setprio")):
                        c["salu"] += 1
                tot = c["plain"] + c["slow"]
                if not tot:
                    continue
                cyc = (c["plain"] * PLAIN_CYC + c["slow"] * SLOW_CYC) / tot
                res[short] = {"valu_static": tot, "plain": c["plain"], "quarter_rate": c["slow"], "salu_static": c["salu"],
                              "quarter_rate_share": c["slow"] / tot, "simd_cycles_per_valu": cyc, "valu_peak_per_clk_per_cu": 4.0 / cyc,
                              "valu_peak_ginst": 4.0 / cyc * 256 * 2.4, "top_ops": ops.most_common(12)}
    res["_source"] = {"kernel_source_hash": kernel_source_hash(), "probe": "profiles/r04b_issue_probe.json",
                      "note": "static instruction census of the compiled kernels; plain = VOP1/VOP2 e32 without DPP/SDWA (measured 1.82 per clock and "
                              "CU), quarter_rate = VOP3 / VOPC / DPP / SDWA / lane access / 64-bit shifts / multiplies (measured 1.0 per clock and CU)"}
    json.dump(res, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
    for k, v in res.items():
        if k != "_source":
            print("%-40s valu %5d  quarter-rate share %.2f  VALU peak %.0f G/s  salu %d" % (k[:40], v["valu_static"], v["quarter_rate_share"], v["valu_peak_ginst"], v["salu_static"]))


if __name__ == "__main__":
    main()
