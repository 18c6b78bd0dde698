#!/bin/bash
# GPU-box tool: SQ issue counters of the search kernels with tie-heavy replies (ELF_NET=random16) or without (random).
# Two separate --pmc passes (never combined with a trace); prints per-kernel sums per counter and per dispatch.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_expand_${ELF_NET:-random16}
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --workload mcts --net ${ELF_NET:-random16} --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 24 --steps 32 --no-cpu-baseline --no-sub --pregrow 0 --features f16"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o pmc --output-format csv -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b -o pmc --output-format csv -- $CMD > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT_DIR", "")
for d in sorted(glob.glob("gpurun_out/pmc_expand_*/[ab]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_mcts" not in k: continue
            k = k.split("<")[0].replace("void elfgo::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k in sorted(acc):
            print(d, k, " ".join("%s=%.4g/disp" % (c, v / cnt[(k, c)]) for c, v in sorted(acc[k].items())))
PY
find $OUT -name '*.csv' -size +1M -delete; find $OUT -name '*.db' -delete
