#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats and PMC passes.
# Usage (from repo root, via gpurun): bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
BENCH_PROF="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
echo "== rocprofv3 stats"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $BENCH_PROF > $OUT/stats.log 2>&1
echo "== pmc passes"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $BENCH_PROF > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $BENCH_PROF > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds -o pmc --output-format csv -- $BENCH_PROF > $OUT/pmc_lds.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc --output-format csv -- $BENCH_PROF > $OUT/pmc_sq.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep merged output small: drop bulky per-dispatch traces, keep stats + pmc csvs
find $OUT -name '*kernel_trace.csv' -size +4M -delete
du -sh $OUT
