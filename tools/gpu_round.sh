#!/bin/bash
# One GPU-box visit: parity tests, bench lines, rocprofv3 kernel stats and PMC passes.
# Usage (from repo root, via gpurun): bash tools/gpu_round.sh <tag> [quick]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench (default: mcts headline + board_step + train_loader)"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
echo "== bench mcts null net (search kernels only)"
NULLNET="python bench.py --workload mcts --net random --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 --no-cpu-baseline"
timeout 300 $NULLNET > $OUT/bench_nullnet.json 2> $OUT/bench_nullnet.err; echo "rc=$?"
cat $OUT/bench_nullnet.json
echo "== bench config 5: 9x9, 65536 boards"
timeout 300 python bench.py --workload board --board-size 9 --boards 65536 --steps 10 --warmup 2 > $OUT/bench_board9.json 2> $OUT/bench_board9.err; echo "rc=$?"
cat $OUT/bench_board9.json
echo "== bench mcts eager net (no fused epilogue, fp32 NCHW features) for comparison"
timeout 400 python bench.py --workload mcts --net-impl eager --features f32 --steps 16 --warmup 6 --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err; echo "rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_eager.json'));print('eager:', d['value'], d['ms_per_step'])"
[ "$2" = "quick" ] && exit 0
PROF_board="python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline"
PROF_mcts="$NULLNET"
PROF_train="python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline"
PROF_feat="python tools/feat_bench.py"
for W in board mcts train feat; do
  eval CMD=\$PROF_$W
  echo "== rocprofv3 stats $W"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o stats --output-format csv -- $CMD > $OUT/stats_$W.log 2>&1
  echo "== pmc passes $W"
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch_$W.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_write_$W.log 2>&1
  if [ $W != feat ]; then
    timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_lds_$W.log 2>&1
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_sq_$W.log 2>&1
  fi
  python tools/summarize_prof.py $OUT $W > $OUT/summary_$W.txt 2>&1
  head -60 $OUT/summary_$W.txt
done
echo "== rocprofv3 stats, mcts with the real net (kernel time shares; MIOpen find results are cached by the runs above)"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- python bench.py --workload mcts --steps 10 --warmup 4 --no-cpu-baseline > $OUT/stats_mctsnet.log 2>&1
python tools/summarize_prof.py $OUT mctsnet > $OUT/summary_mctsnet.txt 2>&1
head -25 $OUT/summary_mctsnet.txt
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
du -sh $OUT
