#!/bin/bash
# One GPU-box visit of round 2: parity tests, the bench line, rocprofv3 kernel stats and PMC passes.
# Usage (from the repo root, via gpurun): bash tools/gpu_round2.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
echo "== bench (default line)"
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
NULLNET="python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
echo "== bench search-only"
timeout 300 $NULLNET > $OUT/bench_search_only.json 2> $OUT/bench_search_only.err; echo "rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_search_only.json'));print('search-only', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'])"
echo "== bench search-only, 2 search threads x 8"
timeout 300 python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 1024 --mcts-threads 2 --rollouts-per-batch 8 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline > $OUT/bench_search_threads2.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_search_threads2.json'));print('threads2', d['value'], d['ms_per_step'], d['config']['mean_depth'])"
PROF_board="python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline"
PROF_board9="python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline"
PROF_mcts="$NULLNET"
PROF_train="python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline"
PROF_feat="python bench.py --workload feature --steps 20 --warmup 3"
for W in board board9 mcts train feat; do
  eval CMD=\$PROF_$W
  echo "== rocprofv3 stats $W"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o stats --output-format csv -- $CMD > $OUT/stats_$W.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch_$W.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_write_$W.log 2>&1
  if [ $W != feat ]; then
    timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_lds_$W.log 2>&1
    timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_sq_$W.log 2>&1
  fi
  python tools/summarize_prof.py $OUT $W > $OUT/summary_$W.txt 2>&1
  grep -E "k_playout|k_mcts|k_replay|k_extract" $OUT/summary_$W.txt | head -12
done
echo "== rocprofv3 stats, headline (with the real net)"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_mctsnet.log 2>&1
python tools/summarize_prof.py $OUT mctsnet > $OUT/summary_mctsnet.txt 2>&1
head -12 $OUT/summary_mctsnet.txt
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
du -sh $OUT
