#!/bin/bash
# GPU-box visit r02e: parity tests, smoke, default bench line, search-only bench, rocprof kernel stats of the search-only run
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
python -c "import json;d=json.load(open('$OUT/bench.json'));print('headline', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth']); print('board', d['board_step']['value'], d['board_step'].get('parity_mismatches'))"
NULLNET="python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
timeout 300 $NULLNET > $OUT/bench_search_only.json 2> $OUT/bench_search_only.err; echo "rc=$?"
python -c "import json;d=json.load(open('$OUT/bench_search_only.json'));print('search-only', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'])"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_mcts -o stats --output-format csv -- $NULLNET > $OUT/stats_mcts.log 2>&1
python tools/summarize_prof.py $OUT mcts > $OUT/summary_mcts.txt 2>&1
head -14 $OUT/summary_mcts.txt
find $OUT -name '*kernel_trace.csv' -size +4M -delete
du -sh $OUT
