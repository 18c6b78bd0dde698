#!/bin/bash
# Final visit of a round: parity tests, smoke, the default bench line (what the driver runs), search-only bench.  Usage: bash tools/gpu_final.sh <tag>
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
python -c "
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'], d['net_roofline']['avg_call_ms'])
for k in ('board_step','board_step_9x9','train_loader','selfplay_games'): print(k, d[k]['value'], d[k].get('roofline',{}).get('frac'))
f=d['feature_extract']; print('feature', f['f32']['avg_kernel_ms'], f['f32']['roofline']['achieved'], f['f16']['avg_kernel_ms'], f['f16']['roofline']['achieved'])
print('boundary', d['boundary']['pinned_host']['rollouts_per_sec'], d['boundary']['device_resident']['rollouts_per_sec'])
print('cpu', d['cpu_baseline']['value'], d['board_step']['cpu_baseline']['value'])"
timeout 300 python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline > $OUT/bench_search_only.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/bench_search_only.json'));print('search-only', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'])"
