#!/bin/bash
# GPU visit r04b: checkpointed replay loader (tests, bench lines, kernel stats), games/s by phase, extended issue probe.
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 build/issue_probe > $OUT/issue_probe.json 2> $OUT/issue_probe.err; echo "issue_probe rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/issue_probe.json'))
for r in d['rows']:
    if r['waves_per_simd'] in (1, 8): print('%-44s w=%d  cyc/inst/wave %.2f  inst/clk/cu(wall) %.3f' % (r['kind'][:44], r['waves_per_simd'], r['cycles_per_instruction_per_wave'], r['inst_per_clk_per_cu_wall_2.4GHz']))
PY
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q --timeout 300 --tb=short -rf -x > $OUT/pytest_train.log 2>&1; echo "train tests rc=$?"
tail -15 $OUT/pytest_train.log
for V in "queues f16" "uniform f16" "queues f32"; do
  set -- $V
  timeout 300 python bench.py --workload train --train-sampler $1 --features $2 --steps 20 --warmup 3 > $OUT/train_$1_$2.json 2> $OUT/train_$1_$2.err; echo "train $V rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$OUT/train_$1_$2.json'))
    r=d['roofline']; c=d['config']
    print('train $V: value %.2f M/s  kernel %.3f ms  kernel-only %.1f M/s  hbm frac %.3f  fwd %.1f plies (ref %.1f)  cpu %s' % (d['value']/1e6, r['avg_kernel_ms'], r['samples_per_sec_kernel_only']/1e6, r['frac'], c['mean_forwarded_plies'], c['mean_replayed_plies'], (d.get('cpu_baseline') or {}).get('value')))
except Exception as e:
    print('no line', e); print(open('$OUT/train_$1_$2.err').read()[-2000:])
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_train -o stats --output-format csv -- python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats_train.log 2>&1
python tools/summarize_prof.py $OUT train > $OUT/summary_train.txt 2>&1; head -8 $OUT/summary_train.txt
( time timeout 900 python bench.py --workload games > $OUT/games.json 2> $OUT/games.err ) 2> $OUT/games.time; echo "games rc=$?"; tail -3 $OUT/games.time
python - <<PY
import json
try:
    d=json.load(open('$OUT/games.json'))
    print('games/s', d['value'], d['moves_per_sec_by_phase'], d['game_length'], d['shortened_run']['games_per_sec'])
except Exception as e:
    print('no line', e); print(open('$OUT/games.err').read()[-3000:])
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
