#!/bin/bash
# GPU visit r04e: search-only with pipelined game groups, clock / perf-level check, the default bench line (timing of the whole run).
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SO="python bench.py --workload mcts --net random --features f16 --pregrow 0 --no-cpu-baseline"
rocm-smi --showclocks --showperflevel > $OUT/smi_idle.txt 2>&1
{
for cfg in "1024 1 8192 2048 88 32" "1024 2 8192 2048 88 32" "1024 4 8192 2048 88 32" "2048 2 4096 1024 44 32" "2048 4 4096 1024 44 32" "4096 4 2048 512 22 16"; do
  set -- $cfg
  timeout 300 $SO --games $1 --groups $2 --nodes-per-game $3 --rollouts $4 --warmup $5 --steps $6 2>$OUT/so_$1_$2.err | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('games $1 groups $2 rollouts $4: ', round(d['value']/1e6,2), 'M/s  ms/step', round(d['ms_per_step'],4), 'select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4), 'depth', round(c['mean_depth'],2))" || tail -3 $OUT/so_$1_$2.err
done
} 2>&1 | tee $OUT/search_only_groups.txt
( $SO --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 400 > /dev/null 2>&1 & sleep 12; rocm-smi --showclocks > $OUT/smi_busy_search.txt 2>&1; wait )
grep -E "sclk|mclk|fclk" $OUT/smi_busy_search.txt | head -5
rocm-smi --setperflevel high > $OUT/smi_set.txt 2>&1; tail -2 $OUT/smi_set.txt
{
timeout 300 $SO --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('perflevel high: search-only', round(d['value']/1e6,2), 'select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4))"
timeout 120 build/pl_base 4096 19
} 2>&1 | tee $OUT/perflevel_high.txt
rocm-smi --setperflevel auto > /dev/null 2>&1
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'])
for k in ('board_step','board_step_9x9','feature_extract','train_loader','boundary','selfplay_games','client_config'):
    v=d.get(k)
    if not v: print(k, None); continue
    if k=='feature_extract': print(k, v['f32']['avg_kernel_ms'], v['f32']['roofline']['frac'], v['f16']['avg_kernel_ms'], v['f16']['roofline']['frac'])
    else: print(k, v.get('value'), (v.get('roofline') or {}).get('frac'), v.get('parity_mismatches'))
print('cpu_baseline', d.get('cpu_baseline'))
PY
