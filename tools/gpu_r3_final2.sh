#!/bin/bash
# GPU visit: the whole -m gpu suite, smoke and the default bench line.  Usage: bash tools/gpu_r3_final2.sh <tag>
TAG=${1:-r03p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout 200 --tb=short -rf > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 120 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
( time timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['move_boundary_ms'])
for k in ('board_step','board_step_9x9','feature_extract','train_loader','boundary','selfplay_games','client_config'):
    v=d.get(k)
    if not v: print(k, None); continue
    if k=='feature_extract': print(k, v['f32']['avg_kernel_ms'], v['f32']['roofline']['frac'], v['f16']['avg_kernel_ms'], v['f16']['roofline']['frac'])
    else: print(k, v.get('value'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('pmc_source_match'), v.get('parity_mismatches'))
print('cpu_baseline', d.get('cpu_baseline'))
PY
