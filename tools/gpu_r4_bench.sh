#!/bin/bash
# the default bench line alone
TAG=${1:-r04z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time; tail -5 $OUT/bench.err
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'])
so=d.get('search_only'); print('search_only', so if isinstance(so,str) else (so['value'], so['ms_per_step'], so['select_ms'], so['expand_backup_ms'], so['mean_depth'], so['roofline']['frac']))
for k in ('board_step','board_step_9x9','train_loader','selfplay_games','client_config'):
    v=d.get(k); r=(v or {}).get('roofline') or {}
    print(k, (v or {}).get('value'), r.get('frac'), r.get('salu_issue_frac'), (r.get('valu_class_weighted') or {}).get('frac'), r.get('pmc_source_match'), (v or {}).get('parity_mismatches'))
f=d['feature_extract']; print('feat', f['f32']['avg_kernel_ms'], f['f16']['avg_kernel_ms'])
PY
