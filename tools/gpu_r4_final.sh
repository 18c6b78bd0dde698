#!/bin/bash
# GPU visit: the whole -m gpu suite, smoke, the search-only lines (1024 games as in round 3; 2048 games in two pipelined groups) and the default bench line.
TAG=${1:-r04z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --tb=short -rf > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -2 $OUT/smoke.log
SO="python bench.py --workload mcts --net random --features f16 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
timeout 300 $SO --games 1024 --groups 1 > $OUT/search_only.json 2> $OUT/search_only.err
timeout 300 $SO --games 2048 --groups 2 > $OUT/search_only_2048.json 2> $OUT/search_only_2048.err
for f in search_only search_only_2048; do python -c "import json;d=json.load(open('$OUT/$f.json'));c=d['config'];print('$f', d['value'], d['ms_per_step'], c['select_ms'], c['expand_backup_ms'], c['mean_depth'])"; done
( time timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['move_boundary_ms'])
for k in ('board_step','board_step_9x9','feature_extract','train_loader','selfplay_games','client_config'):
    v=d.get(k)
    if not v: print(k, None); continue
    if k=='feature_extract': print(k, v['f32']['avg_kernel_ms'], v['f32']['roofline']['frac'], v['f16']['avg_kernel_ms'], v['f16']['roofline']['frac'])
    else:
        r=v.get('roofline') or {}
        print(k, v.get('value'), 'frac', r.get('frac'), 'salu', r.get('salu_issue_frac'), 'weighted', (r.get('valu_class_weighted') or {}).get('frac'), 'binding', r.get('binding_issue_roof'), r.get('pmc_source_match'), v.get('parity_mismatches'))
print('train roofline issue', (d['train_loader']['roofline'].get('issue') or {}).get('frac'), (d['train_loader']['roofline'].get('issue') or {}).get('salu_issue_frac'))
print('cpu_baseline', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('sample'))
PY
