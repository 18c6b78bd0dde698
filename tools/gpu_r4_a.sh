#!/bin/bash
# GPU visit r04a: issue-roof probe, the 2-rank path on one GPU (test + a line with the real net), the whole -m gpu suite.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 build/issue_probe > $OUT/issue_probe.json 2> $OUT/issue_probe.err; echo "issue_probe rc=$?"
python - <<PY
import json
d=json.load(open('$OUT/issue_probe.json'))
for r in d['rows']:
    print('%-44s w=%d  cyc/inst/wave %.2f  salu/clk/cu %.3f  valu/clk/simd %.3f' % (r['kind'][:44], r['waves_per_simd'], r['cycles_per_instruction_per_wave'], r['salu_per_clk_per_cu'], r['valu_per_clk_per_simd']))
PY
timeout 900 python -m pytest tests/test_dist_gloo.py -m gpu -q --timeout 850 --tb=short -rf > $OUT/pytest_two_ranks.log 2>&1; echo "two-ranks test rc=$?"
tail -5 $OUT/pytest_two_ranks.log
# the headline configuration with two ranks on this one GPU (64 games per rank, smaller node pools): exercises run_mcts / run_games at world 2
( time ELF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --workload both --games 64 --nodes-per-game 12288 --steps 20 --warmup 5 > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err ) 2> $OUT/two_ranks.time; echo "two ranks rc=$?"
tail -3 $OUT/two_ranks.time
python - <<PY
import json
try:
    d=json.load(open('$OUT/two_ranks_one_gpu.json'))
    print('2 ranks: value', d['value'], 'ms/step', d['ms_per_step'], 'per_rank', d['config']['per_rank_rollouts_per_sec'], 'games', d.get('selfplay_games',{}).get('value'))
except Exception as e:
    print('no line', e); print(open('$OUT/two_ranks_one_gpu.err').read()[-3000:])
PY
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 --tb=short -rf -k "not two_ranks" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
