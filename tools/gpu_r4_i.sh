#!/bin/bash
# GPU visit: compile-time epochs of the unordered_map order in k_mcts_expand: MCTS tests, search-only lines
TAG=${1:-r04i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_train.py -m gpu -q --timeout 300 --tb=short -rf -x > $OUT/pytest_mcts.log 2>&1; echo "mcts tests rc=$?"
tail -4 $OUT/pytest_mcts.log
SO="python bench.py --workload mcts --net random --features f16 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
for cfg in "1024 1" "1024 1" "2048 2"; do
  set -- $cfg
  timeout 300 $SO --games $1 --groups $2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('games $1 groups $2: ', round(d['value']/1e6,2), 'M/s  select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4))"
done 2>&1 | tee $OUT/search_only.txt
