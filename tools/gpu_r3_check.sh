#!/bin/bash
# GPU visit: the changed tests first (short timeouts), then the train / headline / board sanity numbers.  Usage: bash tools/gpu_r3_check.sh <tag>
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pybind_boundary.py tests/test_gpu_train.py tests/test_gpu_gtp.py -m gpu -q --timeout 120 --tb=short -rf > $OUT/pytest_changed.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_changed.log
tail -25 $OUT/pytest_changed.log
for KB in 1 4 8; do
  timeout 200 python bench.py --workload train --train-prefetch $KB --steps 20 --warmup 3 --no-cpu-baseline > $OUT/train_kb$KB.json 2> $OUT/train_kb$KB.err
  python -c "import json;d=json.load(open('$OUT/train_kb$KB.json'));print('train prefetch $KB', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['mean_replayed_plies'])"
done
timeout 200 python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline > $OUT/board.json 2> $OUT/board.err
python -c "import json;d=json.load(open('$OUT/board.json'));print('board', d['value'], d['roofline']['avg_kernel_ms'], d.get('parity_mismatches'))"
timeout 400 python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline > $OUT/mcts.json 2> $OUT/mcts.err
python -c "import json;d=json.load(open('$OUT/mcts.json'));c=d['config'];print('mcts', d['value'], d['ms_per_step'], c['select_ms'], c['expand_backup_ms'], c['mean_depth'], c['move_boundary_ms'])"
timeout 300 python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline > $OUT/search_only.json 2> $OUT/search_only.err
python -c "import json;d=json.load(open('$OUT/search_only.json'));c=d['config'];print('search-only', d['value'], d['ms_per_step'], c['select_ms'], c['expand_backup_ms'], c['mean_depth'])"
timeout 300 python bench.py --workload games --no-cpu-baseline > $OUT/games.json 2> $OUT/games.err
python -c "import json;d=json.load(open('$OUT/games.json'));print('games', d['value'], d['games_finished'], d['seconds'])"
