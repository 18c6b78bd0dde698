#!/bin/bash
# Quick iteration visit: board + mcts + train parity, phase attribution, board bench. Usage: bash tools/gpu_quick.sh
python -m pytest tests/test_gpu_board.py tests/test_gpu_mcts.py tests/test_gpu_train.py -q -x 2>&1 | tail -4
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/playout_phases.hip -o /tmp/pp 2>/dev/null && /tmp/pp 4096 | tail -11
for b in 4096 16384; do
  python bench.py --workload board --boards $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('board', d['config']['boards_per_gpu'], d['value'], d['roofline']['avg_kernel_ms'])"
done
python bench.py --workload train --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['value'], d['roofline']['avg_kernel_ms'], d['config']['replayed_board_steps_per_sec'])"
python bench.py --workload mcts --net random --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('search-only', d['value'], d['config']['select_ms'], d['config']['expand_backup_ms'])"
