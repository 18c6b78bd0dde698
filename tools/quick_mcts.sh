#!/bin/bash
# GPU-box helper: MCTS parity tests + search-only bench (random net) for kernel iteration
python -m pytest tests/test_gpu_mcts.py tests/test_compat.py -m gpu -x -q 2>&1 | tail -3
python bench.py --workload mcts --net random --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('rollouts/s %.0f  ms/step %.3f  select %.3f  expand+backup %.3f  depth %.2f' % (d['value'], d['ms_per_step'], c['select_ms'], c['expand_backup_ms'], d['roofline']['mean_depth']))"
