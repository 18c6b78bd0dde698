#!/bin/bash
OUT=gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/playout_phases.hip -o /tmp/pp 2>/dev/null && /tmp/pp 4096 | tail -11
for b in 4096 16384; do
  python bench.py --workload board --boards $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('board', d['config']['boards_per_gpu'], d['value'], d['roofline']['avg_kernel_ms'], d.get('parity_mismatches'))"
done
python bench.py --workload mcts --net random --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('search-only', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'])"
python bench.py --workload mcts --net random --games 256 --groups 2 --rollouts 8192 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline-shape search-only', d['value'], d['ms_per_step'], d['config']['select_ms'], d['config']['expand_backup_ms'], d['config']['mean_depth'], d['config']['move_boundary_ms'], d['config']['move_boundary_queue_drain_ms'])"
