#!/bin/bash
# GPU-box helper for board-engine iteration: parity tests, phase attribution, board benches.  Usage: bash tools/gpu_board.sh [tag] [full]
TAG=${1:-board}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" = "full" ]; then SEL="tests"; else SEL="tests/test_gpu_board.py tests/test_gpu_train.py"; fi
timeout 900 python -m pytest $SEL -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/playout_phases.hip -o /tmp/pp 2>/dev/null && /tmp/pp 4096 | tail -11 | tee $OUT/phases.txt
for b in 4096 16384; do
  python bench.py --workload board --boards $b --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_board_$b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('board', d['config']['boards_per_gpu'], d['value'], d['roofline']['avg_kernel_ms'], d.get('parity_mismatches'))"
done
python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_board9.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('board9', d['config']['boards_per_gpu'], d['value'], d['roofline']['avg_kernel_ms'], d.get('parity_mismatches'))"
python bench.py --workload feature --steps 20 --warmup 3 2>/dev/null | tee $OUT/bench_feature.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); f=d.get('feature_extract',d); print('feature f32', f['f32']['avg_kernel_ms'], f['f32']['roofline']['achieved'], 'f16', f['f16']['avg_kernel_ms'], f['f16']['roofline']['achieved'])"
