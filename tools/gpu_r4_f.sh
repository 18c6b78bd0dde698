#!/bin/bash
# GPU visit r04f: k_mcts_select A/B (deferred stores, slot prefetch, root cache), MCTS tests on the chosen build.
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cp elf_amd/lib/libelf_amd.so build/libelf_amd_default.so
NULLNET="python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
for V in none all fspre d4pre rootlds none all; do
  cp build/libelf_amd_$V.so elf_amd/lib/libelf_amd.so
  timeout 200 $NULLNET 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('$V search-only', round(d['value']/1e6,2), 'select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4), 'depth', round(c['mean_depth'],2))"
done 2>&1 | tee $OUT/select_ab.txt
cp build/libelf_amd_default.so elf_amd/lib/libelf_amd.so
timeout 1200 python -m pytest tests/test_gpu_mcts.py -m gpu -q --timeout 300 --tb=short -rf -x > $OUT/pytest_mcts.log 2>&1; echo "mcts tests rc=$?"
tail -6 $OUT/pytest_mcts.log
timeout 300 bash tools/select_phases.sh > $OUT/select_phases.txt 2>&1; tail -10 $OUT/select_phases.txt
