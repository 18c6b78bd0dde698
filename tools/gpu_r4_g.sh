#!/bin/bash
# GPU visit r04g: k_mcts_expand sort A/B (blocked vs strided slots), search-only at more games per launch, MCTS tests.
TAG=${1:-r04g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cp elf_amd/lib/libelf_amd.so build/libelf_amd_default.so
SO="python bench.py --workload mcts --net random --features f16 --pregrow 0 --no-cpu-baseline"
for V in strided blocked strided blocked; do
  cp build/libelf_amd_$V.so elf_amd/lib/libelf_amd.so
  timeout 200 $SO --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('$V search-only', round(d['value']/1e6,2), 'select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4), 'depth', round(c['mean_depth'],2))"
done 2>&1 | tee $OUT/expand_ab.txt
cp build/libelf_amd_default.so elf_amd/lib/libelf_amd.so
{
for cfg in "1536 1 8192 2048 88 32" "2048 1 8192 2048 88 32" "2048 2 8192 2048 88 32" "1536 2 8192 2048 88 32"; do
  set -- $cfg
  timeout 300 $SO --games $1 --groups $2 --nodes-per-game $3 --rollouts $4 --warmup $5 --steps $6 2>$OUT/so_$1_$2.err | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('games $1 groups $2 rollouts $4: ', round(d['value']/1e6,2), 'M/s  ms/step', round(d['ms_per_step'],4), 'select', round(c['select_ms'],4), 'expand+backup', round(c['expand_backup_ms'],4), 'depth', round(c['mean_depth'],2))" || tail -3 $OUT/so_$1_$2.err
done
} 2>&1 | tee $OUT/search_only_games.txt
timeout 1200 python -m pytest tests/test_gpu_mcts.py -m gpu -q --timeout 300 --tb=short -rf -x > $OUT/pytest_mcts.log 2>&1; echo "mcts tests rc=$?"
tail -4 $OUT/pytest_mcts.log
