#!/bin/bash
# A/B of whole-library compiler flag variants (prebuilt build/libelf_amd_<v>.so): board, search-only, train, feature lines
for V in "$@"; do
  cp build/libelf_amd_$V.so elf_amd/lib/libelf_amd.so
  python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$V board', round(d['value']/1e6,1), d.get('parity_mismatches'))"
  python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$V board9', round(d['value']/1e6,1), d.get('parity_mismatches'))"
  python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('$V search', round(d['value']/1e6,2), round(c['select_ms'],4), round(c['expand_backup_ms'],4))"
  python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$V train', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],3))"
  python bench.py --workload feature --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$V feat', round(d['f32']['avg_kernel_ms'],4), round(d['f16']['avg_kernel_ms'],4))"
done
