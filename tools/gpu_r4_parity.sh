#!/bin/bash
# third real-net parity measurement, on the final round-4 kernels (other seeds): tests/real_net_parity.py; then the default bench line
TAG=${1:-r04w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/real_net_parity.py --games 8 --moves 8 --rollouts 512 --seed 2468 --out $OUT/parity_512.json > $OUT/parity_512.log 2>&1; echo "rc=$?"
cat $OUT/parity_512.json; echo
timeout 900 python tests/real_net_parity.py --games 3 --moves 2 --rollouts 8192 --seed 1357 --out $OUT/parity_8192.json > $OUT/parity_8192.log 2>&1; echo "rc=$?"
cat $OUT/parity_8192.json; echo
bash tools/gpu_r4_bench.sh r04z
