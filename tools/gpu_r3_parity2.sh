#!/bin/bash
# second real-net parity measurement (other seeds): tests/real_net_parity.py, see tools/gpu_r3_parity.sh
TAG=${1:-r03m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tests/real_net_parity.py --games 8 --moves 8 --rollouts 512 --seed 777 --out $OUT/parity_512.json > $OUT/parity_512.log 2>&1; echo "rc=$?"
cat $OUT/parity_512.json
timeout 900 python tests/real_net_parity.py --games 3 --moves 2 --rollouts 8192 --seed 999 --out $OUT/parity_8192.json > $OUT/parity_8192.log 2>&1; echo "rc=$?"
cat $OUT/parity_8192.json
