#!/bin/bash
# GPU visit: SURVEY 8(d) config-3 parity artefact with the real 20x256 fp32 net against the real reference stack.
# Usage (from the repo root, via gpurun): bash tools/gpu_r3_parity.sh <tag>
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
echo "== real-net parity: 8 games x 8 moves x 512 rollouts"
timeout 1500 python tests/real_net_parity.py --games 8 --moves 8 --rollouts 512 --out $OUT/parity_512.json > $OUT/parity_512.log 2>&1; echo "rc=$?"
cat $OUT/parity_512.json
echo "== real-net parity: 2 games x 2 moves x 8192 rollouts"
timeout 1500 python tests/real_net_parity.py --games 2 --moves 2 --rollouts 8192 --seed 4321 --out $OUT/parity_8192.json > $OUT/parity_8192.log 2>&1; echo "rc=$?"
cat $OUT/parity_8192.json
