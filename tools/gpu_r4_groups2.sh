#!/bin/bash
# GPU visit: the headline with one net stream per game group (concurrent net calls), over the number of groups
TAG=${1:-r04y2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "256 2 2" "256 4 4" "256 8 8" "256 4 2" "512 8 8"; do
  set -- $cfg
  timeout 400 python bench.py --workload mcts --games $1 --groups $2 --net-streams $3 --steps 20 --warmup 5 --no-cpu-baseline --no-sub 2>$OUT/h_$1_$2_$3.err | python -c "import json,sys;d=json.loads(sys.stdin.read());c=d['config'];print('games $1 groups $2 net streams $3 (rows/call %d): ' % ($1//$2*16), round(d['value']), 'rollouts/s  ms/step', round(d['ms_per_step'],2), 'net call ms', round(d['net_roofline']['avg_call_ms'],2))" || tail -3 $OUT/h_$1_$2_$3.err
done 2>&1 | tee $OUT/headline_net_streams.txt
