#!/bin/bash
# rocprofv3 kernel stats of the final state: headline, search-only, feature extraction.  Usage: bash tools/gpu_stats.sh <tag>
TAG=${1:-stats}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
NULLNET="python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_mcts -o stats --output-format csv -- $NULLNET > $OUT/stats_mcts.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_feat -o stats --output-format csv -- python bench.py --workload feature --steps 20 --warmup 3 > $OUT/stats_feat.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_mctsnet.log 2>&1
for W in mcts feat mctsnet; do python tools/summarize_prof.py $OUT $W > $OUT/summary_$W.txt 2>&1; head -9 $OUT/summary_$W.txt | cut -c1-140; done
find $OUT -name '*kernel_trace.csv' -size +4M -delete
du -sh $OUT
