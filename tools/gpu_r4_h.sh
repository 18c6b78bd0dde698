#!/bin/bash
# GPU visit r04h: one-wave feature kernel per row format (7 / 8 waves per SIMD) vs the run-time-format kernel; k_mcts_expand phases.
TAG=${1:-r04h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
for rep in 1 2; do
for b in feat_old feat_tpl feat_tpl8; do
  for fmt in 0 1; do timeout 60 build/$b 16384 $fmt; done
done
done
for b in feat_old feat_tpl feat_tpl8; do timeout 60 build/$b 65536 0; timeout 60 build/$b 2048 0; done
} 2>&1 | tee $OUT/feat_ab.txt
timeout 400 bash tools/expand_phases.sh > $OUT/expand_phases.txt 2>&1; tail -14 $OUT/expand_phases.txt
