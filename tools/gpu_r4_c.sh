#!/bin/bash
# GPU visit r04c: feature kernel A/B (cooperative fp32 rows), playout A/B (scheduler, priorities), select phase attribution, tests of what changed.
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_board.py -m gpu -q --timeout 300 --tb=short -rf > $OUT/pytest_board_train.log 2>&1; echo "tests rc=$?"
tail -6 $OUT/pytest_board_train.log
{
for b in feat_old feat_base feat_now8 feat_nt; do
  for fmt in 0 1; do timeout 60 build/$b 16384 $fmt; done
done
for w in 512 1024 1536 4096 8192; do ELF_AMD_AGZ_WGS=$w timeout 60 build/feat_base 16384 0 | sed "s/^/wgs=$w /"; done
for r in 2048 65536; do timeout 60 build/feat_old $r 0; timeout 60 build/feat_base $r 0; done
} 2>&1 | tee $OUT/feat_ab.txt
{
for b in pl_base pl_maxilp pl_prio250 pl_prio350 pl_prio350_maxilp pl_base; do timeout 120 build/$b 4096 19; done
for b in pl_base pl_maxilp; do timeout 120 build/$b 65536 9; timeout 120 build/$b 16384 19; done
} 2>&1 | tee $OUT/playout_ab.txt
timeout 600 bash tools/select_phases.sh > $OUT/select_phases.txt 2>&1; echo "select phases rc=$?"; tail -12 $OUT/select_phases.txt
