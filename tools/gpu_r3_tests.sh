#!/bin/bash
# GPU visit: the whole -m gpu suite (+ smoke), logs under gpurun_out/<tag>.  Usage: bash tools/gpu_r3_tests.sh <tag> [pytest args]
TAG=${1:-r03t}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 150 --tb=short -rf "$@" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
