#!/bin/bash
# Short GPU visit: parity tests, smoke, the train workload, the default bench line.  Usage: bash tools/gpu_verify.sh <tag>
TAG=${1:-v}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
echo "== bench train"
timeout 300 python bench.py --workload train > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "rc=$?"
cat $OUT/bench_train.json; tail -3 $OUT/bench_train.err
echo "== features / epilogue microbench"
timeout 200 python tools/net_bench2.py 2048 > $OUT/net_bench2.log 2>&1; echo "rc=$?"
grep -v Warning $OUT/net_bench2.log | tail -12
echo "== bench default"
timeout 700 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
