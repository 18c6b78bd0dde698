#!/bin/bash
# GPU visit: parity tests (incl. new feature formats + net glue), net execution-path comparison, bench with each path.
TAG=${1:-r01e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
echo "== net paths"
timeout 600 python tools/net_bench2.py 2048 4096 > $OUT/net_bench2.log 2>&1; echo "rc=$?"
grep -v Warning $OUT/net_bench2.log | tail -30
for impl in fused miopen eager; do
  echo "== bench mcts net-impl $impl"
  timeout 600 python bench.py --workload mcts --net-impl $impl --steps 16 --warmup 6 --no-cpu-baseline > $OUT/bench_$impl.json 2> $OUT/bench_$impl.err; echo "rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_$impl.json'));print(d['value'], d['ms_per_step'], d['config']['search_ms_per_step'])"
done
