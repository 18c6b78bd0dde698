#!/bin/bash
# round 2, visit A: parity of the rewritten search kernels + quick benches
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_compat.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
NULLNET="python bench.py --workload mcts --net random --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --warmup 88 --steps 32 --no-cpu-baseline"
timeout 300 $NULLNET > $OUT/bench_nullnet.json 2> $OUT/bench_nullnet.err; echo "nullnet rc=$?"
cat $OUT/bench_nullnet.json; tail -3 $OUT/bench_nullnet.err
timeout 300 $NULLNET --wait-rows 1 > $OUT/bench_nullnet_wait.json 2> $OUT/bench_nullnet_wait.err; echo "nullnet(wait) rc=$?"
cat $OUT/bench_nullnet_wait.json
timeout 600 python bench.py --workload mcts --no-cpu-baseline --steps 20 --warmup 6 > $OUT/bench_mcts.json 2> $OUT/bench_mcts.err; echo "mcts rc=$?"
cat $OUT/bench_mcts.json; tail -3 $OUT/bench_mcts.err
