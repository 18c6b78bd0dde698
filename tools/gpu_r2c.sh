#!/bin/bash
# round 2, visit C: the reworked bench line + rocprof of the headline run
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"
tail -3 $OUT/bench.time; tail -5 $OUT/bench.err; cat $OUT/bench.json | head -c 6000; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_mctsnet.log 2>&1
python tools/summarize_prof.py $OUT mctsnet > $OUT/summary_mctsnet.txt 2>&1; head -20 $OUT/summary_mctsnet.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_board9 -o pmc --output-format csv -- python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_board9.log 2>&1
python tools/summarize_prof.py $OUT board9 > $OUT/summary_board9.txt 2>&1; grep k_playout $OUT/summary_board9.txt
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
