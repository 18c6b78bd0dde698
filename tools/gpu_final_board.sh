TAG=r01k; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 120 python bench.py --workload board --board-size 9 --boards 65536 --steps 10 --warmup 2 > $OUT/bench_board9.json 2>/dev/null
CMD="python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline"
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/stats_board -o stats --output-format csv -- $CMD > $OUT/stats_board.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_board -o pmc --output-format csv -- $CMD > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_board -o pmc --output-format csv -- $CMD > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds_board -o pmc --output-format csv -- $CMD > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_board -o pmc --output-format csv -- $CMD > /dev/null 2>&1
python tools/summarize_prof.py $OUT board > $OUT/summary_board.txt 2>&1
find $OUT -name '*kernel_trace.csv' -size +4M -delete
grep "k_playout" $OUT/summary_board.txt | head -3
