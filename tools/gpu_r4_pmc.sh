#!/bin/bash
# GPU visit: rocprofv3 kernel stats + PMC passes of the final kernels (board 19, board 9, search-only MCTS, train, features) and the
# kernel stats of the headline.  Usage (from the repo root, via gpurun): bash tools/gpu_r4_pmc.sh <tag>
# Afterwards, here:  python tools/update_issue.py gpurun_out/<tag> <tag>; python tools/update_traffic.py gpurun_out/<tag> <tag>
TAG=${1:-r04p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0,'.'); from elf_amd._lib import kernel_source_hash; print(kernel_source_hash())" > $OUT/kernel_source_hash.txt
NULLNET="python bench.py --workload mcts --net random --features f16 --games 1024 --groups 1 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
PROF_board="python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline"
PROF_board9="python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline"
PROF_mcts="$NULLNET"
PROF_train="python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline"
PROF_feat32="python bench.py --workload feature --steps 20 --warmup 3 --feature-formats f32"
PROF_feat16="python bench.py --workload feature --steps 20 --warmup 3 --feature-formats f16"
for W in board board9 mcts train feat32 feat16; do
  eval CMD=\$PROF_$W
  echo "== rocprofv3 stats $W"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o stats --output-format csv -- $CMD > $OUT/stats_$W.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch_$W.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_write_$W.log 2>&1
  if [ $W != feat32 ] && [ $W != feat16 ]; then
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_lds_$W.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_sq_$W.log 2>&1
  fi
  python tools/summarize_prof.py $OUT $W > $OUT/summary_$W.txt 2>&1
  grep -E "k_playout|k_mcts|k_replay|k_extract" $OUT/summary_$W.txt | head -12
done
echo "== rocprofv3 stats, search-only with 2048 games in two pipelined groups (215 GB of node records)"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_mcts2048 -o stats --output-format csv -- python bench.py --workload mcts --net random --features f16 --games 2048 --groups 2 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline > $OUT/stats_mcts2048.log 2>&1
python tools/summarize_prof.py $OUT mcts2048 > $OUT/summary_mcts2048.txt 2>&1
head -8 $OUT/summary_mcts2048.txt
echo "== rocprofv3 stats, headline (with the real net)"
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats_mctsnet.log 2>&1
python tools/summarize_prof.py $OUT mctsnet > $OUT/summary_mctsnet.txt 2>&1
head -12 $OUT/summary_mctsnet.txt
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +2M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
