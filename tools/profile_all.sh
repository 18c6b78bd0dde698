#!/bin/bash
# ONE script regenerates every measured artefact under profiles/ for a round.  Run on a GPU box from the repo root:
#     gpurun --timeout 2400 -- 'bash tools/profile_all.sh r06p [tests] [bench] [stats] [pmc] [headline]'
# (no step names = all steps).  Everything lands in gpurun_out/<tag>/ (scratch, merged back by gpurun); afterwards, in the build container:
#     python tools/update_issue.py gpurun_out/<tag> <tag>; python tools/update_traffic.py gpurun_out/<tag> <tag>     (PMC -> profiles/pmc_*.json)
#     cp gpurun_out/<tag>/{pytest_gpu.txt,bench_line.json,bench_full.json,summary_*.txt} profiles/ with the <tag>_ prefix
# Steps
#   tests     python -m pytest tests -m gpu                                          -> pytest_gpu.txt
#   bench     the default `python bench.py` (the driver's command)                   -> bench_line.json, bench_full.json
#   stats     rocprofv3 --kernel-trace --stats of board / board9 / search-only / train / features  -> summary_<w>.txt
#   pmc       separate --pmc passes (FETCH_SIZE, WRITE_SIZE, LDS, SQ issue) of the same commands   -> pmc_*/ (never combined with a trace)
#   chase     tools/chase.hip: ns per dependent load by footprint and number of chasing waves     -> chase.txt, chase.json
#   headline  rocprofv3 --kernel-trace --stats of the headline with a WARM MIOpen user database: the net is run once un-profiled first,
#             so the summary shows the CK implicit-GEMM convolutions, not MIOpen's find / verification kernels (naive_conv_*)
TAG=${1:-r06}
shift
STEPS="${*:-tests bench stats pmc headline chase}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=/tmp/miopen_udb MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_udb
mkdir -p /tmp/miopen_udb
python -c "import sys; sys.path.insert(0,'.'); from elf_amd._lib import kernel_source_hash; print(kernel_source_hash())" > $OUT/kernel_source_hash.txt
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }

SO="python bench.py --workload mcts --net random --features f16 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline --no-sub"
PROF_board="python bench.py --workload board --steps 5 --warmup 1 --no-cpu-baseline"
PROF_board9="python bench.py --workload board --board-size 9 --boards 65536 --steps 3 --warmup 1 --no-cpu-baseline"
PROF_mcts="$SO --games 4096 --groups 2"
PROF_train="python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline"
PROF_feat32="python bench.py --workload feature --steps 20 --warmup 3 --feature-formats f32"
PROF_feat16="python bench.py --workload feature --steps 20 --warmup 3 --feature-formats f16"
HEADLINE="python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline --no-sub"

if has tests; then
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q --tb=short -rf 2>&1 | grep -v "^  File\|^Extension" | tail -40 > $OUT/pytest_gpu.txt
  tail -3 $OUT/pytest_gpu.txt
  timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2 >> $OUT/pytest_gpu.txt
fi
if has bench; then
  ( time ELF_BENCH_FULL=$OUT/bench_full.json timeout 1500 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err ) 2> $OUT/bench.time
  tail -3 $OUT/bench.time; wc -c $OUT/bench_line.json
fi
if has stats || has pmc; then
  for W in ${ELF_PROF_W:-board board9 mcts train feat32 feat16}; do     # ELF_PROF_W="mcts": one workload only
    eval CMD=\$PROF_$W
    if has stats; then
      timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_$W -o stats --output-format csv -- $CMD > $OUT/stats_$W.log 2>&1
    fi
    if has pmc; then   # counters in runs of their own: never together with a trace
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch_$W.log 2>&1
      timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_write_$W.log 2>&1
      if [ $W != feat32 ] && [ $W != feat16 ]; then
        timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lds_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_lds_$W.log 2>&1
        timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$W -o pmc --output-format csv -- $CMD > $OUT/pmc_sq_$W.log 2>&1
      fi
    fi
    python tools/summarize_prof.py $OUT $W > $OUT/summary_$W.txt 2>&1
    grep -E "k_playout|k_mcts|k_replay|k_extract" $OUT/summary_$W.txt | head -8
  done
fi
if has chase; then
  # dependent-load latency per level of a descent (tools/chase.hip) -> chase.json (bench.py: roofline.latency_chain_ms)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/chase.hip -o /tmp/chase 2> $OUT/chase_build.log && timeout 600 /tmp/chase > $OUT/chase.txt 2>&1
  cp gpurun_out/chase.json $OUT/chase.json 2>/dev/null; tail -12 $OUT/chase.txt
fi
if has headline; then
  # warm the per-box MIOpen user database first (un-profiled): the find / verification kernels then stay out of the profiled run
  timeout 600 $HEADLINE > $OUT/headline_warm.json 2> $OUT/headline_warm.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_mctsnet -o stats --output-format csv -- $HEADLINE > $OUT/stats_mctsnet.log 2>&1
  python tools/summarize_prof.py $OUT mctsnet > $OUT/summary_mctsnet.txt 2>&1
  # the same trace split into the tree-growing prologue (random replies) and the steps that follow a net call (every row of the
  # random-init fp16 net's reply carries prior ties: k_mcts_expand runs its exact std::sort replay there)
  python tools/headline_kernel_durations.py $OUT/stats_mctsnet >> $OUT/summary_mctsnet.txt 2>&1
  head -14 $OUT/summary_mctsnet.txt; tail -10 $OUT/summary_mctsnet.txt
  grep -c naive_conv $OUT/summary_mctsnet.txt
fi
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +2M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
