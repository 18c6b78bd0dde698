#!/bin/bash
# GPU visit: SQ counters of the search kernels at 1024 and 2048 games per launch (what the second resident wave buys), and the two-ranks-on-one-GPU line at HEAD.
TAG=${1:-r04x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SO="python bench.py --workload mcts --net random --features f16 --nodes-per-game 8192 --rollouts 2048 --pregrow 0 --warmup 88 --steps 32 --no-cpu-baseline"
for G in 1024 2048; do
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_sq_mcts$G -o pmc --output-format csv -- $SO --games $G --groups 1 > $OUT/pmc_sq_mcts$G.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/pmc_sq_mcts$G/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    if 'k_mcts_select' in k or 'k_mcts_expand' in k:
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        print('games $G %-40s waves %.0f  wait/wave-cycles %.3f  issue/wave-cycles %.3f  VALU %.3g SALU %.3g per launch' % (k, m.get('SQ_WAVES', 0), m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_INSTS_VALU'], m['SQ_INSTS_SALU']))
PY
done 2>&1 | tee $OUT/search_occupancy_pmc.txt
find $OUT -name '*counter_collection.csv' -size +2M -delete; find $OUT -name '*.db' -delete
( time ELF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --workload both --games 64 --nodes-per-game 12288 --steps 20 --warmup 5 > $OUT/two_ranks_one_gpu.json 2> $OUT/two_ranks_one_gpu.err ) 2> $OUT/two_ranks.time; echo "two ranks rc=$?"; tail -3 $OUT/two_ranks.time
grep "^{" $OUT/two_ranks_one_gpu.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print('2 ranks: value', d['value'], 'per_rank', d['config']['per_rank_rollouts_per_sec'], 'games', d['selfplay_games']['value'], d['selfplay_games']['shortened_run']['games_per_sec'])"
