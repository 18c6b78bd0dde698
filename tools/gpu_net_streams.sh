#!/bin/bash
# GPU-box experiment: the headline with one net stream vs one per group (and 3 / 4 groups)
for cfg in "2 1" "2 2" "4 4" "3 3"; do
  set -- $cfg
  python bench.py --workload mcts --steps 20 --warmup 5 --no-cpu-baseline --groups $1 --net-streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']; n=d.get('net_roofline',{})
print('groups $1 net-streams $2: %.0f rollouts/s  %.2f ms/step  net call %.2f ms  games %d' % (d['value'], d['ms_per_step'], n.get('avg_call_ms',0), c['games_per_gpu']))"
done
