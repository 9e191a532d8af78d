#!/bin/bash
# A/B of the LightGCN step between the current tree (.) and variant trees built in-tree (ab_*/, git-ignored), same box:
#   gpurun -- 'bash scripts/ab_step.sh . ab_old'
for rep in 1 2 3; do
for d in "$@"; do
  (cd $d && timeout 120 python3 bench.py --no-cpu-baseline --no-parity --no-topk --no-also 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$d', round(j['value']), '%.2f us/step' % (j['ms_per_step']*1e3), 'spmm %.2f us' % j['roofline']['avg_launch_us'], 'loss', j['last_step_loss'])")
done
done
