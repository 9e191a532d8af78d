#!/bin/bash
# A/B of the LightGCN step between the current tree and variant trees built in-tree (ab_*/), same box
for rep in 1 2; do
for d in "$@"; do
  (cd $d && python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$d', round(j['value']), '%.2f us/step' % (j['ms_per_step']*1e3), 'spmm %.2f us' % j['roofline']['avg_launch_us'])")
done
done
