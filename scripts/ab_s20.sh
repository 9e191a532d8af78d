#!/bin/bash
# A/B of the driver-style invocation (bench.py --gpus 1 --steps 20 --warmup 5) between source trees on ONE box:
#   gpurun -- 'bash scripts/ab_s20.sh . ab_old'
for rep in 1 2 3; do
for d in "$@"; do
  (cd $d && timeout 120 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-topk --no-also 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$d', round(j['value']), '%.2f us/step' % (j['ms_per_step']*1e3), 'loss', j['last_step_loss'])")
done
done
