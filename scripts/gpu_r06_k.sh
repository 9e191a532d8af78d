#!/bin/bash
# Round 6: the LDS SpMM's first task static per wave (this tree) against popped from the queue (ab_old = the same tree with
# -DLDS_FIRST_STATIC=0): the step through bench.py and the launch with its stamps, alternating on one box.
tag=r06k
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( bash scripts/ab_step.sh . ab_old
  for rep in 1 2 3; do for d in . ab_old; do
    echo -n "$d: "; (cd $d && timeout 200 python3 scripts/spmm_lds_probe.py --lds-only 2>/dev/null | tail -1 | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); s=j['stamps_us']; print('%.2f us/launch' % j['lds_us'], 'stage %.2f gather %.2f rows %.2f span %.2f' % (s['stage']['mean'], s['gather']['mean'], s['rows']['mean'], s['span']), 'bit_reproducible', j['bit_reproducible'], 'rel_err', j['rel_err'])")
  done; done ) > $o/${tag}_first_task_ab.txt 2>&1; cat $o/${tag}_first_task_ab.txt
