#!/bin/bash
# round 5: whole-call replay as ONE graph (RK_NO_CHAIN=1, tuning build) against the chain of graphs, same box, alternating
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so
( for i in 1 2 3; do
    echo "== chain (default)"; timeout 200 python scripts/call_overhead_probe.py 20 2>&1 | grep "^steps"
    echo "== one graph (RK_NO_CHAIN=1)"; RK_NO_CHAIN=1 timeout 200 python scripts/call_overhead_probe.py 20 2>&1 | grep "^steps"
  done
  echo "== chain, 64 steps"; timeout 200 python scripts/call_overhead_probe.py 64 2>&1 | grep "^steps"
  echo "== one graph, 64 steps"; RK_NO_CHAIN=1 timeout 200 python scripts/call_overhead_probe.py 64 2>&1 | grep "^steps"
  echo "== chain, first calls behind 2 ms of steps"; PROBE_FIRST=1 PROBE_PREWARM_MS=2 timeout 200 python scripts/call_overhead_probe.py 20 2>&1 | grep "^steps\|^call [0-3]"
  echo "== one graph, first calls behind 2 ms of steps"; RK_NO_CHAIN=1 PROBE_FIRST=1 PROBE_PREWARM_MS=2 timeout 200 python scripts/call_overhead_probe.py 20 2>&1 | grep "^steps\|^call [0-3]"
) > $o/r05j_chain_ab.txt 2>&1; cat $o/r05j_chain_ab.txt
