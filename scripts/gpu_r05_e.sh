#!/bin/bash
# round 5: is the first 20-step call slower because the device's clocks are not up? (pre-warm / idle in front of each timed call)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( echo "== first calls, nothing in front"; PROBE_FIRST=1 timeout 200 python scripts/call_overhead_probe.py 20
  echo "== 20 ms of steps right in front of every timed call"; PROBE_FIRST=1 PROBE_PREWARM_MS=20 timeout 200 python scripts/call_overhead_probe.py 20
  echo "== 2 ms of steps in front"; PROBE_FIRST=1 PROBE_PREWARM_MS=2 timeout 200 python scripts/call_overhead_probe.py 20
  echo "== 20 ms of steps, then 5 ms idle"; PROBE_FIRST=1 PROBE_PREWARM_MS=20 PROBE_IDLE_MS=5 timeout 200 python scripts/call_overhead_probe.py 20
  echo "== 50 ms idle in front of every call"; PROBE_FIRST=1 PROBE_IDLE_MS=50 timeout 200 python scripts/call_overhead_probe.py 20
) 2>&1 | grep -v amdgpu.ids > $o/r05e_call_clock.txt
cat $o/r05e_call_clock.txt
