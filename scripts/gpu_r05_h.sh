#!/bin/bash
# round 5: panel form with the user operand read one MFMA pair ahead: probe at the four shapes (parity ran in the call before)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for shape in "5893 3702 64" "16384 34474 64" "54617 34474 128 5" "8192 34474 256" "8192 34474 256"; do
    PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done
  PROBE_ROWS=16 PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py 8192 34474 256 2>/dev/null | grep "^panel" | sed 's/^panel /panel (16-row workgroups) /' ) > $o/r05h_score_probe.txt; cat $o/r05h_score_probe.txt
