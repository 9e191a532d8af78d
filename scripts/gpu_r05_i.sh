#!/bin/bash
# round 5: the panel form with several targets (its NTG = 4 instantiation) against GEMM + selection
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for shape in "8192 34474 256" "16384 34474 64" "54617 34474 128"; do
    for rows in 16 32; do PROBE_TARGETS=4 PROBE_ROWS=$rows PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py $shape 3 2>/dev/null | grep "^panel" | sed "s/^panel /panel 4 targets rows=$rows /"; done
    PROBE_TARGETS=4 PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 3 2>/dev/null | grep -v amdgpu
  done ) > $o/r05i_score_targets.txt; cat $o/r05i_score_targets.txt
