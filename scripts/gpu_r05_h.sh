#!/bin/bash
# round 5: panel form (user operand one MFMA pair ahead, transposed seen bitmap): bit-exact paths, probe at the four shapes, stamps
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "score_topk or panel or stress or eval" 2>&1 | tail -3
( for shape in "5893 3702 64" "16384 34474 64" "54617 34474 128 5" "8192 34474 256" "8192 34474 256"; do
    PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done
  PROBE_ROWS=16 PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py 8192 34474 256 2>/dev/null | grep "^panel" | sed 's/^panel /panel (16-row workgroups) /' ) > $o/r05h_score_probe.txt; cat $o/r05h_score_probe.txt
export RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so
(python3 scripts/pan_stamps.py 8192 34474 256; python3 scripts/pan_stamps.py 54617 34474 128) 2>&1 | grep -v amdgpu.ids > $o/r05h_pan_stamps.txt; grep "pass 1\|collect\|lifetime" $o/r05h_pan_stamps.txt
