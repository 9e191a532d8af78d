#!/bin/bash
# round 5: L2 warm-up of the next panel's item rows in the panel form: parity, A/B on one box (tuning build, RK_PAN_NO_WARM=1 = without)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "score_topk or panel or stress or eval" 2>&1 | tail -2
export RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so
( for i in 1 2; do for shape in "8192 34474 256" "54617 34474 128 5" "16384 34474 64" "16384 131072 64 3"; do
    echo "== warm-up (default)"; PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep "^panel"
    echo "== RK_PAN_NO_WARM=1"; RK_PAN_NO_WARM=1 PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep "^panel"
  done; done
  echo "== stamps, warm-up"; python3 scripts/pan_stamps.py 8192 34474 256 2>&1 | grep "lifetime\|panel 2"
  echo "== stamps, RK_PAN_NO_WARM=1"; RK_PAN_NO_WARM=1 python3 scripts/pan_stamps.py 8192 34474 256 2>&1 | grep "lifetime\|panel 2"
) > $o/r05m_pan_warm_ab.txt 2>&1; cat $o/r05m_pan_warm_ab.txt
