#!/bin/bash
# Round 6: A/B of the selection kernel's target counters on ONE box, alternating: per-lane add-with-carry (cm0), the equal count on
# the scalar side (cm1), both on the scalar side (cm2); rocprofv3 kernel durations through the evaluation session.
tag=r06h
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for rep in 1 2 3; do for m in 0 1 2; do
    echo -n "cm$m: "; RECAD_TUNING_LIB=$PWD/ab_tune/librecad_hip_cm$m.so timeout 300 bash scripts/eval_session_trace.sh 2>&1 | grep "topk_wave_kernel\|gemm_f32_wide" | tr '\n' ' '; echo
  done; done ) > $o/${tag}_topk_count_ab.txt 2>&1; cat $o/${tag}_topk_count_ab.txt
