#!/bin/bash
# Round 6: does the selection kernel read the score matrix faster when the GEMM's stores are cacheable (not nt)?  One box, alternating.
tag=r06j
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for rep in 1 2 3; do for m in nt nont; do
    echo -n "$m: "; RECAD_TUNING_LIB=$PWD/ab_tune/librecad_hip_$m.so timeout 300 bash scripts/eval_session_trace.sh 2>&1 | grep "topk_wave_kernel\|gemm_f32_wide\|per evaluation" | cut -c1-100 | tr '\n' ' '; echo
  done; done ) > $o/${tag}_score_store_ab.txt 2>&1; cat $o/${tag}_score_store_ab.txt
