#!/bin/bash
# Round 6: topk_wave_kernel with 2 / 4 / 8 waves per workgroup (the dispatcher starts ~3 500 waves at once and the rest at ~400 per us)
tag=r06q
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for rep in 1 2 3; do for m in wg2 wg4 wg8; do
    echo -n "$m: "; RECAD_TUNING_LIB=$PWD/ab_tune/librecad_hip_$m.so timeout 300 bash scripts/eval_session_trace.sh 2>&1 | grep "topk_wave_kernel\|gemm_f32_wide" | cut -c1-90 | tr '\n' ' '; echo
  done; done ) > $o/${tag}_topk_wg_ab.txt 2>&1; cat $o/${tag}_topk_wg_ab.txt
