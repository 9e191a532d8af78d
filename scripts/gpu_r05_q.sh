#!/bin/bash
# round 5: staggered start of the co-resident workgroups of the score-matrix GEMM, now that a tile's stores are not drained (tuning build)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so
( for ns in 0 2000 3500 5000 7000 0; do echo "== RK_GEMM_STAGGER_NS=$ns"; RK_GEMM_STAGGER_NS=$ns timeout 200 bash scripts/topk_trace.sh 5893 3702 64 2>&1 | grep "gemm_f32\|nb="; done
  for ns in 0 3500; do echo "== RK_GEMM_STAGGER_NS=$ns (8192 x 34474 x 256)"; RK_GEMM_STAGGER_NS=$ns timeout 200 bash scripts/topk_trace.sh 8192 34474 256 2>&1 | grep "gemm_f32\|nb="; done
) > $o/r05q_gemm_stagger.txt 2>&1; cat $o/r05q_gemm_stagger.txt
