#!/bin/bash
# round 5: selection kernel A/B (wave-per-row vs row-per-workgroup) and shallow-k GEMM variants at the headline evaluation shape
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "topk or score or eval or golden" 2>&1 | tail -3
export RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so
( echo "== default"; timeout 200 bash scripts/topk_trace.sh 5893 3702 64
  echo "== RK_TOPK_NO_WAVE=1"; RK_TOPK_NO_WAVE=1 timeout 200 bash scripts/topk_trace.sh 5893 3702 64
  for v in "RK_GEMM_WIDE_MINK=128" "RK_GEMM_WIDE_MINK=128 RK_GEMM_TPB=2" "RK_GEMM_WIDE_MINK=128 RK_GEMM_TPB=3" "RK_GEMM_WIDE_WGS=768" "RK_GEMM_WIDE_WGS=1024" "RK_GEMM_WIDE_WGS=1363" "RK_GEMM_WIDE_WGS=256"; do
    echo "== $v"; env $v timeout 200 bash scripts/topk_trace.sh 5893 3702 64
  done
  echo "== game shape 3179 5600 64"; timeout 200 bash scripts/topk_trace.sh 3179 5600 64
  echo "== game shape, RK_TOPK_NO_WAVE=1"; RK_TOPK_NO_WAVE=1 timeout 200 bash scripts/topk_trace.sh 3179 5600 64
  echo "== 8192 x 2000 x 64"; timeout 200 bash scripts/topk_trace.sh 8192 2000 64
  echo "== 8192 x 2000 x 64 RK_TOPK_NO_WAVE=1"; RK_TOPK_NO_WAVE=1 timeout 200 bash scripts/topk_trace.sh 8192 2000 64
) > $o/r05b_topk_ab.txt 2>&1
cat $o/r05b_topk_ab.txt
