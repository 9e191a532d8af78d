#!/bin/bash
# Round 6: A/B of the wide GEMM's grouped last chunk (round-5 kernel vs this tree, same box, alternating), per-tile stamps,
# bit-exact scoring tests, the evaluation session by kernel.   gpurun --timeout 1500 -- bash scripts/gpu_r06_c.sh
tag=r06c
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=recad_amd/lib/probes
( for rep in 1 2 3; do
    for shape in "5893 3702 64" "8192 34474 256" "54617 34474 128" "16384 34474 64"; do
      echo -n "r05: "; timeout 120 $P/gemm_probe_r05 $shape 0 0 1 0 1 | tail -1
      echo -n "r06: "; timeout 120 $P/gemm_probe_r06 $shape 0 0 1 0 1 | tail -1
    done
  done
  timeout 120 $P/gemm_probe_r06_stamps 5893 3702 64 0 0 1 0 1 ) > $o/${tag}_gemm_ab.txt 2>&1; cat $o/${tag}_gemm_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "score_topk or topk_rows or eval_session or eval_golden or users_rating or wide_gemm or randomised_stress or ncf_init_eval or ncf_train_golden" 2>&1 | tail -5 | tee $o/${tag}_tests.txt
timeout 300 bash scripts/eval_session_trace.sh > $o/${tag}_eval_session_trace.txt 2>&1; cat $o/${tag}_eval_session_trace.txt
