#!/bin/bash
# round 5: score-matrix GEMM whose compiler-counted waits let a finished tile's stores drain under the next tile: parity, timings
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "score_topk or panel or stress or eval or mf or gemm" 2>&1 | tail -2
( for shape in "5893 3702 64" "3179 5600 64" "8192 34474 256" "54617 34474 128"; do timeout 200 bash scripts/topk_trace.sh $shape 2>&1 | grep -v amdgpu | head -4; done ) > $o/r05p_gemm_waits.txt 2>&1; cat $o/r05p_gemm_waits.txt
bash scripts/eval_session_trace.sh 2>&1 | tee $o/r05p_eval_session_trace.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic --no-cpu-baseline > $o/r05p_bench_s20.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05p_bench_s20.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["topk"]["seconds"], d["mfma_gemm"]["frac"], d["mfma_gemm"]["avg_launch_us"], d["parity"]["ok"])
PY
