#!/bin/bash
# round 5: wide GEMM instantiations without possible accumulator-init loads (INIT template flag): parity, NCF / MF timings, the d = 256 probe
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "ncf or mf or score or gemm or stress or eval" 2>&1 | tail -2
timeout 600 python scripts/bench_victims.py > /dev/null 2>&1; python - <<PY
import json
v=json.load(open("$o/bench_victims.json"))
for k,x in v.items(): print(k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in x.items() if a in ("us_per_step","train_tflops","eval_users_per_s","eval_tflops")})
PY
cp $o/bench_victims.json $o/r05r_bench_victims.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic --no-cpu-baseline > $o/r05r_bench_s20.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05r_bench_s20.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["topk"]["seconds"], d["mfma_gemm"]["frac"], d["mfma_gemm"]["avg_launch_us"], d["parity"]["ok"])
PY
