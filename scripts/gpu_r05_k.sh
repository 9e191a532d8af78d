#!/bin/bash
# round 5: after the chain's revert: lightgcn parity, the driver's bench command x3, the default run
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "lightgcn or workflow or timed" 2>&1 | tail -2
for i in 1 2 3; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/r05k_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05k_bench_s20_$i.json").read().strip().splitlines()[-1]); t=d["timed_region"]; print(d["ms_per_step"], t["gpu_span_us"], t["host_path_us"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["lds_frac"], d["parity"]["ok"], d["topk"].get("seconds"))
PY
done
timeout 400 python bench.py > $o/r05k_bench_default.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05k_bench_default.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["parity"]["ok"])
PY
