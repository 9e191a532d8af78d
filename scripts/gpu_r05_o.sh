#!/bin/bash
# round 5: intermediate layers written in the consumer's table order, pre-scaled (y_staged / x_staged): parity, step trace, bench
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "lightgcn or lds or spmm or golden or workflow or eval_session" 2>&1 | tail -2
timeout 300 bash scripts/step_trace.sh --no-live-traffic 2>&1 | tail -3 | tee $o/r05o_step_trace.txt
for i in 1 2; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic > $o/r05o_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05o_bench_s20_$i.json").read().strip().splitlines()[-1]); t=d["timed_region"]; print(d["ms_per_step"], t["gpu_span_us"], d["roofline"]["avg_launch_us"], d["parity"], d["topk"]["seconds"])
PY
done
timeout 400 python bench.py --no-also --no-live-traffic --no-cpu-baseline > $o/r05o_bench_default.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05o_bench_default.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], d["parity"]["ok"])
PY
