#!/bin/bash
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic --no-cpu-baseline > $o/r05g_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05g_bench_s20_$i.json").read().strip().splitlines()[-1]); t=d["timed_region"]; print(d["ms_per_step"], t["gpu_span_us"], t["host_path_us"], t["enqueue_returns_after_us"])
PY
done
