#!/bin/bash
# round 5: the driver's bench command, probe order A/B on one box (probe in front of the timed region / before the warm-up)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in "" "--roofline-first"; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic --no-cpu-baseline $v > $o/r05f_bench_s20_$i$v.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05f_bench_s20_$i$v.json").read().strip().splitlines()[-1]); t=d["timed_region"]; print("$v", d["ms_per_step"], t["gpu_span_us"], t["c_call_entered_after_us"], t["c_call_us"], t["enqueue_returns_after_us"], d["roofline"]["avg_launch_us"], d["parity"]["ok"])
PY
done; done
