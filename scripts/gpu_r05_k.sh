#!/bin/bash
# round 5: host path of an epoch call (handle fingerprint, raw stream accessor): full GPU suite (handles are rebuilt in many tests), bench x3
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 200 python scripts/host_path_profile.py 2>&1 | grep "^per call"
for i in 1 2 3; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded --no-live-traffic > $o/r05k_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05k_bench_s20_$i.json").read().strip().splitlines()[-1]); t=d["timed_region"]; print(d["ms_per_step"], t["gpu_span_us"], t["host_path_us"], d["roofline"]["avg_launch_us"], d["parity"]["ok"])
PY
done
