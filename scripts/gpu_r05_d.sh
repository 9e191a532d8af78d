#!/bin/bash
# round 5: whole-call replay as a chain of graphs (3-step head first): parity, call-overhead probe, the driver's bench command
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "lightgcn or workflow or timed" 2>&1 | tail -3
( timeout 200 python scripts/call_overhead_probe.py 20; PROBE_FIRST=1 timeout 200 python scripts/call_overhead_probe.py 20; timeout 200 python scripts/call_overhead_probe.py 64 ) > $o/r05d_call_overhead.txt 2>&1
cat $o/r05d_call_overhead.txt
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded > $o/r05d_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05d_bench_s20_$i.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["timed_region"]["gpu_span_us"], d["timed_region"]["enqueue_returns_after_us"], d["roofline"]["avg_launch_us"], d["parity"])
PY
done
