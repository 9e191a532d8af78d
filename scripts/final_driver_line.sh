#!/bin/bash
# the driver's command, wall clock around it, and the fields the contract names
cd $GRAFT_REPO_ROOT
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_driver_line.json 2> gpurun_out/final_driver_err.txt
t1=$(date +%s.%N)
echo "wall clock of the whole command: $(echo "$t1 - $t0" | bc) s; stdout lines: $(wc -l < gpurun_out/final_driver_line.json)"
python - <<PY
import json; d=json.loads(open("gpurun_out/final_driver_line.json").read().strip().splitlines()[-1]); print(d["metric"], d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["vs_baseline"], d["dtype"], d["config"]["workload"][:60]); print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_frac"), d["roofline"]["lds_frac"]); print(d["cpu_baseline"]); print(d["parity"]["ok"], list(d["also"].keys()))
PY
