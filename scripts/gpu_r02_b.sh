#!/bin/bash
# round 2, second pass: new sharded tests (1 and 2 ranks on the box's GPU) + bench rows mode end to end over gloo
python -m pytest tests -m gpu -x -q -k "sharded" 2>&1 | tail -5
python bench.py --gpus 2 --backend gloo --workload ml1m --steps 10 --warmup 2 2>gpurun_out/r02_b_err1.txt | grep '^{' | tail -1 > gpurun_out/bench_r02_b_rows2_ml1m.json
tail -5 gpurun_out/r02_b_err1.txt
python bench.py --gpus 2 --backend gloo --workload yelp --steps 6 --warmup 2 2>gpurun_out/r02_b_err2.txt | grep '^{' | tail -1 > gpurun_out/bench_r02_b_rows2_yelp.json
tail -5 gpurun_out/r02_b_err2.txt
python bench.py --gpus 2 --backend gloo --parallel replicas --workload ml1m --steps 20 --warmup 5 2>gpurun_out/r02_b_err3.txt | grep '^{' | tail -1 > gpurun_out/bench_r02_b_repl2.json
tail -3 gpurun_out/r02_b_err3.txt
python - <<PY
import json
for n in ("bench_r02_b_rows2_ml1m.json", "bench_r02_b_rows2_yelp.json", "bench_r02_b_repl2.json"):
    try:
        d = json.load(open("gpurun_out/" + n))
        print(n, "%.3g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), d["scaling"], d["config"]["parallelism"], "| 1gpu:", d.get("same_workload_1gpu"), "| topk:", d.get("topk"))
    except Exception as e:
        print(n, "FAILED", e)
PY
