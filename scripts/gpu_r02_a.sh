#!/bin/bash
# round 2, first pass: parity tests + the driver's bench invocation (steps 20 / warmup 5) + default bench
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_a_err1.txt | tail -1 > gpurun_out/bench_r02_a_s20.json
python bench.py 2>gpurun_out/r02_a_err2.txt | tail -1 > gpurun_out/bench_r02_a.json
python - <<PY
import json
for n in ("bench_r02_a_s20.json", "bench_r02_a.json"):
    d = json.load(open("gpurun_out/" + n)); r = d["roofline"]
    print(n, "%.3g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "spmm %.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]), "topk %.3g users/s" % d["topk"]["value"], "epoch", d["epoch_with_sampler"], "cpu", d["cpu_baseline"], d["cpu_baseline_aten"])
PY
tail -3 gpurun_out/r02_a_err1.txt gpurun_out/r02_a_err2.txt
