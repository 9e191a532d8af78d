#!/bin/bash
# round 2 checkpoint: full GPU suite, smoke, bench lines (driver-style + default + big shapes), rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r02_h_err1.txt | tail -1 > gpurun_out/bench_r02_h_s20.json
python bench.py 2>gpurun_out/r02_h_err2.txt | tail -1 > gpurun_out/bench_r02_h.json
python bench.py --graph reference --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r02_h_asis.json
python bench.py --workload yelp --no-cpu-baseline --steps 100 --warmup 16 2>gpurun_out/r02_h_err3.txt | tail -1 > gpurun_out/bench_r02_h_yelp_d128.json
python bench.py --workload c4s --no-cpu-baseline --steps 16 --warmup 8 2>gpurun_out/r02_h_err4.txt | tail -1 > gpurun_out/bench_r02_h_c4s.json
python bench.py --workload config4 --no-cpu-baseline --steps 8 --warmup 8 --eval-users 65536 2>gpurun_out/r02_h_err5.txt | tail -1 > gpurun_out/bench_r02_h_config4.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_h -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_r02_h_profiled.json 2>gpurun_out/prof_r02_h.err
f=$(ls gpurun_out/prof_r02_h/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r02_h_kernel_stats.csv; rm -rf gpurun_out/prof_r02_h
python - <<PY
import json
for n in ("bench_r02_h_s20.json", "bench_r02_h.json", "bench_r02_h_asis.json", "bench_r02_h_yelp_d128.json", "bench_r02_h_c4s.json", "bench_r02_h_config4.json", "bench_r02_h_profiled.json"):
    try:
        d = json.load(open("gpurun_out/" + n)); r = d["roofline"]
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "spmm %.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]), "topk %.3g users/s %.1f us" % (d["topk"]["value"], d["topk"]["seconds"] * 1e6), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_aten") or {}).get("value"), (d.get("cpu_baseline_aten") or {}).get("cores"))
    except Exception as e:
        print(n, "FAILED", e)
PY
head -6 gpurun_out/r02_h_kernel_stats.csv | cut -c1-160
tail -2 gpurun_out/r02_h_err5.txt
