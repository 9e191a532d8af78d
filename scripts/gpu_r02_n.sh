#!/bin/bash
# round 2, closing pass: driver-style and default bench + kernel stats, the larger single-GPU workloads, smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02_n_bench_s20.json
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r02_n_bench.json
python3 bench.py --deterministic --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_n_bench_ordered.json
python3 bench.py --workload yelp --dim 128 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02_n_bench_yelp_d128.json
python3 bench.py --workload c4s --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1 > gpurun_out/r02_n_bench_c4s.json
python3 bench.py --workload config4 --no-cpu-baseline --steps 10 --warmup 8 2>/dev/null | tail -1 > gpurun_out/r02_n_bench_config4.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
f=$(ls gpurun_out/prof_n/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-260 > gpurun_out/r02_n_bench_kernel_stats.csv; rm -rf gpurun_out/prof_n
python3 - <<PY
import json
for n in ("s20", "", "ordered", "yelp_d128", "c4s", "config4"):
    fn = "gpurun_out/r02_n_bench" + ("_" + n if n else "") + ".json"
    try:
        d = json.load(open(fn)); r = d["roofline"]; t = d.get("topk") or {}
        print(n or "default", "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "spmm %.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
              "topk %.4g users/s %.4g s" % (t.get("value", 0), t.get("seconds", 0)))
    except Exception as e:
        print(n, "FAILED", e)
PY
head -4 gpurun_out/r02_n_bench_kernel_stats.csv | cut -c1-160
