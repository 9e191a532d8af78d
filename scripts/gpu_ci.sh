#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprof kernel stats.  Usage (from the container):
#   gpurun --timeout 1200 -- scripts/gpu_ci.sh r01_b
tag=${1:-run}
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py 2>&1 | tail -1 > gpurun_out/bench_${tag}.json
python bench.py --graph reference --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_asis.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --no-live-traffic --no-cpu-baseline > gpurun_out/bench_${tag}_profiled.json 2>gpurun_out/prof_${tag}.err
f=$(ls gpurun_out/prof_${tag}/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv; rm -rf gpurun_out/prof_${tag}
python - <<PY
import json
for n in ("bench_${tag}.json", "bench_${tag}_asis.json", "bench_${tag}_profiled.json"):
    d = json.load(open("gpurun_out/" + n)); r = d["roofline"]
    print(n, "%.3g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "spmm %.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]), "topk %.3g users/s" % d["topk"]["value"])
PY
head -4 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
