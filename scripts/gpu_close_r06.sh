#!/bin/bash
# Round 6 closing pass on one MI355X box (every command under its own timeout):
#   gpurun --timeout 3000 -- bash scripts/gpu_close_r06.sh
# -> gpurun_out/r05_*: GPU tests, smoke, the driver-style and the default bench line (live PMC traffic), variants, the larger
#    shapes, rocprofv3 kernel stats of the driver-style command, step / call / evaluation traces, MF / NCF, the workflow loop,
#    the N = 2 flow on one GPU (gloo), the scoring paths side by side.
tag=r06
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $o/${tag}_bench_s20.json
timeout 900 python bench.py 2>/dev/null | grep "^{" > $o/${tag}_bench.json
timeout 300 python bench.py --graph reference --no-cpu-baseline --no-also --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_asis.json
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-topk --no-also --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_ordered.json
timeout 300 python bench.py --spmm csr --no-cpu-baseline --no-also --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_ldsoff.json
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
timeout 600 python bench.py --workload config4 --no-cpu-baseline --eval-users 65536 --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_config4.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag} -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-live-traffic > $o/${tag}_bench_profiled.json 2>/dev/null
f=$(ls $o/prof_${tag}/*/*kernel_stats.csv | head -1); cp $f $o/${tag}_bench_kernel_stats.csv; rm -rf $o/prof_${tag}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag} -- python3 bench.py --no-cpu-baseline --no-also --no-live-traffic > /dev/null 2>&1
f=$(ls $o/prof_${tag}/*/*kernel_stats.csv | head -1); cp $f $o/${tag}_bench_default_kernel_stats.csv; rm -rf $o/prof_${tag}
timeout 300 python scripts/spmm_lds_probe.py 2>/dev/null | tail -1 > $o/${tag}_spmm_lds_probe.json
timeout 300 bash scripts/step_trace.sh --no-live-traffic > $o/${tag}_step_trace.txt 2>&1; cat $o/${tag}_step_trace.txt
timeout 300 bash scripts/call_trace.sh --no-live-traffic > $o/${tag}_call_trace.txt 2>&1
timeout 300 bash scripts/eval_session_trace.sh 2>&1 | tail -12 > $o/${tag}_eval_session_trace.txt; cat $o/${tag}_eval_session_trace.txt
timeout 600 python scripts/bench_victims.py > /dev/null 2>&1; cp $o/bench_victims.json $o/${tag}_bench_victims.json
timeout 600 python bench.py --workflow --rec-epoch 10 2>/dev/null | grep "^{" > $o/${tag}_workflow_ml1m.json
( for shape in "5893 3702 64 20" "16384 34474 64 5" "54617 34474 128 3" "8192 34474 256 5"; do
    PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done ) > $o/${tag}_score_probe.txt; cat $o/${tag}_score_probe.txt
bash scripts/topk_wave_probe.sh 5893 3702 58 > $o/${tag}_topk_wave_probe.txt 2>&1
timeout 300 bash scripts/step_trace.sh --no-live-traffic --workload yelp > $o/${tag}_step_trace_yelp.txt 2>&1; cat $o/${tag}_step_trace_yelp.txt
python3 - <<PY
import json
for n in ("bench_s20", "bench", "bench_asis", "bench_ordered", "bench_ldsoff", "bench_yelp_d128", "bench_config4", "bench_profiled"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); r = d["roofline"]; t = d.get("topk") or {}
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
              "lds_frac", r.get("lds_frac"), "traffic", r.get("traffic"), "topk %.3g users/s %.1f us" % (t.get("value", 0), t.get("seconds", 0) * 1e6),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "missing", e)
try:
    d = json.load(open("$o/${tag}_bench_n2_gloo.json")); print("n2:", d["n_gpus"], d["value"], d["scaling"], d["config"]["mode"], list((d.get("also") or {}).keys()))
except Exception as e:
    print("n2 missing", e)
PY
