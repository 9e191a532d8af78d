#!/bin/bash
# Round 6, mid-round pass on the tree with the grouped GEMM as its own instantiation and the trimmed timed region:
# whole GPU suite, driver-style + default bench, rocprofv3 kernel stats of the driver-style command, step / eval traces,
# yelp line, the N = 2 flow over gloo on one GPU (all eight sharded legs).   gpurun --timeout 3000 -- bash scripts/gpu_r06_f.sh
tag=r06f
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=recad_amd/lib/probes
( for shape in "5893 3702 64" "8192 34474 256" "54617 34474 128"; do for pad in 1 0; do
    echo -n "r05 ldpad=$pad: "; timeout 120 $P/gemm_probe_r05 $shape 0 0 1 0 1 $pad | tail -1
    echo -n "r06 ldpad=$pad: "; timeout 120 $P/gemm_probe_r06 $shape 0 0 1 0 1 $pad | tail -1
  done; done ) > $o/${tag}_gemm_ab.txt 2>&1; cat $o/${tag}_gemm_ab.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
for i in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $o/${tag}_bench_s20_$i.json; done
timeout 900 python bench.py 2>/dev/null | grep "^{" > $o/${tag}_bench.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag} -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-live-traffic > $o/${tag}_bench_profiled.json 2>/dev/null
f=$(ls $o/prof_${tag}/*/*kernel_stats.csv | head -1); cp $f $o/${tag}_bench_kernel_stats.csv; rm -rf $o/prof_${tag}
timeout 300 bash scripts/step_trace.sh --no-live-traffic > $o/${tag}_step_trace.txt 2>&1; cat $o/${tag}_step_trace.txt
timeout 300 bash scripts/eval_session_trace.sh 2>&1 | tail -12 > $o/${tag}_eval_session_trace.txt; cat $o/${tag}_eval_session_trace.txt
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
timeout 1700 python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 --also-timeout 1500 2>$o/${tag}_bench_n2_gloo.err | grep "^{" > $o/${tag}_bench_n2_gloo.json; echo "n2 rc=$?"; tail -3 $o/${tag}_bench_n2_gloo.err
python3 - <<PY
import json
for n in ("bench_s20_1", "bench_s20_2", "bench_s20_3", "bench", "bench_yelp_d128", "bench_profiled"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); r = d["roofline"]; t = d.get("topk") or {}
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
              "lds_frac", r.get("lds_frac"), "traffic", r.get("traffic"), "topk %.3g users/s %.1f us" % (t.get("value", 0), t.get("seconds", 0) * 1e6),
              "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity", (d.get("parity") or {}).get("ok"), "mfma", (d.get("mfma_gemm") or {}).get("frac"),
              "region", {k: round(v, 1) for k, v in d["timed_region"].items() if isinstance(v, float)})
    except Exception as e:
        print(n, "missing", e)
try:
    d = json.load(open("$o/${tag}_bench_n2_gloo.json")); print("n2:", d["n_gpus"], d["value"], d["scaling"], d["config"]["mode"])
    for k, v in (d.get("also") or {}).items():
        print("  ", k, (v.get("ms_per_step"), (v.get("exchange") or {}), (v.get("exposed_communication") or {}).get("exposed_comm_share"), v.get("error")) if isinstance(v, dict) else v)
except Exception as e:
    print("n2 missing", e)
PY
