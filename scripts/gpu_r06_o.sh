#!/bin/bash
# Round 6, final pass on the closing tree: whole GPU suite, smoke, the driver-style line (x2), the default line, yelp / config-4 lines
tag=r06o
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $o/${tag}_bench_s20_$i.json; done
timeout 900 python bench.py 2>/dev/null | grep "^{" > $o/${tag}_bench.json
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
timeout 600 python bench.py --workload config4 --no-cpu-baseline --eval-users 65536 --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_config4.json
timeout 400 python bench.py --workload c4s --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_c4s.json
python3 - <<PY
import json
for n in ("bench_s20_1", "bench_s20_2", "bench", "bench_yelp_d128", "bench_config4", "bench_c4s"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); r = d["roofline"]; t = d.get("topk") or {}
        a = (d.get("also") or {})
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
              "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"), "mfma", (d.get("mfma_gemm") or {}).get("frac"),
              "also yelp", (a.get("config3_yelp") or {}).get("ms_per_step"), ((a.get("config3_yelp") or {}).get("parity") or {}).get("ok"), "c4", (a.get("config4") or {}).get("ms_per_step"))
    except Exception as e:
        print(n, "missing", e)
PY
