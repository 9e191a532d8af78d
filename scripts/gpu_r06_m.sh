#!/bin/bash
# Round 6: the tree with (1) topk_rows_kernel's unconditional row loads, (2) the filtered first backward layer not reading the addend
# of non-frontier rows: whole GPU suite, the scoring paths side by side, yelp / config-4 step traces and lines
tag=r06m
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
( for shape in "5893 3702 64 20" "16384 34474 64 5" "54617 34474 128 3" "8192 34474 256 5"; do
    PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done ) > $o/${tag}_score_probe.txt; cat $o/${tag}_score_probe.txt
timeout 300 bash scripts/step_trace.sh --no-live-traffic --workload yelp > $o/${tag}_step_trace_yelp.txt 2>&1; cat $o/${tag}_step_trace_yelp.txt
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
timeout 600 python bench.py --workload config4 --no-cpu-baseline --eval-users 65536 --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_config4.json
timeout 400 bash scripts/step_trace.sh --no-live-traffic --workload config4 --steps 12 --warmup 3 > $o/${tag}_step_trace_config4.txt 2>&1; cat $o/${tag}_step_trace_config4.txt
python3 - <<PY
import json
for n in ("bench_yelp_d128", "bench_config4"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); r = d["roofline"]; t = d.get("topk") or {}
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]), "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "missing", e)
PY
