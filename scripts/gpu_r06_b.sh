#!/bin/bash
# Round 6, second pass: the whole GPU suite again (the first pass stopped at its first failure), SQ counters of the round-5/6
# spmm_lds_kernel (what bounds the gather phase), driver-style bench.   gpurun --timeout 2400 -- bash scripts/gpu_r06_b.sh
tag=r06b
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 600 bash scripts/lds_pmc.sh > $o/${tag}_spmm_lds_sq_counters.txt 2>&1; cat $o/${tag}_spmm_lds_sq_counters.txt
timeout 300 python scripts/spmm_lds_probe.py 2>/dev/null | tail -1 > $o/${tag}_spmm_lds_probe.json; cat $o/${tag}_spmm_lds_probe.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-config4 2>$o/${tag}_bench_s20.err | grep "^{" > $o/${tag}_bench_s20.json; tail -3 $o/${tag}_bench_s20.err
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_s20.json")); r = d["roofline"]; t = d.get("topk") or {}
print("%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
      "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"))
print("also.config3_yelp.parity", json.dumps(d["also"]["config3_yelp"].get("parity")))
PY
