#!/bin/bash
# Round 6, first GPU pass: the whole GPU suite (new: yelp / c4s timed path vs oracle, NCF init-eval goldens, top-K advice tests),
# smoke, the driver-style bench line (also.config3_yelp.parity).   gpurun --timeout 1800 -- bash scripts/gpu_r06_a.sh
tag=r06a
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$o/${tag}_bench_s20.err | grep "^{" > $o/${tag}_bench_s20.json; tail -3 $o/${tag}_bench_s20.err
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_s20.json")); r = d["roofline"]; t = d.get("topk") or {}
print("%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
      "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"))
print("also.config3_yelp.parity", json.dumps(d["also"]["config3_yelp"].get("parity")))
print("also.config3_yelp ms/step", d["also"]["config3_yelp"]["ms_per_step"], "config4", d["also"].get("config4", {}).get("ms_per_step"))
PY
