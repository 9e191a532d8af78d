#!/bin/bash
# Round 6: whole GPU suite + smoke + driver-style bench on the tree with the lighter selection kernel
tag=r06i
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $o/${tag}_smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-config4 2>/dev/null | grep "^{" > $o/${tag}_bench_s20.json
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_s20.json")); r = d["roofline"]; t = d.get("topk") or {}
print("%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
      "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"), "mfma", d["mfma_gemm"]["frac"])
PY
