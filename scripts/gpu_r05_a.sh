#!/bin/bash
# round 5, first GPU pass: the whole GPU suite, the evaluation's kernel trace, the driver-style bench line (live PMC traffic),
# and the N > 1 flow end to end on one GPU (2 gloo ranks sharing it)
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $o/r05a_tests.txt; cat $o/r05a_tests.txt
timeout 300 bash scripts/eval_trace.sh > $o/r05a_eval_trace.txt 2>&1; cat $o/r05a_eval_trace.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $o/r05a_bench_s20.err | grep "^{" > $o/r05a_bench_s20.json; tail -3 $o/r05a_bench_s20.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r05a_bench_s20.json"))
print("s20:", "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), d["roofline"]["kernel"], "%.2f us" % d["roofline"]["avg_launch_us"],
      "frac %.3f lds_frac %.3f" % (d["roofline"]["frac"], d["roofline"].get("lds_frac", 0)), "traffic", d["roofline"]["traffic"], (d["roofline"]["traffic_source"] or "")[:60])
print("topk", d["topk"]["seconds"], "parity", d["parity"]["ok"], d["parity"]["max_rel_loss_err"], d["parity"]["tables_relerr"])
PY
timeout 1500 python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 --also-timeout 900 2> $o/r05a_bench_n2_gloo.err | grep "^{" > $o/r05a_bench_n2_gloo.json; tail -5 $o/r05a_bench_n2_gloo.err
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05a_bench_n2_gloo.json"))
    print("n2:", d["metric"], d["n_gpus"], d["value"], d["scaling"], d["config"]["mode"], d["ranks_seen"])
    for k, v in (d.get("also") or {}).items():
        print("  also", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "step_captured")} if isinstance(v, dict) else v)
        if isinstance(v, dict) and v.get("same_workload_1gpu"):
            print("     1gpu", v["same_workload_1gpu"]["ms_per_step"], "replicas", (v.get("same_workload_replicas") or {}).get("ms_per_step"))
except Exception as e:
    print("n2 line missing", e)
PY
