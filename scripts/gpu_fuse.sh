#!/bin/bash
# A/B of the multi-phase propagation launch against one launch per layer on ONE box: tests, bench lines, step traces.
tag=${1:-fuse}
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_layers or timed_path or lightgcn_train_golden or lightgcn_propagate_golden or deterministic_scatter or long_epoch or ncf_train_golden" 2>&1 | tail -15 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-topk 2>/dev/null | grep "^{" > $o/${tag}_bench_fused_$rep.json
  RK_LDS_NO_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-topk 2>/dev/null | grep "^{" > $o/${tag}_bench_unfused_$rep.json
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-topk 2>/dev/null | grep "^{" > $o/${tag}_bench_s20_fused.json
RK_LDS_NO_FUSE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-topk 2>/dev/null | grep "^{" > $o/${tag}_bench_s20_unfused.json
python3 - <<PY
import json
for n in ("bench_fused_1", "bench_unfused_1", "bench_fused_2", "bench_unfused_2", "bench_s20_fused", "bench_s20_unfused"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); print(n, "%.2f us/step" % (d["ms_per_step"] * 1e3), "loss", d["last_step_loss"], "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "missing", e)
PY
( timeout 300 bash scripts/step_trace.sh --no-topk; RK_LDS_NO_FUSE=1 timeout 300 bash scripts/step_trace.sh --no-topk ) > $o/${tag}_step_trace.txt 2>&1; cat $o/${tag}_step_trace.txt
