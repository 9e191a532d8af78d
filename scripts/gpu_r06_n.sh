#!/bin/bash
# Round 6: row-filtered launches whose unwanted waves end early: LightGCN / SpMM / sharded tests, yelp + config-4 step traces
tag=r06n
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -k "lightgcn or spmm or sharded or grid2d or full_size or workflow or norm_adj" 2>&1 | tail -4 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 bash scripts/step_trace.sh --no-live-traffic --workload yelp > $o/${tag}_step_trace_yelp.txt 2>&1; cat $o/${tag}_step_trace_yelp.txt
timeout 400 bash scripts/step_trace.sh --no-live-traffic --workload config4 --steps 12 --warmup 3 > $o/${tag}_step_trace_config4.txt 2>&1; cat $o/${tag}_step_trace_config4.txt
timeout 300 bash scripts/step_trace.sh --no-live-traffic --spmm csr > $o/${tag}_step_trace_ml1m_csr.txt 2>&1; cat $o/${tag}_step_trace_ml1m_csr.txt
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_yelp_d128.json")); r = d["roofline"]
print("yelp %.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "parity", d["parity"])
PY
