#!/bin/bash
# rocprofv3 per-kernel averages of the scoring GEMM + top-K kernels: scripts/topk_trace.sh nb I d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/tk_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/score_bench.py "$@" 2>&1 | grep "nb="
f=$(ls $out/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "topk" in r["Name"] or "gemm" in r["Name"]:
        print("%-40s calls %4s avg %9.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out
