#!/bin/bash
# usage: spmm_trace.sh <graph> ; prints rocprof avg/min kernel ns for the spmm kernel under env settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_$$_$RANDOM
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/spmm_sweep.py $1 > /dev/null 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
python3 - "$f" "$1 hot=${RK_HOT:-0} dbg=${RK_SPMM_DEBUG:-0} var=${RK_SPMM_VARIANT:-0} seg=${RK_SEG_NNZ:-64}" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm" in r["Name"]: print(sys.argv[2], "::", r["Name"][:44], "calls", r["Calls"], "avg_us %.2f" % (float(r["AverageNs"]) / 1e3), "min_us %.2f" % (float(r["MinNs"]) / 1e3))
PY
rm -rf $out
