#!/bin/bash
# rocprofv3 per-kernel stats of 40 NCF train steps: scripts/ncf_trace.sh <factor> <layers>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ncf_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/ncf_prof.py "$@" > /dev/null 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-60s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f us per step (40 steps + warm)" % (tot / 1e3 / 80))
PY
rm -rf $out
