#!/bin/bash
# kernels of ONE NCF train step in launch order with durations: scripts/ncf_step_trace.sh <factor> <layers>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ncfs_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/ncf_prof.py "$@" > /dev/null 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "multi_adam" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us +%7.1f  grid %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"), r["Kernel_Name"][:60]))
PY
rm -rf $out
