#!/bin/bash
# GPU kernels of one full evaluation (ml1m-shaped LightGCN), in launch order, from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/eval_host_profile.py > /dev/null 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last evaluation = from the last-but-one topk kernel's successor to the end
idx = [i for i, r in enumerate(rows) if "topk_" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
# include trailing HR reduction kernels of the last evaluation
seg = rows[lo:min(len(rows), hi + 6)]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us +%7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
PY
rm -rf $out
