#!/bin/bash
# GPU timeline of EvalSession replays (ml1m-shaped LightGCN): per evaluation the kernels' sum against the span, and the gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/eval_session_probe.py 2>/dev/null | grep evaluations
out=gpurun_out/evs_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/eval_session_probe.py > /dev/null 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lds_pack" in r["Kernel_Name"]]
ev = [rows[a:b] for a, b in zip(idx, idx[1:])][-11:]   # the last run of 12 evaluations, minus the final one
spans, sums, gaps = [], [], []
for seg in ev:
    seg = seg[:7]
    spans.append((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3)
    sums.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
for a, b in zip(ev, ev[1:]):
    gaps.append((int(b[0]["Start_Timestamp"]) - int(a[6]["End_Timestamp"])) / 1e3)
med = lambda x: sorted(x)[len(x) // 2]
print("per evaluation (median of %d): kernels %.1f us, first start .. last end %.1f us, gap to the next evaluation's first kernel %.1f us" % (len(ev), med(sums), med(spans), med(gaps)))
seg = ev[-1][:7]; t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print("  %7.1f us +%6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
PY
rm -rf $out
