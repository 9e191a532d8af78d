#!/bin/bash
# per-launch durations of one LightGCN train step by position (f1 f2 f3 bpr b1 b2 b3), from rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/st_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 64 --warmup 8 --graph-steps 0 "$@" > /dev/null 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spmm_" in r["Kernel_Name"] or "bpr_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the training region: sequences of (spmm x3, bpr, spmm x3)
seq = ["s" if "spmm" in r["Kernel_Name"] else "b" for r in rows]
pos = collections.defaultdict(list)
gap = collections.defaultdict(list)
i = 0
while i + 7 <= len(rows):
    if seq[i:i + 7] == list("sssbsss"):
        for k in range(7):
            pos[k].append((int(rows[i + k]["End_Timestamp"]) - int(rows[i + k]["Start_Timestamp"])) / 1e3)
            if i + k > 0:
                gap[k].append((int(rows[i + k]["Start_Timestamp"]) - int(rows[i + k - 1]["End_Timestamp"])) / 1e3)
        i += 7
    else:
        i += 1
names = ["f1", "f2", "f3", "bpr", "b1", "b2", "b3"]
print(rows[0]["Kernel_Name"][:60])
print(" ".join("%s %.2f" % (names[k], sum(v) / len(v)) for k, v in sorted(pos.items())), "| steps", len(pos[0]), "| sum %.1f us" % sum(sum(v) / len(v) for v in pos.values()))
med = lambda v: sorted(v)[len(v) // 2]
print("gap before: " + " ".join("%s %.2f" % (names[k], med(v)) for k, v in sorted(gap.items())), "| sum of median gaps %.1f us" % sum(med(v) for v in gap.values()))
PY
rm -rf $out
