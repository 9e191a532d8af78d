#!/bin/bash
# per-launch durations of one LightGCN train step by position, from rocprofv3: one launch per layer (f1 f2 f3 bpr b1 b2 b3) or
# the multi-phase form (F bpr B).  Extra arguments go to bench.py: one launch per layer is the default, --fuse-layers selects the latter.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/st_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-parity --no-also --no-topk --steps 64 --warmup 8 --graph-steps 0 "$@" > /dev/null 2>&1
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spmm_" in r["Kernel_Name"] or "bpr_kernel" in r["Kernel_Name"] or "bpr_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = "".join("s" if "spmm" in r["Kernel_Name"] else "b" for r in rows)
for pat, names in (("sssbsss", ["f1", "f2", "f3", "bpr", "b1", "b2", "b3"]), ("sbs", ["F", "bpr", "B"])):
    pos, gap, spans = collections.defaultdict(list), collections.defaultdict(list), []
    i, n = 0, len(pat)
    while i + n <= len(rows):
        # a step = the pattern NOT preceded / followed by further spmm launches of the same step
        if seq[i:i + n] == pat and (i == 0 or seq[i - 1] != "b") and (pat != "sbs" or ((i == 0 or seq[max(i - 2, 0):i] in ("bs", "s", "") or True))):
            if pat == "sbs" and ((i > 0 and seq[i - 1] == "s" and (i < 2 or seq[i - 2] == "s")) or seq[i + n:i + n + 2] == "ss"):
                i += 1
                continue
            for k in range(n):
                pos[k].append((int(rows[i + k]["End_Timestamp"]) - int(rows[i + k]["Start_Timestamp"])) / 1e3)
                if i + k > 0:
                    gap[k].append((int(rows[i + k]["Start_Timestamp"]) - int(rows[i + k - 1]["End_Timestamp"])) / 1e3)
            spans.append((int(rows[i + n - 1]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3)
            i += n
        else:
            i += 1
    if len(pos[0]) < 8:
        continue
    med = lambda v: sorted(v)[len(v) // 2]
    print(rows[0]["Kernel_Name"][:70])
    print(" ".join("%s %.2f" % (names[k], sum(v) / len(v)) for k, v in sorted(pos.items())), "| steps", len(pos[0]),
          "| sum %.1f us" % sum(sum(v) / len(v) for v in pos.values()), "| median first-start..last-end %.1f us" % med(spans))
    print("gap before: " + " ".join("%s %.2f" % (names[k], med(v)) for k, v in sorted(gap.items())), "| sum of median gaps %.1f us" % sum(med(v) for v in gap.values()))
    break
PY
rm -rf $out
