#!/bin/bash
# GPU timeline of the driver-style invocation (bench.py --steps 20 --warmup 5): per call, the span from the first kernel's start
# to the last kernel's end against the sum of the kernel durations, from rocprofv3's kernel trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ct_$$
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --no-topk --no-also --no-parity --steps 20 --warmup 5 "$@" > gpurun_out/ct_bench.json 2>/dev/null
f=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# calls start with state_init_kernel
idx = [i for i, r in enumerate(rows) if "state_init" in r["Kernel_Name"]]
for a, b in zip(idx, idx[1:] + [len(rows)]):
    seg = rows[a:b]
    # cut the segment at the first kernel that is not part of a train call
    names = ("state_init", "spmm_", "bpr_", "rk_zero", "lds_pack", "zero_f4")
    k = 0
    while k < len(seg) and any(n in seg[k]["Kernel_Name"] for n in names):
        k += 1
    seg = seg[:k]
    if len(seg) < 10:
        continue
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    first_gap = (int(seg[1]["Start_Timestamp"]) - int(seg[0]["End_Timestamp"])) / 1e3
    print("call of %d kernels: span %.1f us, kernel time %.1f us, gap after state_init %.1f us; head: %s" % (
        len(seg), (t1 - t0) / 1e3, busy / 1e3, first_gap,
        " ".join("%s:%.1f" % (r["Kernel_Name"].split("(")[0].split("<")[0][-14:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in seg[:6])))
    print("   tail: " + " ".join("%s:%.1f" % (r["Kernel_Name"].split("(")[0].split("<")[0][-14:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in seg[-3:]))
PY
python3 -c "import json; d=json.loads(open('gpurun_out/ct_bench.json').read().strip().splitlines()[-1]); print('bench ms_per_step (under the profiler)', d['ms_per_step'])"
rm -rf $out
