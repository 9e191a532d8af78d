#!/bin/bash
# HBM-side traffic of the SpMM kernel from the PMC counters, one counter per pass (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE / WRITE_SIZE come from the TCC EA request counters, in KiB; on gfx950
# FETCH_SIZE counts 64 B per 128-B request for wide streaming reads => x2 for those), plus the L2 hit rate.
#   usage: scripts/pmc_traffic.sh "<graph> <workload> <dim> <reps>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    out=gpurun_out/pmc_tr_$$
    rocprofv3 --pmc $c --output-format csv -d $out -- python3 scripts/spmm_sweep.py $spec > /dev/null 2>&1
    f=$(ls $out/*/*counter_collection.csv | head -1)
    python3 - "$f" "$spec" <<'PY'
import csv, sys, statistics, collections
f, spec = sys.argv[1:3]
vals = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "spmm_csr" in row["Kernel_Name"]:
        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in vals.items():
    print("PMC", spec.replace(" ", "_"), k, "dispatches", len(v), "mean", sum(v) / len(v), "median", statistics.median(v))
PY
    rm -rf $out
  done
done
