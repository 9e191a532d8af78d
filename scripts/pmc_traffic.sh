#!/bin/bash
# HBM traffic of the SpMM kernel from the PMC counters, one counter per pass (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE / WRITE_SIZE come from the TCC EA request counters, in KiB; on gfx950
# FETCH_SIZE counts 64 B per 128-B request for wide streaming reads => x2 for those).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in train reference; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=gpurun_out/pmc_${g}_${c}
    rocprofv3 --pmc $c --output-format csv -d $out -- python3 scripts/spmm_sweep.py $g > /dev/null 2>&1
    f=$(ls $out/*/*counter_collection.csv | head -1)
    python3 - "$f" "$g" "$c" <<'PY'
import csv, sys, statistics
f, g, c = sys.argv[1:4]
vals = []
for row in csv.DictReader(open(f)):
    if "spmm_csr" in row["Kernel_Name"] and row["Counter_Name"] == c:
        vals.append(float(row["Counter_Value"]))
print(g, c, "dispatches", len(vals), "median", statistics.median(vals), "mean", sum(vals) / len(vals), "min", min(vals), "max", max(vals))
PY
    rm -rf $out
  done
done
