#!/bin/bash
# L1 (TCP) behaviour of the SpMM gather: accesses vs requests forwarded to L2.   usage: scripts/pmc_l1.sh "<graph> <workload> <dim> <reps>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
    out=gpurun_out/pmc_l1_$$
    rocprofv3 --pmc $c --output-format csv -d $out -- python3 scripts/spmm_sweep.py $spec > /dev/null 2>&1
    f=$(ls $out/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "no counters for: $c"; continue; }
    python3 - "$f" "$spec" <<'PY'
import csv, sys, statistics, collections
f, spec = sys.argv[1:3]
vals = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "spmm_csr" in row["Kernel_Name"]:
        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in vals.items():
    print("PMC", spec.replace(" ", "_"), k, "dispatches", len(v), "mean %.5g" % (sum(v) / len(v)))
PY
    rm -rf $out
  done
done
