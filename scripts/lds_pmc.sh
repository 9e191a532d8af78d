#!/bin/bash
# SQ counters of spmm_lds_kernel (LDS bank conflicts, LDS-array cycles, VALU / LDS instruction counts, busy cycles)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"; do
  out=gpurun_out/pmc_$$
  rm -rf $out
  rocprofv3 --pmc $set --output-format csv -d $out -- python3 scripts/spmm_lds_probe.py --iters 10 "$@" > /dev/null 2>&1
  f=$(ls $out/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_lds_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print("%-24s %14.0f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $out
done
