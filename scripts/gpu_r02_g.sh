#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 3000; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sel -- python3 scripts/score_probe.py 5893 3702 64 20 > /dev/null 2>&1
f=$(ls gpurun_out/prof_sel/*/*kernel_stats.csv | head -1); head -5 $f | cut -c1-160; rm -rf gpurun_out/prof_sel
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_sel -- python3 scripts/score_probe.py 5893 3702 64 3 > /dev/null 2>&1
  f=$(ls gpurun_out/pmc_sel/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "score_select" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(k, "%.4g" % (sum(v) / len(v)), "n", len(v))
PY
  rm -rf gpurun_out/pmc_sel
done
