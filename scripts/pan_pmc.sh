#!/bin/bash
# SQ counters of score_panel_kernel (VALU / MFMA / SALU instruction counts, MFMA busy and co-execution cycles, waits):
#   bash scripts/pan_pmc.sh <n_users> <n_items> <dim>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PROBE_MODES=panel
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  out=gpurun_out/pmc_$$
  rm -rf $out
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out -- python3 scripts/score_probe.py "$@" 3 > /dev/null 2>&1
  f=$(ls $out/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "no counters for: $set"; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "score_panel_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s %16.0f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $out
done
