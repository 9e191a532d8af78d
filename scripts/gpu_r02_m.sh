#!/bin/bash
# d=256 scoring GEMM evidence after the wide kernel: kernel stats + three PMC passes (same recipe as gpu_r02_i.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/score_probe.py 8192 34474 256 5 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -- python3 scripts/score_probe.py 8192 34474 256 5 > /dev/null 2>&1
f=$(ls gpurun_out/prof_g/*/*kernel_stats.csv | head -1); grep -E '^"Name"|gemm_f32|topk' $f | cut -c1-200 > gpurun_out/r02_gemm_d256_wide_kernel_stats.csv; cat gpurun_out/r02_gemm_d256_wide_kernel_stats.csv; rm -rf gpurun_out/prof_g
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_g -- python3 scripts/score_probe.py 8192 34474 256 2 > /dev/null 2>&1
  f=$(ls gpurun_out/pmc_g/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "gemm_f32_wide_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("PMC", k, "%.5g" % (sum(v) / len(v)), "n", len(v))
PY
  rm -rf gpurun_out/pmc_g
done
