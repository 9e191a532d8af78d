#!/bin/bash
# round 6: fresh PMC counters of the two evaluation kernels that changed (the grouped wide GEMM at k = 64 and at d = 256 through
# rk_score_topk's line-aligned score rows; the lighter topk_wave_kernel) -> gpurun_out/r06_counters.txt.  One counter set per pass;
# the profiled program stands directly behind `--`.
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pmc() {   # pmc "<counters>" <kernel substring> <label> -- program args...
  local counters="$1" kern="$2" label="$3"; shift 3
  local out=$o/pmc_$$; rm -rf $out
  timeout 600 rocprofv3 --pmc $counters --output-format csv -d $out -- "$@" > /dev/null 2>&1
  python3 - "$out" "$kern" "$label" <<'PY'
import csv, glob, sys, collections
d, kern, label = sys.argv[1:4]
v = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(v.items()):
    print("PMC %s %s %s dispatches %d mean %.1f" % (label, kern, k, len(x), sum(x) / len(x)))
PY
  rm -rf $out
}
{
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  pmc "$c" gemm_f32_wide gemm_8192x34474x256 python3 scripts/score_bench.py 8192 34474 256
done
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  pmc "$c" topk_wave topk_5893x3702 python3 scripts/score_bench.py 5893 3702 64
  pmc "$c" gemm_f32_wide gemm_5893x3702x64 python3 scripts/score_bench.py 5893 3702 64
done
timeout 200 bash scripts/topk_trace.sh 5893 3702 64
timeout 200 bash scripts/topk_trace.sh 8192 34474 256
} 2>&1 | grep -v amdgpu.ids | tee $o/r06_counters.txt
