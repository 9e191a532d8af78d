#!/bin/bash
# The clock the chip holds inside the fp32-MFMA kernels: GRBM_GUI_ACTIVE (busy cycles of the dispatch, gfx clock domain) over the
# kernel's duration (kernel trace, separate run).   bash scripts/mfma_clock_probe.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in panel unfused; do
  export PROBE_MODES=$mode
  out=gpurun_out/clk_$$; rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/score_probe.py 8192 34474 256 5 > /dev/null 2>&1
  ft=$(ls $out/*/*kernel_trace.csv | head -1); cp $ft /tmp/clk_trace_$mode.csv; rm -rf $out
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out -- python3 scripts/score_probe.py 8192 34474 256 5 > /dev/null 2>&1
  fc=$(ls $out/*/*counter_collection.csv | head -1); cp $fc /tmp/clk_pmc_$mode.csv; rm -rf $out
  python3 - $mode <<'PY'
import csv, sys, collections
mode = sys.argv[1]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"/tmp/clk_trace_{mode}.csv")):
    dur[r["Kernel_Name"].split("(")[0][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"/tmp/clk_pmc_{mode}.csv")):
    cnt[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in dur.items():
    if not any(s in k for s in ("score_panel", "gemm_f32_wide", "topk_")):
        continue
    v = sorted(v); d = v[len(v) // 2]
    c = cnt.get(k, {})
    line = f"{mode:8s} {k:40s} median {d / 1e3:9.1f} us"
    if "GRBM_GUI_ACTIVE" in c:
        g = sorted(c["GRBM_GUI_ACTIVE"]); g = g[len(g) // 2]
        line += f" | GRBM_GUI_ACTIVE {g:.4g} cycles -> {g / d:.3f} GHz (PMC run and trace run are separate launches)"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        m = sorted(c["SQ_VALU_MFMA_BUSY_CYCLES"]); m = m[len(m) // 2]
        line += f" | MFMA busy {m / (g * 1024):.3f} of the SIMD cycles"
    print(line)
PY
done
