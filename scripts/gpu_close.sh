#!/bin/bash
# The round's closing pass on one MI355X box (every command under its own timeout):
#   gpurun --timeout 2400 -- scripts/gpu_close.sh r03
# -> gpurun_out/<tag>_*: GPU tests, smoke, the driver-style and the default bench line, the as-is graph, the ordered
#    scatter, the larger shapes, rocprofv3 kernel stats of the default bench, PMC traffic of the dominant kernel
#    (FETCH_SIZE / WRITE_SIZE / L2 hits in separate passes), the LDS kernel's in-kernel stamps and SQ counters, the scoring
#    paths side by side (score_probe), MF / NCF (f256 / L5 with its step trace).
tag=${1:-run}
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $o/${tag}_bench_s20.json
timeout 600 python bench.py 2>/dev/null | grep "^{" > $o/${tag}_bench.json
timeout 300 python bench.py --graph reference --no-cpu-baseline --no-also 2>/dev/null | grep "^{" > $o/${tag}_bench_asis.json
timeout 300 python bench.py --deterministic --no-cpu-baseline --no-topk --no-also 2>/dev/null | grep "^{" > $o/${tag}_bench_ordered.json
timeout 300 python bench.py --spmm csr --no-cpu-baseline --no-also 2>/dev/null | grep "^{" > $o/${tag}_bench_ldsoff.json
timeout 300 python bench.py --fuse-layers --no-cpu-baseline --no-also --no-topk 2>/dev/null | grep "^{" > $o/${tag}_bench_fused_layers.json
timeout 300 python bench.py --workload yelp --no-cpu-baseline 2>/dev/null | grep "^{" > $o/${tag}_bench_yelp_d128.json
timeout 400 python bench.py --workload c4s --no-cpu-baseline 2>/dev/null | grep "^{" > $o/${tag}_bench_c4s.json
timeout 600 python bench.py --workload config4 --no-cpu-baseline --eval-users 65536 2>/dev/null | grep "^{" > $o/${tag}_bench_config4.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag} -- python3 bench.py --no-live-traffic --no-cpu-baseline --no-also > $o/${tag}_bench_profiled.json 2>/dev/null
f=$(ls $o/prof_${tag}/*/*kernel_stats.csv | head -1); cp $f $o/${tag}_bench_kernel_stats.csv; rm -rf $o/prof_${tag}
# PMC traffic of spmm_lds_kernel: one counter per pass (MI355X_MICROARCH.md "HBM": FETCH_SIZE x2 on gfx950, WRITE_SIZE as is)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf $o/pmc_${tag}
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $o/pmc_${tag} -- python3 scripts/spmm_lds_probe.py --iters 20 > /dev/null 2>&1
  f=$(ls $o/pmc_${tag}/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY' >> $o/${tag}_spmm_lds_pmc.txt
import csv, sys, collections, statistics
vals = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    for k in ("spmm_lds_kernel", "spmm_csr_kernel"):
        if k in row["Kernel_Name"]:
            vals[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    print("PMC", k, c, "dispatches", len(v), "mean", sum(v) / len(v), "median", statistics.median(v))
PY
  rm -rf $o/pmc_${tag}
done
cat $o/${tag}_spmm_lds_pmc.txt
timeout 300 python scripts/spmm_lds_probe.py 2>/dev/null | tail -1 > $o/${tag}_spmm_lds_probe.json
timeout 400 bash scripts/lds_pmc.sh > $o/${tag}_spmm_lds_sq_counters.txt 2>&1
timeout 300 bash scripts/step_trace.sh > $o/${tag}_step_trace.txt 2>&1; cat $o/${tag}_step_trace.txt
timeout 300 bash scripts/call_trace.sh > $o/${tag}_call_trace.txt 2>&1
timeout 600 python scripts/bench_victims.py > /dev/null 2>&1; cp $o/bench_victims.json $o/${tag}_bench_victims.json
# the register-resident panel scoring form against GEMM + selection (and the older sweep), its phase stamps and SQ counters
( for shape in "5893 3702 64 20" "16384 34474 64 5" "54617 34474 128 3" "8192 34474 256 5" "16384 131072 64 3"; do
    PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done
  PROBE_MODES=panel,unfused timeout 600 python3 scripts/score_probe.py 16384 500000 64 2 2>/dev/null | grep -v amdgpu.ids
  PROBE_ROWS=16 PROBE_MODES=panel timeout 300 python3 scripts/score_probe.py 54617 34474 128 3 2>/dev/null | grep "^panel" | sed 's/^panel /panel (16-row workgroups) /' ) > $o/${tag}_score_probe.txt; cat $o/${tag}_score_probe.txt
timeout 300 bash scripts/ncf_step_trace.sh 256 5 > $o/${tag}_ncf_f256_l5_step_trace.txt 2>&1
python3 - <<PY
import json
for n in ("bench_s20", "bench", "bench_asis", "bench_ordered", "bench_ldsoff", "bench_fused_layers", "bench_yelp_d128", "bench_c4s", "bench_config4", "bench_profiled"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); r = d["roofline"]; t = d.get("topk") or {}
        print(n, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
              "topk %.3g users/s" % t.get("value", 0), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "missing", e)
PY
