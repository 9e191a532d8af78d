#!/bin/bash
# A/B of the multi-phase propagation launch against one launch per layer on ONE box: probe, tests, bench lines, per-item stamps.
# Every command under its own SHORT timeout (a hand-off bug shows up as a hang).
tag=${1:-fuse}
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 60 python scripts/lds_multi_probe.py lightgcn_dev_d64 2 2>&1 | grep -v amdgpu.ids | tail -8
timeout 60 python scripts/lds_multi_probe.py lightgcn_game_d64_tg 3 --time 2>&1 | grep -v amdgpu.ids | tail -8
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_layers or timed_path or ncf_train_golden" 2>&1 | tail -15 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
for rep in 1 2; do
  timeout 120 python bench.py --fuse-layers --no-cpu-baseline --no-topk --no-also 2>/dev/null | grep "^{" > $o/${tag}_bench_fused_$rep.json
  timeout 120 python bench.py --no-cpu-baseline --no-topk --no-also 2>/dev/null | grep "^{" > $o/${tag}_bench_unfused_$rep.json
done
python3 - <<PY
import json
for n in ("bench_fused_1", "bench_unfused_1", "bench_fused_2", "bench_unfused_2"):
    try:
        d = json.load(open("$o/${tag}_" + n + ".json")); print(n, "%.2f us/step" % (d["ms_per_step"] * 1e3), "loss", d["last_step_loss"], "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "missing", e)
PY
RECAD_TUNING_LIB=$PWD/recad_amd/lib/librecad_hip_tuning.so RK_LDS_MSTAMPS=1 timeout 100 python scripts/lds_multi_stamps.py 2>&1 | grep -v amdgpu.ids | tail -12 > $o/${tag}_multi_stamps.txt; cat $o/${tag}_multi_stamps.txt
