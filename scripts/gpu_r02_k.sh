#!/bin/bash
# one-rank RCCL group: the sharded trainer's real collective calls on a 1-GPU box
for w in ml1m yelp c4s; do
  python bench.py --gpus 1 --force-collectives --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r02_k_err_$w.txt | grep '^{' | tail -1 > gpurun_out/bench_r02_k_rows1_$w.json
  tail -2 gpurun_out/r02_k_err_$w.txt | grep -v amdgpu
done
python - <<PY
import json
for w in ("ml1m", "yelp", "c4s"):
    try:
        d = json.load(open(f"gpurun_out/bench_r02_k_rows1_{w}.json"))
        print(w, "%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), "| fused 1gpu:", d["same_workload_1gpu"]["ms_per_step"] * 1e3, "us/step | eval", d["topk"]["seconds"], d["topk"]["hr@50"], "| loss", d["last_step_loss"])
    except Exception as e:
        print(w, "FAILED", e)
PY
