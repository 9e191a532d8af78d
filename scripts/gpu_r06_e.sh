#!/bin/bash
# Round 6: final grouped wide GEMM (line-aligned C only): A/B probe once more, scoring / evaluation tests, eval trace, bench s20
tag=r06e
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=recad_amd/lib/probes
( for rep in 1 2; do
    for shape in "5893 3702 64" "8192 34474 256" "54617 34474 128"; do
      for pad in 1 0; do
        echo -n "r05 ldpad=$pad: "; timeout 120 $P/gemm_probe_r05 $shape 0 0 1 0 1 $pad | tail -1
        echo -n "r06 ldpad=$pad: "; timeout 120 $P/gemm_probe_r06 $shape 0 0 1 0 1 $pad | tail -1
      done
    done
  done ) > $o/${tag}_gemm_ab.txt 2>&1; cat $o/${tag}_gemm_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "score_topk or topk_rows or eval_session or eval_golden or users_rating or wide_gemm or randomised_stress or ncf_init_eval or ncf_train_golden or full_size" 2>&1 | tail -5 | tee $o/${tag}_tests.txt
timeout 300 bash scripts/eval_session_trace.sh 2>&1 | tail -12 > $o/${tag}_eval_session_trace.txt; cat $o/${tag}_eval_session_trace.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-config4 2>$o/${tag}_bench_s20.err | grep "^{" > $o/${tag}_bench_s20.json
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_s20.json")); r = d["roofline"]; t = d.get("topk") or {}
print("%.4g trip/s" % d["value"], "%.1f us/step" % (d["ms_per_step"] * 1e3), r["kernel"], "%.2f us frac %.3f" % (r["avg_launch_us"], r["frac"]),
      "topk %.1f us" % (t.get("seconds", 0) * 1e6), "parity", (d.get("parity") or {}).get("ok"), "mfma", d["mfma_gemm"]["frac"], "yelp eval", d["also"]["config3_yelp"]["topk"])
PY
