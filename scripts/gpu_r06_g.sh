#!/bin/bash
# Round 6: topk_wave_kernel with unclamped keys / OR exclusion / add-with-carry counting / byte-address candidate stores:
# selection tests (bit-exact vs oracle, NaN rows, tie-heavy, allocation end), stress sweep, eval trace
tag=r06g
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "score_topk or topk_rows or eval_session or eval_golden or randomised_stress or ncf_init_eval or device_eval or full_size" 2>&1 | tail -5 | tee $o/${tag}_tests.txt
timeout 300 bash scripts/eval_session_trace.sh 2>&1 | tail -12 > $o/${tag}_eval_session_trace.txt; cat $o/${tag}_eval_session_trace.txt
