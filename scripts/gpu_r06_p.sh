#!/bin/bash
# Round 6: compact-gradient backward on the LDS path (this tree) against the dense-gprop / float-atomics form (ab_old = the same tree
# with -DLDS_COMPACT_GRAD=0): LightGCN tests, then the step through bench.py alternating on one box, then the step by launch.
tag=r06p
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "lightgcn or workflow or smoke or eval_session" 2>&1 | tail -6 > $o/${tag}_tests.txt; cat $o/${tag}_tests.txt
( bash scripts/ab_step.sh . ab_old ) > $o/${tag}_compact_ab.txt 2>&1; cat $o/${tag}_compact_ab.txt
timeout 300 bash scripts/step_trace.sh --no-live-traffic > $o/${tag}_step_trace.txt 2>&1; cat $o/${tag}_step_trace.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" > $o/${tag}_bench_s20.json
python3 - <<PY
import json
d = json.load(open("$o/${tag}_bench_s20.json")); print("s20 %.1f us/step" % (d["ms_per_step"] * 1e3), "parity", d["parity"])
PY
