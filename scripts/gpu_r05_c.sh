#!/bin/bash
# round 5: packed adds in the LDS gather (v_pk_add_f32): phase stamps, parity, bench
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "lightgcn or lds or spmm or golden" 2>&1 | tail -3
timeout 300 python scripts/spmm_lds_probe.py > $o/r05c_spmm_lds_probe.json 2> $o/r05c_spmm_lds_probe.err; tail -c 1500 $o/r05c_spmm_lds_probe.json
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also-sharded > $o/r05c_bench_s20_$i.json 2>/dev/null; python - <<PY
import json; d=json.loads(open("$o/r05c_bench_s20_$i.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"])
PY
done
