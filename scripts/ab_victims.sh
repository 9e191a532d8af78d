#!/bin/bash
# A/B of scripts/bench_victims.py between older checkouts built in-tree (ab_*/) and the current tree, same box
show() { python3 -c "
import json,sys
j=json.load(open(sys.argv[1]))
print(sys.argv[2], {k:(round(v['train_samples_per_s_body']), round(v['eval_users_per_s'])) for k,v in j.items()})" "$1" "$2"; }
for d in "$@"; do
  (cd $d && mkdir -p gpurun_out && python3 scripts/bench_victims.py >/dev/null 2>gpurun_out/err.txt; show gpurun_out/bench_victims.json $d || tail -3 gpurun_out/err.txt)
done
