#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "32 5" "256 3"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ne -- python3 scripts/ncf_eval_prof.py $cfg 2>&1 | grep users
  f=$(ls gpurun_out/prof_ne/*/*kernel_stats.csv | head -1); head -9 $f | cut -c1-150; rm -rf gpurun_out/prof_ne
done
