#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_det -- python3 bench.py --no-cpu-baseline --deterministic --no-topk > /dev/null 2>&1
f=$(ls gpurun_out/prof_det/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-150; rm -rf gpurun_out/prof_det
