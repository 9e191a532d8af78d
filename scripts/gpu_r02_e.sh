#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/score_probe.py 5893 3702 64 20 2>&1 | grep -v amdgpu
RK_SEL_CONFIG=2 python3 scripts/score_probe.py 5893 3702 64 20 2>&1 | grep -v amdgpu | head -1
python3 scripts/score_probe.py 54617 34474 128 3 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sel -- python3 scripts/score_probe.py 5893 3702 64 20 > /dev/null 2>&1
f=$(ls gpurun_out/prof_sel/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200; rm -rf gpurun_out/prof_sel
