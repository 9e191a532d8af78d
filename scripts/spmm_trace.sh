#!/bin/bash
# usage: spmm_trace.sh <graph> ; prints rocprof avg/min kernel ns for the spmm kernel under env settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_$$_$RANDOM
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/spmm_sweep.py $1 > /dev/null 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
echo "$1 dbg=${RK_SPMM_DEBUG:-0} var=${RK_SPMM_VARIANT:-0} seg=${RK_SEG_NNZ:-64} :: $(grep spmm_csr $f | cut -d, -f2,4,6 )"
rm -rf $out
