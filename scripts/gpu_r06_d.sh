#!/bin/bash
# Round 6: wide GEMM grouped last chunk, A/B with the score matrix's rows padded to 128-byte lines (what the library does) and not.
tag=r06d
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=recad_amd/lib/probes
( for rep in 1 2 3; do
    for shape in "5893 3702 64" "8192 34474 256" "54617 34474 128" "16384 34474 64" "8192 6144 64" "8192 2000 64"; do
      for pad in 1 0; do
        echo -n "r05 ldpad=$pad: "; timeout 120 $P/gemm_probe_r05 $shape 0 0 1 0 1 $pad | tail -1
        echo -n "r06 ldpad=$pad: "; timeout 120 $P/gemm_probe_r06 $shape 0 0 1 0 1 $pad | tail -1
      done
    done
  done
  timeout 120 $P/gemm_probe_r06_stamps 5893 3702 64 0 0 1 0 1 1 ) > $o/${tag}_gemm_ab.txt 2>&1; cat $o/${tag}_gemm_ab.txt
