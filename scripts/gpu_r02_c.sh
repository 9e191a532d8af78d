#!/bin/bash
for w in "yelp 128 50" "c4s 64 5" "config4 64 3"; do
  set -- $w
  echo "=== default lib: $w"; timeout 600 python3 scripts/spmm_panel_probe.py $1 $2 $3 2>&1 | grep -v amdgpu.ids
  echo "=== nt lib: $w"; RECAD_HIP_LIB=$PWD/ab_nt/lib/librecad_hip.so timeout 600 python3 scripts/spmm_panel_probe.py $1 $2 $3 2>&1 | grep -v amdgpu.ids | head -4
done
