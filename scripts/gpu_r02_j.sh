#!/bin/bash
for tpb in 1 2 4 8 16 32; do echo -n "tpb=$tpb "; RK_GEMM_TPB=$tpb RK_SEL_OFF=1 python3 scripts/score_probe.py 8192 34474 256 5 2>&1 | grep "^unfused" ; done
echo -n "variant1 "; RK_GEMM_VARIANT=1 python3 scripts/score_probe.py 8192 34474 256 5 2>&1 | grep "^unfused"
