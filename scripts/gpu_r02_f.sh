#!/bin/bash
python -m pytest tests -m gpu -q -x -k "score_topk or fused_sweep or eval_golden or workflow" 2>&1 | tail -5
python3 scripts/score_probe.py 5893 3702 64 20 2>&1 | grep -v amdgpu
RK_SEL_CONFIG=2 python3 scripts/score_probe.py 5893 3702 64 20 2>&1 | grep -v amdgpu | head -1
python3 scripts/score_probe.py 54617 34474 128 3 2>&1 | grep -v amdgpu
RK_SEL_CONFIG=1 python3 scripts/score_probe.py 54617 34474 128 3 2>&1 | grep -v amdgpu | head -1
