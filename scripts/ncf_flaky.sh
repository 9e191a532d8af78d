#!/bin/bash
# repeat the NCF golden/oracle tests N times and count failing runs: scripts/ncf_flaky.sh [N]
n=${1:-14}; f=0
for i in $(seq 1 $n); do
  python3 -m pytest tests -m gpu -q -k "ncf" 2>&1 | grep -q "failed" && f=$((f+1))
done
echo "NCF tests: failing runs $f / $n"
