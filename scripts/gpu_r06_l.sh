#!/bin/bash
# Round 6: GEMM + selection at the large shapes got slower between the round-5 and round-6 closing passes (16384 x 34474 x 64:
# 1.73 -> 2.31 ms).  Which kernel?  The round-5 library, this tree, and this tree with the round-5 score_key, one box, alternating,
# per-kernel durations by rocprofv3.
tag=r06l
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( for rep in 1 2; do for m in r05 new oldkey; do
    for shape in "16384 34474 64 4" "8192 34474 256 4"; do
      echo "== $m $shape"
      RECAD_TUNING_LIB=$PWD/ab_tune/librecad_hip_$m.so PROBE_MODES=unfused timeout 200 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
      out=$o/sp_$$; rm -rf $out
      RECAD_TUNING_LIB=$PWD/ab_tune/librecad_hip_$m.so PROBE_MODES=unfused timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 scripts/score_probe.py $shape > /dev/null 2>&1
      f=$(ls $out/*/*kernel_stats.csv | head -1); python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'gemm_f32' in r['Name'] or 'topk_' in r['Name']: print('   ', r['Name'][:48], r['Calls'], 'calls, avg %.1f us' % (float(r['AverageNs'])/1e3))"; rm -rf $out
    done
  done; done ) > $o/${tag}_unfused_ab.txt 2>&1; cat $o/${tag}_unfused_ab.txt
