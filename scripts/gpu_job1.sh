cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
( for shape in "2048 300000 64 3" "1024 500000 64 3" "3000 262144 128 3" "512 1000000 32 2" "4000 300000 64 3" "256 500000 64 3"; do
    PROBE_MODES=panel,fused,unfused timeout 200 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids
  done ) > $o/r04_score_corner.txt 2>&1; cat $o/r04_score_corner.txt
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $o/r04e_bench_s20.json ) 2>&1 | grep real
python3 -c "
import json; d=json.load(open('$o/r04e_bench_s20.json')); print(d['value'], d['ms_per_step'], d['parity'], json.dumps(d['also'])[:1500])"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $o/r04e_tests.txt; cat $o/r04e_tests.txt
