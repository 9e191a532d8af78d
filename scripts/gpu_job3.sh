cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $o/r04f_tests.txt; cat $o/r04f_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python scripts/bench_victims.py 2>&1 | grep -v amdgpu.ids | cut -c1-700; cp $o/bench_victims.json $o/r04_bench_victims.json
timeout 300 bash scripts/ncf_step_trace.sh 256 5 > $o/r04_ncf_f256_l5_step_trace.txt 2>&1; cat $o/r04_ncf_f256_l5_step_trace.txt | cut -c1-150
( for shape in "5893 3702 64 20" "8192 34474 256 5" "54617 34474 128 3"; do PROBE_MODES=panel,unfused timeout 300 python3 scripts/score_probe.py $shape 2>/dev/null | grep -v amdgpu.ids; done ) > $o/r04_score_probe.txt; cat $o/r04_score_probe.txt
