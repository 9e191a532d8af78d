cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "marked_block or lightgcn_train_golden or timed_path or vs_oracle_shapes" 2>&1 | tail -3
timeout 300 bash scripts/step_trace.sh --workload yelp --dim 128 2>&1 | tail -3 | tee gpurun_out/r05_step_trace_yelp.txt
timeout 400 bash scripts/step_trace.sh --workload config4 --steps 6 --warmup 2 2>&1 | tail -3 | tee gpurun_out/r05_step_trace_config4.txt
