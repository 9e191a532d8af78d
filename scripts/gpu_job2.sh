cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid2d or capture_failure or sharded" 2>&1 | tail -4
timeout 300 python bench.py --gpus 2 --backend gloo --workload yelp --parallel rows2d --steps 4 --warmup 1 --no-topk 2>/dev/null | grep "^{" > $o/r04_rows2d_gloo_yelp.json; python3 -c "
import json; d=json.load(open('$o/r04_rows2d_gloo_yelp.json')); print({k: d[k] for k in ('value','ms_per_step','n_gpus','same_workload_1gpu','same_workload_replicas','rows_comm_model')}, d['config']['parallelism'])"
timeout 600 python scripts/shard_probe.py config4 64 5 4 2>/dev/null | grep -v amdgpu.ids > $o/r04_shard_probe_config4.txt; cat $o/r04_shard_probe_config4.txt | cut -c1-600
timeout 300 python scripts/shard_probe.py yelp 128 20 2 2>/dev/null | grep -v amdgpu.ids | grep -v "^{" > $o/r04_shard_probe_yelp.txt; cat $o/r04_shard_probe_yelp.txt
