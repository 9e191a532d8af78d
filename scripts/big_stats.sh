cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in yelp config4; do
  o=gpurun_out/prof_big_$w; rm -rf $o
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python3 bench.py --workload $w --no-cpu-baseline --no-topk --steps 20 --warmup 5 > gpurun_out/r03_bench_${w}_profiled.json 2>/dev/null
  f=$(ls $o/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200 > gpurun_out/r03_${w}_kernel_stats.csv; rm -rf $o
  cat gpurun_out/r03_${w}_kernel_stats.csv | cut -c1-150
done
