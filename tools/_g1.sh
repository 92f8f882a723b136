mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r06_bench_default_c.log 2> gpurun_out/r06_bench_default_c.err
T1=$(date +%s)
echo "default bench wall seconds: $((T1-T0))" | tee gpurun_out/r06_bench_default_c.time
tail -c 1500 gpurun_out/r06_bench_default_c.log
