# round-4 closing call 2: default bench line with the final bench.py + PMC passes over the MX GEMM laboratory
mkdir -p gpurun_out
timeout 500 python bench.py > gpurun_out/r04_bench_default_c.log 2> gpurun_out/r04_bench_default_c.err
echo "bench rc=$?"
bash tools/pmc_mx.sh r04_mx_gemm_pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cut -c1-300 gpurun_out/r04_bench_default_c.log | tail -1; tail -2 gpurun_out/r04_bench_default_c.err; cat gpurun_out/r04_mx_gemm_pmc.txt
