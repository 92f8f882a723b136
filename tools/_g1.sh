mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_modes_gpu.py -q -x 2>&1 | grep -v "visual target" | tail -12 > gpurun_out/r06_splitk_dgrad_tests.txt
cat gpurun_out/r06_splitk_dgrad_tests.txt
for b in 256 64; do
timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --gemm-mode bf16 --no-cpu-baseline --no-alt-mode --no-extra-legs --gemm-breakdown 2>gpurun_out/_bd.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
grep "30522" gpurun_out/_bd.txt
done > gpurun_out/r06_bf16_splitk_dgrad_bench.txt 2>&1
cat gpurun_out/r06_bf16_splitk_dgrad_bench.txt
