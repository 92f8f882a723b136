set -x
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r04_gpu_suite_a.txt
timeout 400 python bench.py > gpurun_out/r04_bench_default_a.log 2> gpurun_out/r04_bench_default_a.err
VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0 bash tools/prof_step.sh r04_train_b256_single_stream > gpurun_out/r04_train_b256_top_kernels_single_stream.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 6 --warmup 2 --no-alt-mode --no-cpu-baseline --no-extra-legs --gemm-breakdown > gpurun_out/r04_bench_train_b256_gemm_breakdown.txt 2>&1
timeout 300 python bench.py --batch 64 --steps 10 --warmup 3 --no-alt-mode --no-cpu-baseline --no-extra-legs --gemm-breakdown > gpurun_out/r04_bench_b64_gemm_breakdown.txt 2>&1
tail -3 gpurun_out/r04_gpu_suite_a.txt; cut -c1-600 gpurun_out/r04_bench_default_a.log
