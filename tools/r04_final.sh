# round-4 closing GPU call: whole GPU suite, default bench line, single-stream kernel statistics + per-shape GEMM times
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r04_gpu_suite_final.txt
timeout 500 python bench.py > gpurun_out/r04_bench_default_b.log 2> gpurun_out/r04_bench_default_b.err
VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0 bash tools/prof_step.sh r04_train_b256_single_stream > gpurun_out/r04_train_b256_top_kernels_single_stream.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/r04_gpu_suite_final.txt; cut -c1-400 gpurun_out/r04_bench_default_b.log | tail -2
