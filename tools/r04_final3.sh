# round-4 closing call 3: tests touched after the full-suite run + smoke + the fp32 tile-menu A/B the planner's comments cite
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_attention_chunks.py tests/test_mx_gpu.py tests/test_kernels_gpu.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04_gpu_tests_after_final.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) >> gpurun_out/r04_gpu_tests_after_final.txt
timeout 400 bash tools/lab_v4_menu.sh r04_gemm_lab_v4_menu > gpurun_out/r04_gemm_lab_v4_menu_summary.txt 2>&1
tail -8 gpurun_out/r04_gpu_tests_after_final.txt; tail -30 gpurun_out/r04_gemm_lab_v4_menu_summary.txt
