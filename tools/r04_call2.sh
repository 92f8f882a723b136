# round-4 GPU call 2: the rest of the GPU suite (call 1 stopped at the loss-curve test) + batch-64 experiments
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_loss_curve_gpu.py tests/test_model_parity_gpu.py tests/test_optim.py tests/test_random_shapes_gpu.py tests/test_task_forward_gpu.py tests/test_timed_config_gpu.py tests/test_wgrad_skinny_gpu.py tests/test_input_pipeline.py -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r04_gpu_suite_b.txt
B="python bench.py --batch 64 --steps 20 --warmup 5 --no-alt-mode --no-cpu-baseline --no-extra-legs"
run() { name=$1; shift; ( env "$@" timeout 200 $B $EXTRA 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], d['roofline']['achieved'])" ) >> gpurun_out/r04_bench_b64_menu_ab.txt 2>&1; }
rm -f gpurun_out/r04_bench_b64_menu_ab.txt
EXTRA=""
run eager_default A=1
run eager_smallm VB_GEMM_V4_SMALLM=1
run eager_smallm_single_stream VB_GEMM_V4_SMALLM=1 VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0
run eager_single_stream VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0
EXTRA="--graph"
run graph_q4 GPU_MAX_HW_QUEUES=4
run graph_q4_smallm GPU_MAX_HW_QUEUES=4 VB_GEMM_V4_SMALLM=1
run graph_q4_smallm_single VB_TWO_STREAMS=0 GPU_MAX_HW_QUEUES=4 VB_GEMM_V4_SMALLM=1
tail -3 gpurun_out/r04_gpu_suite_b.txt; cat gpurun_out/r04_bench_b64_menu_ab.txt
