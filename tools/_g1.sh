mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_layers_native_gpu.py tests/test_backward_gpu.py tests/test_bf16_stream_gpu.py tests/test_arena_gpu.py tests/test_graphed_gpu.py tests/test_loss_curve_gpu.py tests/test_ddp_two_ranks_one_gpu.py -q 2>&1 | grep -v "visual target" | tail -15 > gpurun_out/r06_plan_cache_tests.txt
cat gpurun_out/r06_plan_cache_tests.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
for spec in "bf16 64" "bf16 256"; do
set -- $spec
VB_GEMM_MODE=$1 timeout 600 python tools/host_profile.py --batch $2 --steps 10 --top 25 > gpurun_out/r06_host_profile_$1_b$2_plans.txt 2>&1
head -6 gpurun_out/r06_host_profile_$1_b$2_plans.txt | tail -4
done
for b in 64 64; do
timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --gemm-mode bf16 --no-cpu-baseline --no-alt-mode --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
