mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_ddp_two_ranks_one_gpu.py tests/test_arena_gpu.py -q 2>&1 | grep -v "visual target" | grep -E "passed|failed|Error" | tail -3
KINDS=plain,normal,delay VB_GEMM_MODE=bf16 timeout 900 python tools/ddp_overhead2.py 2>&1 | grep -E "wall"
