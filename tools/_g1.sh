mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
DDP_MIB=1024 VB_GEMM_MODE=bf16 timeout 900 python tools/ddp_overhead.py 2>&1 | grep -E "plain step|stand-alone" > gpurun_out/r06_ddp_one_rank_avg.txt
cat gpurun_out/r06_ddp_one_rank_avg.txt
