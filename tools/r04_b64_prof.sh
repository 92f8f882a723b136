# kernel statistics of the batch-64 train step (BASELINE configs[2] per GPU), default planner and with the small-M menu
mkdir -p gpurun_out
VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0 bash tools/prof_step.sh r04_train_b64_single_stream --batch 64 > gpurun_out/r04_train_b64_top_kernels_single_stream.txt 2>&1
cd $GRAFT_REPO_ROOT
VB_GEMM_V4_SMALLM=1 VB_TWO_STREAMS=0 VB_WGRAD_STREAM=0 bash tools/prof_step.sh r04_train_b64_smallm_single_stream --batch 64 > gpurun_out/r04_train_b64_smallm_top_kernels_single_stream.txt 2>&1
cd $GRAFT_REPO_ROOT
head -12 gpurun_out/r04_train_b64_top_kernels_single_stream.txt; head -12 gpurun_out/r04_train_b64_smallm_top_kernels_single_stream.txt
