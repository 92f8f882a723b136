mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 tools/bf16_lab time > gpurun_out/r06_bf16_lab_time.txt 2>&1
tail -14 gpurun_out/r06_bf16_lab_time.txt
bash tools/pmc_r06.sh 2>&1 | tail -60
