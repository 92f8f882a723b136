mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_loss_curve_gpu.py -q -x -s 2>&1 | grep -v "visual target" | tail -40 > gpurun_out/r06_loss_curve_fail.txt
cat gpurun_out/r06_loss_curve_fail.txt
