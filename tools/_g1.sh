mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_layers_native_gpu.py -q 2>&1 | grep -v "visual target" | tail -12
