mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
( echo "=== new (two-stage LDS for head_dim 128)"; python tools/attn_bench.py --check 2>&1 | grep -v "visual target" | tail -12; python tools/attn_bench.py 2>&1 | tail -10
  echo "=== old"; python tools/attn_bench.py --lib vilbert-multi-task_amd/csrc/libvilbert_hip_oldattn.so 2>&1 | tail -10 ) > gpurun_out/r06_attn_two_stage_ab.txt 2>&1
cat gpurun_out/r06_attn_two_stage_ab.txt
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_backward_gpu.py -q -k "attention or attn" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_bf16_stream_gpu.py -q -k "attention" 2>&1 | tail -3
