mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_chunks.py tests/test_kernels_gpu.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_attn_long_a.txt
cat gpurun_out/r06_attn_long_a.txt
B=64 VB_GEMM_MODE=bf16 timeout 300 python tools/aten_census.py 2>&1 | head -60 | cut -c1-200 > gpurun_out/r06_aten_census_bf16_b64.txt
cat gpurun_out/r06_aten_census_bf16_b64.txt
