mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_bf16_stream_gpu.py tests/test_bf16_bench_shapes_gpu.py tests/test_layers_native_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r06_bf16_tests_fused.txt
cat gpurun_out/r06_bf16_tests_fused.txt
for f in 0 1; do
echo "== fused=$f bf16 b256 / b64"
VB_BF16_WG_FUSED=$f timeout 600 python bench.py --batch 256 --steps 10 --warmup 3 --gemm-mode bf16 --no-cpu-baseline --no-alt-mode --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
VB_BF16_WG_FUSED=$f timeout 600 python bench.py --batch 64 --steps 20 --warmup 5 --gemm-mode bf16 --no-cpu-baseline --no-alt-mode --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done > gpurun_out/r06_bf16_fused_bench_ab.txt 2>&1
cat gpurun_out/r06_bf16_fused_bench_ab.txt
