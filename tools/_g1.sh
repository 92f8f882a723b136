mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_layers_native_gpu.py tests/test_arena_gpu.py -q 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do for v in each batch; do
for spec in "bf16 256" "bf16 64" "f32 256" "f32 64"; do set -- $spec
echo -n "VB_WGRAD_FORK=$v $1 b$2: "
VB_WGRAD_FORK=$v timeout 600 python bench.py --batch $2 --steps 20 --warmup 5 --gemm-mode $1 --no-cpu-baseline --no-alt-mode --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done > gpurun_out/r06_wgrad_fork_batching_ab.txt 2>&1
cat gpurun_out/r06_wgrad_fork_batching_ab.txt
