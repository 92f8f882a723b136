set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for s in S5 S6 S5 S6; do timeout 300 python tools/dbg_graph_nan.py $s 2>&1 | tail -3; done > gpurun_out/r06_dbg_nan_repro.txt 2>&1
for a in "--prior f32" "--prior f32 --poison" "--prior none --poison" "--prior f32 --taps" "--prior none --taps" "--prior fp8" "--prior fp8 --poison"; do
  echo "=== $a"; timeout 300 python tools/graph_audit.py $a 2>&1 | tail -12; done > gpurun_out/r06_graph_audit_a.txt 2>&1
timeout 1500 python -m pytest tests/test_bf16_bench_shapes_gpu.py -x -q -s 2>&1 | tail -60 > gpurun_out/r06_bf16_bench_shapes_a.txt
timeout 900 python bench.py > gpurun_out/r06_bench_default_start.log 2>&1
tail -c 3000 gpurun_out/r06_bench_default_start.log
