set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/memset_node_repro.py > gpurun_out/r06_memset_node_repro.txt 2>&1
for s in S5 S6 S1 S5 S6 S1 S5 S6; do timeout 300 python tools/dbg_graph_nan.py $s 2>&1 | grep "^S"; done > gpurun_out/r06_dbg_nan_after_fix.txt 2>&1
for a in "--prior f32" "--prior fp8" "--prior f32 --taps" "--prior fp8 --poison" "--prior f32 --mode fp8+bf16" "--prior fp8 --mode f32"; do
  echo "=== $a"; timeout 300 python tools/graph_audit.py $a 2>&1 | grep -E "^graphed|^eager|^RESULT|^pointer"; done > gpurun_out/r06_graph_audit_b.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r06_gpu_suite_a.txt
