mkdir -p gpurun_out
K="bert_base_2layer_2conect.json-False-ring"
for v in 1 2 3 4 5; do
env DBG_ALL=1 timeout 600 python -m pytest tests/test_ddp_two_ranks_one_gpu.py -x -q -s -k "$K" 2>&1 | grep -E "BAD|passed|failed|Error" | head -5
done > gpurun_out/r06_ddp2_dbg.txt 2>&1
cat gpurun_out/r06_ddp2_dbg.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06_gpu_suite_b.txt
cat gpurun_out/r06_gpu_suite_b.txt
