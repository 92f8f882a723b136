mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "visual target" | tail -30 > gpurun_out/r06_gpu_suite_e.txt
tail -6 gpurun_out/r06_gpu_suite_e.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/r06_bench_default_d.log 2> gpurun_out/r06_bench_default_d.err
tail -c 1200 gpurun_out/r06_bench_default_d.log
