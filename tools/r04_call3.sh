# round-4 GPU call 3: MX path tests + forward A/B (row-scaled fp8 vs MX) + kernel statistics of the MX forward
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_mx_gpu.py -q 2>&1 | tail -30 ) > gpurun_out/r04_mx_tests_a.txt
F="python bench.py --mode fwd --batch 512 --steps 10 --warmup 3 --no-alt-mode --no-cpu-baseline --no-extra-legs"
for m in mxfp8; do
  ( timeout 300 $F --gemm-mode $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])" ) >> gpurun_out/r04_fwd_b512_fp8_vs_mx.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in mxfp8; do
  d=$R/gpurun_out/${m}_prof; rm -rf $d
  timeout 400 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python $R/bench.py --mode fwd --batch 512 --steps 6 --warmup 2 --no-alt-mode --no-cpu-baseline --no-extra-legs --gemm-mode $m > $R/gpurun_out/r04_${m}_fwd_b512_prof.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r04_${m}_fwd_b512_kernel_stats.csv; rm -rf $d
done
cd $R
python3 - gpurun_out/r04_mxfp8_fwd_b512_kernel_stats.csv <<'PY' > gpurun_out/r04_mxfp8_fwd_b512_top_kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:24]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
tail -12 gpurun_out/r04_mx_tests_a.txt; cat gpurun_out/r04_fwd_b512_fp8_vs_mx.txt; cat gpurun_out/r04_mxfp8_fwd_b512_top_kernels.txt
