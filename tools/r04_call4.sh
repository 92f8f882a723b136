# round-4 GPU call: MX lab check + time, MX tests, forward bench (mxfp8), kernel statistics
mkdir -p gpurun_out
( timeout 120 tools/mx_lab all 2>&1 | grep -v "out (" | tail -14 ) > gpurun_out/r04_mx_lab_f.txt
( timeout 900 python -m pytest tests/test_mx_gpu.py -q 2>&1 | tail -30 ) > gpurun_out/r04_mx_tests_a.txt
F="python bench.py --mode fwd --batch 512 --steps 10 --warmup 3 --no-alt-mode --no-cpu-baseline --no-extra-legs"
rm -f gpurun_out/r04_fwd_b512_fp8_vs_mx.txt
for m in mxfp8; do
  ( timeout 300 $F --gemm-mode $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])" ) >> gpurun_out/r04_fwd_b512_fp8_vs_mx.txt 2>&1
done
( VB_MX_STREAM=f32 timeout 300 $F --gemm-mode mxfp8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mxfp8_f32_stream', d['value'], d['ms_per_step'])" ) >> gpurun_out/r04_fwd_b512_fp8_vs_mx.txt 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/mx_prof; rm -rf $d
timeout 300 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python $R/bench.py --mode fwd --batch 512 --steps 6 --warmup 2 --no-alt-mode --no-cpu-baseline --no-extra-legs --gemm-mode mxfp8 > $R/gpurun_out/r04_mxfp8_fwd_b512_prof.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r04_mxfp8_fwd_b512_kernel_stats.csv; rm -rf $d
cd $R
python3 - gpurun_out/r04_mxfp8_fwd_b512_kernel_stats.csv <<'PY' > gpurun_out/r04_mxfp8_fwd_b512_top_kernels.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:24]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
cat gpurun_out/r04_mx_lab_f.txt; tail -12 gpurun_out/r04_mx_tests_a.txt; cat gpurun_out/r04_fwd_b512_fp8_vs_mx.txt; head -16 gpurun_out/r04_mxfp8_fwd_b512_top_kernels.txt
