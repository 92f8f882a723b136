# kernel statistics of the MX forward (batch 512) with the final binaries
mkdir -p gpurun_out
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
print("total GPU time %.1f ms (11 forwards)" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:24]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
head -14 gpurun_out/r04_mxfp8_fwd_b512_top_kernels.txt
