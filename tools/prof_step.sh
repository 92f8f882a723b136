#!/bin/bash
# rocprofv3 kernel-trace statistics of the bench step; writes gpurun_out/<name>_kernel_stats.csv
#   tools/prof_step.sh <name> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
d=$R/gpurun_out/${name}_prof
rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-alt-mode --no-cpu-baseline --no-extra-legs "$@" > $R/gpurun_out/${name}_prof.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${name}_kernel_stats.csv
rm -rf $d
python3 - $R/gpurun_out/${name}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:28]:
    print("%6.2f%% %7d calls %9.1f us avg  %s" % (100*float(r["TotalDurationNs"])/tot, int(r["Calls"]), float(r["AverageNs"])/1e3, r["Name"][:110]))
PY
tail -1 $R/gpurun_out/${name}_prof.log | cut -c1-200
