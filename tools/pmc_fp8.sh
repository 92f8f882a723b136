#!/bin/bash
# rocprofv3 PMC passes (utilisation | FETCH_SIZE | WRITE_SIZE, separate runs) over tools/fp8_lab.py:
#   tools/pmc_fp8.sh <out-name>   -> gpurun_out/<out-name>.txt, one line per (kernel, grid) group of the lab's launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1
for pass in "util:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  d=$R/gpurun_out/${name}_$tag
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $d --output-format csv -- python $R/tools/fp8_lab.py > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $d fp8 > $R/gpurun_out/${name}_$tag.raw
  rm -rf $d $d.log
done
python3 - $R/gpurun_out/$name <<'PY' | tee $R/gpurun_out/$name.txt
import ast, re, sys
base = sys.argv[1]
def groups(tag):
    out, cur = [], None
    for l in open(base + "_" + tag + ".raw"):
        m = re.match(r"dispatch (\d+): (.*)", l)
        if not m:
            continue
        e = ast.literal_eval(m.group(2))
        key = (e["kernel"], e["grid"])
        if cur is None or cur[0] != key:
            cur = [key, []]
            out.append(cur)
        cur[1].append(e)
    return out
util, fetch, write = groups("util"), groups("fetch"), groups("write")
print("rocprofv3 --kernel-trace --pmc over tools/fp8_lab.py (shapes in lab order, 23 launches each, averaged); FETCH / WRITE from their own passes;")
print("gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM) - shown raw")
for i, (key, es) in enumerate(util):
    n = len(es)
    dur = sum(e.get("dur_us", 0) for e in es) / n
    cyc = sum(e.get("GRBM_GUI_ACTIVE", 0) for e in es) / n / 8
    mf = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for e in es) / n
    wi = sum(e.get("SQ_WAIT_INST_ANY", 0) for e in es) / max(sum(e.get("SQ_WAVE_CYCLES", 1) for e in es), 1)
    f = sum(e.get("FETCH_SIZE", 0) for e in fetch[i][1]) / max(len(fetch[i][1]), 1) if i < len(fetch) else 0
    w = sum(e.get("WRITE_SIZE", 0) for e in write[i][1]) / max(len(write[i][1]), 1) if i < len(write) else 0
    print("%-40s blocks %6d x %3d  %8.1f us  clk %.2f GHz  mfma_busy %.3f  wait_inst %.2f  FETCH %9.0f KB  WRITE %9.0f KB" % (
        key[0][:40], key[1], n, dur, cyc / dur / 1e3 if dur else 0, mf / (1024 * cyc) if cyc else 0, wi, f, w))
PY
rm -f $R/gpurun_out/${name}_*.raw
