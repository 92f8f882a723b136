#!/bin/bash
# rocprofv3 PMC passes over the MX GEMM laboratory (tools/mx_lab time: the encoder's forward shapes at batch 512):
# utilisation, FETCH_SIZE, WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md), one line per shape =
# the LAST dispatch of each timing loop (23 launches per shape: 3 warm-up + 20 timed).
#   tools/pmc_mx.sh <out-name>   -> gpurun_out/<out-name>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=${1:-r04_mx_gemm_pmc}
for pass in "util:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  d=$R/gpurun_out/${name}_$tag
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d $d --output-format csv -- $R/tools/mx_lab time > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $d gemm_mx > $R/gpurun_out/${name}_$tag.raw
  rm -rf $d
done
python3 - $R/gpurun_out/${name} <<'PY' > $R/gpurun_out/${name}.txt
import ast, re, sys
base = sys.argv[1]
def load(tag):
    out = []
    for l in open(base + "_" + tag + ".raw"):
        m = re.match(r"dispatch (\d+): (.*)", l)
        out.append(ast.literal_eval(m.group(2)))
    return out
util, fetch, write = load("util"), load("fetch"), load("write")
shapes = ["2304x768 fp32", "2304x768 bf16", "768x768 fp32+res", "3072x768 GELU+MX", "3072x768 fp32", "768x3072 fp32+res",
          "3072x1024 fp32", "1024x1024 fp32+res", "1024x1024 GELU+MX", "1024x2048 fp32"]
print("# M = 18432; per launch: duration under the counter pass, effective clock, matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles),")
print("# FETCH_SIZE x 2 (gfx950 read correction, MI355X_MICROARCH.md) and WRITE_SIZE in MB (rocprofv3 reports KB)")
for i, s in enumerate(shapes):
    k = 23 * i + 22
    if k >= len(util): break
    e = util[k]
    cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
    f = fetch[k].get("FETCH_SIZE", 0) * 2 / 1024 if k < len(fetch) else float("nan")
    w = write[k].get("WRITE_SIZE", 0) / 1024 if k < len(write) else float("nan")
    print("%-22s %-44s %7.1f us  clk %.2f GHz  mfma_busy %.3f  valu insts %.2e  fetch %7.1f MB  write %7.1f MB" % (
        s, e["kernel"][:44], e["dur_us"], cyc / e["dur_us"] / 1e3 if e["dur_us"] else 0,
        e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0, e.get("SQ_INSTS_VALU", 0), f, w))
PY
rm -f $R/gpurun_out/${name}_*.raw $R/gpurun_out/${name}_*.log
cat $R/gpurun_out/${name}.txt
