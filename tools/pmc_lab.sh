#!/bin/bash
# rocprofv3 PMC pass over the GEMM lab (quick shape list) for each configuration given as argument:
#   tools/pmc_lab.sh <out-name> "VB_GEMM_TILE=33" "VB_GEMM_TILE=33 VB_GEMM_ABL=1" ...
# prints, per gemm dispatch: duration, effective shader clock (GRBM_GUI_ACTIVE / 8 / duration), matrix-pipe busy
# fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles))
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
out=$R/gpurun_out/$name.txt
: > $out
i=0
for cfg in "$@"; do
  i=$((i+1))
  d=$R/gpurun_out/${name}_$i
  env $cfg timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM -d $d --output-format csv -- $R/tools/gemm_lab quick nocheck > $d.log 2>&1
  echo "=== $cfg" >> $out
  python3 $R/tools/pmc_summary.py $d gemm | python3 -c "
import sys,ast,re
for l in sys.stdin:
    m=re.match(r'dispatch (\d+): (.*)',l)
    e=ast.literal_eval(m.group(2))
    if not e.get('dur_us'): continue
    cyc=e.get('GRBM_GUI_ACTIVE',0)/8
    print('%-46s blocks %5d  %8.1f us  clk %.2f GHz  mfma_busy %.3f  wait_inst %.2f wait_any %.2f' % (e['kernel'][:46], e['grid'], e['dur_us'], cyc/e['dur_us']/1e3, e.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*cyc) if cyc else 0, e.get('SQ_WAIT_INST_ANY',0)/max(e.get('SQ_WAVE_CYCLES',1),1), e.get('SQ_WAIT_ANY',0)/max(e.get('SQ_WAVE_CYCLES',1),1)))
" | awk 'NR%23==4 || NR%23==5 || NR%23==6 || NR%23==12 || NR%23==13 {print}' >> $out
  rm -rf $d
done
cat $out
