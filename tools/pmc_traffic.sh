#!/bin/bash
# HBM traffic of the GEMM lab launches: two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) + a utilisation pass.
#   tools/pmc_traffic.sh <out-name>      -> gpurun_out/<out-name>_{fetch,write,util}.txt (one line per gemm dispatch)
#   LAB_BIN / LAB_ARGS / LAB_KERNELS: another laboratory binary, its arguments, the kernel-name substring to keep
#   (bf16 kernels: LAB_BIN=bf16_lab LAB_ARGS=time LAB_KERNELS=bf16_kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "util:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM"; do
  tag=${pass%%:*}; ctr=${pass#*:}
  d=$R/gpurun_out/${name}_$tag
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $d --output-format csv -- $R/tools/${LAB_BIN:-gemm_lab_prod} ${LAB_ARGS:-nocheck} > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $d ${LAB_KERNELS:-gemm} > $R/gpurun_out/${name}_$tag.txt
  rm -rf $d $d.log
  wc -l $R/gpurun_out/${name}_$tag.txt
done
