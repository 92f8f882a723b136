#!/bin/bash
# runs the GEMM lab under a list of kernel configurations (arguments after the output name, default set below)
out=gpurun_out/${1:-lab}.txt
shift
: > $out
if [ $# -eq 0 ]; then set -- "VB_GEMM_V2=0" "VB_GEMM_V2=1" "VB_GEMM_TILE=33" "VB_GEMM_TILE=34" "VB_GEMM_TILE=44" "VB_GEMM_ABL=1"; fi
for cfg in "$@"; do
  echo "=== $cfg" >> $out
  env $cfg timeout 120 tools/gemm_lab ${LAB_ARGS:-} >> $out 2>&1
done
cat $out
