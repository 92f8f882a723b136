#!/bin/bash
# per-block timeline of the GEMM launches (tools/gemm_lab LAB_TIMELINE=1, library built with LAB=1) under a list of configurations
out=gpurun_out/${1:-lab_timeline}.txt
shift
: > $out
if [ $# -eq 0 ]; then set -- "X=0" "VB_GEMM_ABL=1" "VB_GEMM_ABL=5" "VB_GEMM_ABL=7" "VB_GEMM_TILE=44" "LAB_M=2368"; fi
for cfg in "$@"; do
  echo "=== $cfg" >> $out
  env $cfg LAB_TIMELINE=1 timeout 120 tools/gemm_lab quick nocheck >> $out 2>&1
done
cat $out
