#!/bin/bash
# Round 4: persistent-kernel tile menu against the 4-wave blocks, in-process A/B on the PRODUCT library
# (tools/gemm_lab_prod LAB_V4_AB=1: alternating 20-launch blocks, best of 3), forward + dgrad.
#   usage: tools/lab_v4_menu.sh <out-prefix>
out=gpurun_out/${1:-r04_gemm_lab_v4_menu}
export LAB_V4_AB=1 LAB_NOWGRAD=1
run() {  # name, M, shapes, cfg list
  local f=${out}_$1.txt; : > $f
  for cfg in $4; do
    echo "=== M=$2 VB_GEMM_V4_CFG=$cfg" >> $f
    if [ "$cfg" = plan ]; then env LAB_M=$2 LAB_SHAPES="$3" timeout 120 tools/gemm_lab_prod >> $f 2>&1
    else env VB_GEMM_V4_CFG=$cfg LAB_M=$2 LAB_SHAPES="$3" timeout 120 tools/gemm_lab_prod >> $f 2>&1; fi
  done
  grep -E "^===|A/B|err [0-9.e+-]*$" $f | grep -E "^===|A/B" 
}
run M9472 9472 "1024,1024,1;1024,1024,3;1024,2048,1" "plan"
run M2304 2304 "768,768,1;768,768,3;3072,768,1;768,3072,1;1024,768,3;768,1024,1" "plan 6204 6203 6104 6103 4203 4202 4104"
run M2368 2368 "1024,1024,1;1024,1024,3;1024,2048,1;768,1024,3" "plan 6204 6104 4203 4202 4104"
run M9216 9216 "768,768,1;3072,768,1;768,3072,1;1024,768,3" "plan"
