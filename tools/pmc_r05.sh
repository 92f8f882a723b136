#!/bin/bash
# Round-5 counter evidence (review item 6): FETCH_SIZE / WRITE_SIZE / utilisation passes (separate rocprofv3 runs) over
#   * the fp32 persistent kernels at the timed shapes: text stream 9216 x 3072 x 768 (gemm_v4_kernel<6,3,0,4>, gemm_v4w_kernel) and
#     the 37-region image stream 9472 x 1024 x 1024 (the mixed 320 | 256-row launch gemm_v4_kernel<4,5,4,4>);
#   * the bf16 training kernels at the step's shapes (tools/bf16_lab time).
# Tables: python tools/pmc_r03.py gpurun_out/<name>
R=$GRAFT_REPO_ROOT
LAB_WARM=3 LAB_M=9216 LAB_SHAPES="3072,768,1;768,768,1" bash $R/tools/pmc_traffic.sh r05_pmc_f32_text
LAB_WARM=3 LAB_M=9472 LAB_SHAPES="1024,1024,1;1024,1024,3" bash $R/tools/pmc_traffic.sh r05_pmc_f32_image
LAB_BIN=bf16_lab LAB_ARGS=time LAB_KERNELS=bf16_kernel bash $R/tools/pmc_traffic.sh r05_pmc_bf16
for n in r05_pmc_f32_text r05_pmc_f32_image r05_pmc_bf16; do python3 $R/tools/pmc_r03.py $R/gpurun_out/$n > $R/gpurun_out/$n.table.txt; done
