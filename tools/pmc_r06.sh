#!/bin/bash
# Round-6 counter evidence: FETCH_SIZE / WRITE_SIZE / utilisation passes (separate rocprofv3 runs) over
#   * the bf16 training kernels at the step's shapes (tools/bf16_lab time; weight gradient in its default deterministic form:
#     wgrad_bf16_kernel + wgrad_bf16_reduce_kernel);
#   * the MX GEMMs at the forward's shapes (tools/mx_lab time), refreshed after the round-5 changes of mx8.hip.
# Tables: gpurun_out/r06_pmc_bf16.table.txt, gpurun_out/r06_mx_gemm_pmc.txt
R=$GRAFT_REPO_ROOT
LAB_BIN=bf16_lab LAB_ARGS=time LAB_KERNELS=bf16 bash $R/tools/pmc_traffic.sh r06_pmc_bf16
python3 $R/tools/pmc_r03.py $R/gpurun_out/r06_pmc_bf16 > $R/gpurun_out/r06_pmc_bf16.table.txt
cat $R/gpurun_out/r06_pmc_bf16.table.txt
bash $R/tools/pmc_mx.sh r06_mx_gemm_pmc
