cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  n=$(echo $pass | cut -c4-12)
  GEMM_BENCH_QUICK=1 VB_GEMM_MODE=${PMC_GEMM_MODE:-bf16x6} timeout 300 rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmc_${PMC_GEMM_MODE:-bf16x6}_$n --output-format csv -- python $R/tools/gemm_bench.py > $R/gpurun_out/pmc_${PMC_GEMM_MODE:-bf16x6}_$n.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${PMC_GEMM_MODE:-bf16x6}_$n gemm | tail -4
done
