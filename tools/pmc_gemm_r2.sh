# SQ counter pass over the round-2 GEMM kernels at the library's default dispatch: 32000x2304x768 NT (gemm_p8d_kernel), 8192^3 NT
# (gemm_p8_kernel), 32000x768x3072 NN (gemm_dma_kernel<false,true>, the step's dominant kernel); summary: tools/pmc_sq_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_r2
for cfg in "32000 2304 768 0 0 1" "8192 8192 8192 0 0 1" "32000 768 3072 0 1 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_r2/pmc_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $cfg > $R/gpurun_out/pmc_r2/pmc_$tag.log 2>&1
done
python $R/tools/pmc_sq_summary.py $R/gpurun_out/pmc_r2
