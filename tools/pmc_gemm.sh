cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "32000 2304 768 0 0 1" "32000 2304 768 0 0 3" "8192 8192 8192 0 0 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $cfg > $R/gpurun_out/pmc_$tag.log 2>&1
done
ls -R $R/gpurun_out/pmc_* | head -30
