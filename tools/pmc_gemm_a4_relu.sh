# SQ counter pass: the FFN wi forward (32000 x 3072 x 768) on the persistent asm kernel with the plain, ReLU and ReLU + dropout epilogues, and on the
# kernel the dropout form replaced (gemm_a4_relu = 0); summary: tools/pmc_sq_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_r5relu
i=0
for spec in "|0|" "relu|0|" "relu|0.1|" "relu|0.1|gemm_a4_relu=0"; do
  act=$(echo "$spec" | cut -d'|' -f1); drop=$(echo "$spec" | cut -d'|' -f2); opts=$(echo "$spec" | cut -d'|' -f3)
  i=$((i+1))
  GEMM_ONE_ACT=$act GEMM_ONE_DROP=$drop V2S_OPTIONS=$opts timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_r5relu/pmc_32000_3072_768_v$i -o pmc --output-format csv -- python $R/tools/gemm_one.py 32000 3072 768 0 0 1 > $R/gpurun_out/pmc_r5relu/pmc_v$i.log 2>&1
  echo "== 32000 3072 768 act=[$act] dropout=$drop [$opts]"
done
python $R/tools/pmc_sq_summary.py $R/gpurun_out/pmc_r5relu
