# SQ counter passes over the encoder-layer attention kernels (tools/attn_one.py); summary: tools/pmc_attn_summary.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/pmc_attn_a -o pmc --output-format csv -- python $R/tools/attn_one.py ${ATTN_N:-1000} > $R/gpurun_out/pmc_attn_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --kernel-trace -d $R/gpurun_out/pmc_attn_b -o pmc --output-format csv -- python $R/tools/attn_one.py ${ATTN_N:-1000} > $R/gpurun_out/pmc_attn_b.log 2>&1
ls $R/gpurun_out/pmc_attn_a $R/gpurun_out/pmc_attn_b
