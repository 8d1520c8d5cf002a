cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_abl
i=0
for d in 2 11 12 13 14; do
  i=$((i+1))
  V2S_LIB=$R/tools/libvid2seq_hip_abl.so V2S_OPTIONS=gemm_a4=3,gemm_dbg=$d timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_abl/pmc_8192_8192_8192_v$i -o pmc --output-format csv -- python $R/tools/gemm_one.py 8192 8192 8192 0 0 1 > $R/gpurun_out/pmc_abl/log_$i.txt 2>&1
  echo "== dbg $d"
done
python $R/tools/pmc_sq_summary.py $R/gpurun_out/pmc_abl
