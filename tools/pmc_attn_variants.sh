for v in base pair; do
  export V2S_LIB=$GRAFT_REPO_ROOT/tools/libvid2seq_hip_attn_$v.so
  bash tools/pmc_attn.sh > /dev/null 2>&1
  python tools/pmc_attn_summary.py gpurun_out > gpurun_out/r06_pmc_sq_attention_$v.txt 2>&1
  rm -rf gpurun_out/pmc_attn_a gpurun_out/pmc_attn_b
  grep -A4 "dkv_kernel" gpurun_out/r06_pmc_sq_attention_$v.txt | cut -c1-330
done
