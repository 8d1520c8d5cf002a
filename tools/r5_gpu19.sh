cd $GRAFT_REPO_ROOT
for tag in base sc1 nt sc0sc1; do
  lib=vidchapters_amd/libvid2seq_hip.so; [ $tag != base ] && lib=tools/libvid2seq_hip_st_$tag.so
  echo "== stores: $tag"
  V2S_LIB=$PWD/$lib timeout 300 python tools/gemm_a4_ab.py --quick 2>&1 | grep -v "amdgpu.ids\|^check\|correctness" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,"default",$10,"a4",$13,"nostore",$16,"vendor",$25}' 
done > gpurun_out/r05_a4_store_policy.txt 2>&1
cat gpurun_out/r05_a4_store_policy.txt
timeout 900 python tools/step_ab.py "lib=vidchapters_amd/libvid2seq_hip.so" "lib=tools/libvid2seq_hip_st_sc1.so" "lib=tools/libvid2seq_hip_st_nt.so" "lib=tools/libvid2seq_hip_st_sc0sc1.so" --steps 8 --block 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_store_policy.txt
