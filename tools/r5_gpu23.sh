cd $GRAFT_REPO_ROOT
for tag in base z3 z5; do
  lib=vidchapters_amd/libvid2seq_hip.so; [ $tag != base ] && lib=tools/libvid2seq_hip_$tag.so
  echo "== z loads: $tag"
  V2S_LIB=$PWD/$lib timeout 300 python tools/gemm_a4_dact_ab.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05_a4p_dact_zspread_ab.txt
timeout 900 python tools/step_ab.py "lib=vidchapters_amd/libvid2seq_hip.so,gemm_a4=1" "lib=vidchapters_amd/libvid2seq_hip.so,gemm_a4=5" "lib=tools/libvid2seq_hip_z3.so,gemm_a4=5" "lib=tools/libvid2seq_hip_z5.so,gemm_a4=5" --steps 8 --block 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_dact_zspread.txt
