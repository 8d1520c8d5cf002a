# Runtime / build knobs of the launch path, A/B on one box: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host-coherent
# memory) and the -amdgpu-kernarg-preload-count=16 build (tools/libvid2seq_hip_preload.so).  Separate processes per arm, arms repeated and interleaved.
# usage: gpurun -- bash tools/env_ab.sh
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for arm in unset k0 k1 preload; do
  case $arm in
    unset)   unset HIP_FORCE_DEV_KERNARG; unset V2S_LIB;;
    k0)      export HIP_FORCE_DEV_KERNARG=0; unset V2S_LIB;;
    k1)      export HIP_FORCE_DEV_KERNARG=1; unset V2S_LIB;;
    preload) unset HIP_FORCE_DEV_KERNARG; export V2S_LIB=$PWD/tools/libvid2seq_hip_preload.so;;
  esac
  echo "== arm $arm rep $rep"
  ROUNDS=5 timeout 300 python tools/decode_ab.py gemm_skinny=1 2>&1 | tail -2
  timeout 300 python tools/step_ab.py "gemm_p8=1" --steps 6 --block 4 2>&1 | tail -1
done
done
