cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py "gemm_a4=1,gemm_a4_relu=1" "gemm_a4=5,gemm_a4_relu=1" "gemm_a4=1,gemm_a4_relu=0" "gemm_a4=5,gemm_a4_relu=0" "gemm_a4=4,gemm_a4_relu=1" --steps 12 --block 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_dact_again.txt
