cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_a4 or gemm_epilogues or p8" 2>&1 | tail -5
timeout 300 python tools/gemm_a4_relu_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gemm_a4_relu_ab.txt
timeout 900 python tools/step_ab.py "gemm_a4_relu=0" "gemm_a4_relu=1" --steps 10 --block 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_a4_relu.txt
