# round 5: a4 weight-gradient form -- GPU tests, wgrad A/B, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
timeout 900 python tools/gemm_wgrad_ab.py "gemm_a4=0" "gemm_a4=1" 2>&1 | tee gpurun_out/r05_gemm_a4_wgrad_ab.txt | tail -16
timeout 900 python tools/step_ab.py "gemm_a4=0" "gemm_a4=1" --steps 8 --block 4 2>&1 | tail -3 | tee gpurun_out/r05_step_ab_a4.txt
