cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
timeout 900 python tools/gemm_vendor_ab.py 2>&1 | tee gpurun_out/r05_gemm_vendor_ab.txt | tail -34
timeout 900 python tools/step_ab.py "gemm_a4=0" "gemm_a4=1" --steps 10 --block 4 2>&1 | tail -3 | tee gpurun_out/r05_step_ab_a4.txt
