cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/step_ab.py "gemm_a4=0" "gemm_a4=4" "gemm_a4=1" --steps 10 --block 4 2>&1 | tail -4 | tee gpurun_out/r05_step_ab_a4.txt
