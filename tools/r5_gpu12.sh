cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py "gemm_a4=0" "gemm_a4=1,gemm_a4_grid=0" "gemm_a4=1,gemm_a4_grid=224" "gemm_a4=1,gemm_a4_grid=208" "gemm_a4=1,gemm_a4_grid=192" "gemm_a4=1,gemm_a4_grid=160" --steps 10 --block 4 2>&1 | tail -6 | tee gpurun_out/r05_step_ab_a4_grid.txt
