cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py "eng:defer_wgrads=0,gemm_a4=1" "eng:defer_wgrads=1,gemm_a4=1" "eng:defer_wgrads=2,gemm_a4=1" "eng:defer_wgrads=1,gemm_a4=5" "eng:defer_wgrads=2,gemm_a4=5" "eng:defer_wgrads=2,gemm_a4=4" "eng:defer_wgrads=0,gemm_a4=5" --steps 10 --block 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_step_ab_defer.txt
cat gpurun_out/r05_step_ab_defer.txt
