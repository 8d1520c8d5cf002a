cd $GRAFT_REPO_ROOT
timeout 900 python tools/step_ab.py "eng:overlap=0,gemm_a4=1" "eng:overlap=0,gemm_a4=5" "eng:overlap=1,gemm_a4=1" "eng:overlap=1,gemm_a4=5" "eng:overlap=1,gemm_a4=5,eng:dbg_skip_wgrad=1" "eng:overlap=1,gemm_a4=1,eng:dbg_skip_wgrad=1" --steps 10 --block 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_step_ab_dact_overlap.txt
