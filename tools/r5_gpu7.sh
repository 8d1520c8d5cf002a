cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gemm_a4_ablate.py 2>&1 | tee gpurun_out/r05_gemm_a4_ablation.txt
timeout 900 bash tools/pmc_a4_abl.sh 2>&1 | tail -6 | tee -a gpurun_out/r05_gemm_a4_ablation.txt
