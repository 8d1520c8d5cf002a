# round 5, GPU call 1: gemm_a4 correctness, A/B table, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gemm_a4_ab.py --check-only > gpurun_out/r05_a4_check.txt 2>&1; rc=$?
tail -20 gpurun_out/r05_a4_check.txt
if [ $rc -ne 0 ]; then echo "CHECK FAILED rc=$rc"; fi
timeout 900 python tools/gemm_a4_ab.py > gpurun_out/r05_gemm_a4_ab.txt 2>&1
tail -30 gpurun_out/r05_gemm_a4_ab.txt
timeout 900 bash tools/pmc_gemm_a4.sh > gpurun_out/r05_pmc_sq_gemm_a4.txt 2>&1
tail -12 gpurun_out/r05_pmc_sq_gemm_a4.txt
