cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh pmc_traffic_r05_dgrad dgrad > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic_r05_dgrad gpurun_out/r05_pmc_traffic_gemm_dma_dgrad.json dgrad | head -12
bash tools/pmc_traffic.sh pmc_traffic_r05_a4p a4p > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic_r05_a4p gpurun_out/r05_pmc_traffic_gemm_a4p_dgrad.json a4p | head -12
