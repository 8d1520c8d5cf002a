cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_configs_gpu.py tests/test_dp_gpu.py "tests/test_kernels_gpu.py" -m gpu -q -k "cfg5 or two_ranks or sum_n or gemm_a4 or bf16_mode" 2>&1 | tail -15 > gpurun_out/r05_new_tests.txt
cat gpurun_out/r05_new_tests.txt
