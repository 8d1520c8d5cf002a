cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s -k "cfg2_shape" 2>&1 | grep -v "^$" | tail -25 | tee gpurun_out/r05_bf16mode_parity.txt
