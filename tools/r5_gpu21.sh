cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "bit_identical" 2>&1 | tail -5
