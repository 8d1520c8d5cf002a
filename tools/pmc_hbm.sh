#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_hbm}; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python $R/tools/hbm_kernels.py > $O/$c.log 2>&1
  echo "$c rc=$?"
done
tail -1 $O/FETCH_SIZE.log
