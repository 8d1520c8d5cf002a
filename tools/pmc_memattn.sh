#!/bin/bash
# HBM traffic of the decode cross-attention kernels by PMC, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit together)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_memattn; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python $R/tools/memattn_pmc.py > $O/$c.log 2>&1
  echo "$c rc=$?"
done
tail -1 $O/FETCH_SIZE.log
