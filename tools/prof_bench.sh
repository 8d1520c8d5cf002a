#!/bin/bash
# rocprofv3 passes over the SAME bench command (stream overlap off so that per-kernel durations are not inflated by
# concurrent kernels): (1) kernel trace + stats, (2) PMC FETCH_SIZE, (3) PMC WRITE_SIZE.  Outputs under gpurun_out/$1.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
CMD="python $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-generate"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
tail -1 $OUT/trace.log | cut -c1-1500
ls -la $OUT/trace $OUT/fetch $OUT/write
