#!/bin/bash
# rocprofv3 kernel trace + stats over the bench command (stream overlap off so that per-kernel durations are not inflated by
# concurrent kernels).  Outputs under gpurun_out/$1.  (PMC passes: tools/pmc_traffic.sh -- rocprofv3 --pmc around the whole
# bench.py segfaults inside torch's integer elementwise kernels on this stack.)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
CMD="python $R/bench.py --steps 3 --warmup 1 --no-overlap --no-cpu-baseline --no-generate"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/trace.log 2>&1
rm -f $OUT/trace/t_kernel_trace.csv
tail -1 $OUT/trace.log | cut -c1-600
