#!/bin/bash
# rocprofv3 kernel stats of beam search at B=16, 4 beams (the bench's generate_beam4 leg), eager steps
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_beam}; mkdir -p $OUT
BEAM_B=${BEAM_B:-16} rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $R/tools/beam_prof.py > $OUT/trace.log 2>&1
rm -f $OUT/trace/t_kernel_trace.csv
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/trace/*_kernel_stats.csv")[0])))
for r in rows[:24]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:80]
    print(f"{n:82s} {int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):6.2f}%")
PY
