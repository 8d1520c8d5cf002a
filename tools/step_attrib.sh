#!/bin/bash
# Overlapped per-arm kernel traces of the cfg-2 train step + per-symbol attribution (tools/step_attrib.py).
# usage: bash tools/step_attrib.sh <out name> <steps> "arm1" "arm2" ...     (arms = tools/step_ab.py settings, e.g. "gemm_a4=1" "gemm_a4=5")
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; STEPS=$2; shift 2
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
SPEC=""
i=0
for arm in "$@"; do
  d=$OUT/arm$i
  rm -rf $d
  rocprofv3 --kernel-trace -d $d -o t --output-format csv -- python $R/tools/step_ab.py "$arm" --steps $STEPS --block 1 > $OUT/arm$i.log 2>&1
  f=$(ls $d/*kernel_trace.csv $d/*/*kernel_trace.csv 2>/dev/null | head -1)
  tail -2 $OUT/arm$i.log
  n=${arm//[,:]/_}; SPEC="$SPEC ${n//=/-}=$f"
  i=$((i+1))
done
python $R/tools/step_attrib.py --steps $STEPS $SPEC > $OUT.txt 2>&1
python $R/tools/step_attrib.py --steps $STEPS --grid --top 70 $SPEC > ${OUT}_by_grid.txt 2>&1
rm -rf $OUT/arm*/  # the raw traces are large
head -40 $OUT.txt
