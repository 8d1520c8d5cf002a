cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcprobe; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_BUBBLE|TCC_EA0_RD_UNCACHED" $O/counters.txt | head -40
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python $R/tools/gemm_one.py 32000 768 3072 0 1 1 > $O/$c.log 2>&1
  echo "$c rc=$?"; ls $O/$c 2>/dev/null
done
