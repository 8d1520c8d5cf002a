#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_order; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/FETCH_SIZE -o p --output-format csv -- python $R/tools/pmc_order_probe.py > $O/run.log 2>&1
echo "rc=$?"; grep "order" $O/run.log
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/FETCH_SIZE/p_counter_collection.csv")) if "gemm" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
vals=[float(r["Counter_Value"]) for r in rows]
ORD=[1,2,4,8,16,32]; SH=[(32000,768,3072),(32000,768,768),(32000,768,3072)]
i=0
for M,N,K in SH:
    for o in ORD:
        v=vals[i+2]; alg_read=2.0*(M*K+K*N)
        print(f"{M}x{N}x{K} order {o}: FETCH_SIZE {v/1024:.1f} MB raw, x2 = {2*v*1024/1e6:.1f} MB vs algorithmic reads {alg_read/1e6:.1f} MB -> {2*v*1024/alg_read:.2f}x")
        i+=13
PY
