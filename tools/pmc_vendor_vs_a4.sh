cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_va
mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
for shape in "8192 8192 8192" "32000 2304 768"; do
  tag=$(echo $shape | tr ' ' '_')
  V2S_OPTIONS=gemm_a4=3 timeout 300 rocprofv3 --pmc $SQ --kernel-trace -d $O/sq_a4_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $shape 0 0 1 > $O/log_a4_$tag.txt 2>&1
  timeout 300 rocprofv3 --pmc $SQ --kernel-trace -d $O/sq_vendor_$tag -o pmc --output-format csv -- python $R/tools/vendor_one.py $shape > $O/log_v_$tag.txt 2>&1
  V2S_OPTIONS=gemm_a4=3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_a4_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $shape 0 0 1 >> $O/log_a4_$tag.txt 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_vendor_$tag -o pmc --output-format csv -- python $R/tools/vendor_one.py $shape >> $O/log_v_$tag.txt 2>&1
  V2S_OPTIONS=gemm_a4=2 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_a4p_$tag -o pmc --output-format csv -- python $R/tools/gemm_one.py $shape 0 0 1 >> $O/log_a4_$tag.txt 2>&1
done
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_va"
for d in sorted(glob.glob(O + "/*_*")):
    if not os.path.isdir(d): continue
    try:
        rows = list(csv.DictReader(open(d + "/pmc_counter_collection.csv")))
        kt = list(csv.DictReader(open(d + "/pmc_kernel_trace.csv")))
    except OSError:
        print(os.path.basename(d), "no data"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r["Kernel_Name"]
        if "gemm_a4" in n or "Cijk" in n:
            agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = collections.defaultdict(list)
    for r in kt:
        n = r["Kernel_Name"]
        if "gemm_a4" in n or "Cijk" in n:
            durs[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, c in agg.items():
        v = {n: sum(x[-3:]) / len(x[-3:]) for n, x in c.items()}
        dur = sum(durs[k][-3:]) / len(durs[k][-3:])
        if "SQ_BUSY_CYCLES" in v:
            cyc = v["SQ_BUSY_CYCLES"] / 32; mf = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
            print(f"{os.path.basename(d):28s} {k[:48]:48s} {dur:8.1f} us  clock {cyc / dur / 1e3:.2f} GHz  MFMA busy {100 * mf / cyc:5.1f} %  parked {100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:4.1f} %  "
                  f"issue-stalled {100 * v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:4.1f} %  LDS insts {v['SQ_INSTS_LDS']:.3g}  bank conflicts {v['SQ_LDS_BANK_CONFLICT']:.3g}  waves x cycles {v['SQ_WAVE_CYCLES']:.3g}")
        else:
            print(f"{os.path.basename(d):28s} {k[:48]:48s} {dur:8.1f} us  FETCH_SIZE {v.get('FETCH_SIZE', 0):.4g} (x2 on gfx950 for wide reads; unit KB?)")
PY
