"""Summarise the SQ counter passes of tools/pmc_gemm.sh (MFMA utilisation of the GEMM kernels) -> text for profiles/."""
import csv, collections, sys, glob, os
out = []
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*_*_*"))):
    if not os.path.isdir(d):
        continue
    rows = list(csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))))
    kt = list(csv.DictReader(open(os.path.join(d, "pmc_kernel_trace.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "gemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("((anon")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt if "gemm" in r["Kernel_Name"]]
    M, N, K = os.path.basename(d).split("_")[1:4]
    for k, c in agg.items():
        v = {n: sum(x[-3:]) / 3 for n, x in c.items()}
        dur = sum(durs[-3:]) / 3
        cyc = v["SQ_BUSY_CYCLES"] / 32                       # per shader engine -> elapsed shader cycles
        mfma = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024           # per SIMD (256 CUs x 4)
        tot = v["SQ_WAVE_CYCLES"]
        out.append(f"{k}  M={M} N={N} K={K}: {dur:8.1f} us, shader clock {cyc / dur / 1e3:.2f} GHz, MFMA busy {100 * mfma / cyc:5.1f} % of elapsed cycles, "
                   f"{2.0 * int(M) * int(N) * int(K) / dur / 1e6:7.1f} TFLOP/s; wave time: parked (waitcnt/barrier) {100 * v['SQ_WAIT_ANY'] / tot:4.1f} %, "
                   f"issue-stalled {100 * v['SQ_WAIT_INST_ANY'] / tot:4.1f} %, issuing {100 * v['SQ_ACTIVE_INST_ANY'] / tot:4.1f} %, "
                   f"LDS bank-conflict cycles {v['SQ_LDS_BANK_CONFLICT']:.0f}")
print("\n".join(out))
