"""Summarise the SQ counter passes of tools/pmc_attn.sh -> text for profiles/.  usage: pmc_attn_summary.py gpurun_out"""
import csv, collections, sys, os
root = sys.argv[1]
def load(d):
    rows = list(csv.DictReader(open(os.path.join(root, d, "pmc_counter_collection.csv"))))
    kt = list(csv.DictReader(open(os.path.join(root, d, "pmc_kernel_trace.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in rows:
        if "attn" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("((anon")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in kt:
        if "attn" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("((anon")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return agg, dur
a, da = load("pmc_attn_a"); b, db = load("pmc_attn_b")
N = int(os.environ.get("ATTN_N", "1000"))
out = [f"rocprofv3 --pmc over tools/attn_one.py (encoder-layer attention, B=32 H=12 N={N}, bias + mask + dropout 0.1 + dbias), 1 x MI355X; last 3 of 4 launches averaged"]
for k in a:
    if "delta" in k: continue
    va = {n: sum(x[-3:]) / 3 for n, x in a[k].items()}; vb = {n: sum(x[-3:]) / 3 for n, x in b.get(k, {}).items()}
    dur = sum(da[k][-3:]) / 3
    cyc = va["SQ_BUSY_CYCLES"] / 32                       # per shader engine -> elapsed shader cycles
    tot = va["SQ_WAVE_CYCLES"]
    q = 4.0 / 1024 / cyc                                  # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, summed over 1024 SIMDs
    out.append(f"{k}: {dur:7.1f} us under the profiler, shader clock {cyc / dur / 1e3:.2f} GHz, resident waves per SIMD {tot * q:4.2f}")
    out.append(f"    per SIMD, % of elapsed cycles: some wave issuing {100 * va['SQ_ACTIVE_INST_ANY'] * q:5.1f} (VALU {100 * va['SQ_ACTIVE_INST_VALU'] * q:5.1f}"
               + (f", scalar {100 * vb['SQ_ACTIVE_INST_SCA'] * q:5.1f}, LDS {100 * vb['SQ_ACTIVE_INST_LDS'] * q:5.1f}" if vb else "")
               + f"), MFMA pipe busy {100 * va['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.1f}")
    out.append(f"    wave time: parked at waitcnt/barrier {100 * va['SQ_WAIT_ANY'] / tot:4.1f} %, issue-stalled {100 * va['SQ_WAIT_INST_ANY'] / tot:4.1f} %"
               f" (of which on LDS {100 * va['SQ_WAIT_INST_LDS'] / tot:4.1f} %), issuing {100 * va['SQ_ACTIVE_INST_ANY'] / tot:4.1f} %")
    if vb:
        nel = 32 * 12 * N * N / 64.0                # score elements per lane-slot (wave-level instruction counts / this = per element)
        out.append(f"    wave-level instructions per score element: VALU {vb['SQ_INSTS_VALU'] / nel:5.2f} (+ MFMA {vb['SQ_INSTS_MFMA'] / nel:4.2f}), scalar {vb['SQ_INSTS_SALU'] / nel:5.2f},"
                   f" LDS {vb['SQ_INSTS_LDS'] / nel:4.2f}; {va['SQ_ACTIVE_INST_VALU'] * 4 / vb['SQ_INSTS_VALU']:.2f} cycles per VALU instruction;"
                   f" LDS bank-conflict cycles / LDS active cycles = {vb['SQ_LDS_BANK_CONFLICT'] / max(vb['SQ_LDS_IDX_ACTIVE'], 1):.2f}")
    out.append("    raw: " + " ".join(f"{n}={v:.4g}" for n, v in sorted({**va, **vb}.items())))
print("\n".join(out))
