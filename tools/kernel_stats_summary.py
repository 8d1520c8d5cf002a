"""rocprofv3 --kernel-trace --stats CSV -> text summary for profiles/.  usage: kernel_stats_summary.py <dir with *_kernel_stats.csv> <bench log> <out.txt> <steps>"""
import csv, glob, json, sys
src, log, out, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
rows = list(csv.DictReader(open(glob.glob(src + "/*_kernel_stats.csv")[0])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
lines = [f"rocprofv3 --kernel-trace --stats summary ({steps} trainer steps incl. warm-up and the roofline leg; setup kernels included)",
         f"total kernel time {tot / 1e6:.1f} ms", "", f"{'kernel':92s} {'calls':>7s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}"]
for r in rows[:70]:
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n if len(n) <= 90 else n[:87] + "..."
    lines.append(f"{n:92s} {int(r['Calls']):7d} {int(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
j = None
for l in open(log):
    if l.startswith("{"):
        j = json.loads(l)
if j:
    lines += ["", "bench.py JSON line of the same command:", json.dumps({k: v for k, v in j.items() if k != "kernel_breakdown_ms_per_step"})]
    lines += ["", "live HIP-event breakdown of one step (stream overlap off):"]
    for k, v in j.get("kernel_breakdown_ms_per_step", {}).items():
        lines.append(f"  {k:70s} launches={v['launches']:4d} ms={v['ms']:8.3f} avg_us={v['ms'] / v['launches'] * 1e3:8.1f} TF/s={v['tflops']}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
