"""Per-symbol attribution of a change inside the OVERLAPPED train step: which kernels pay when a launch that is
faster alone makes the step slower.  Input: one rocprofv3 --kernel-trace CSV per arm (tools/step_attrib.sh runs tools/step_ab.py with ONE
setting under the profiler, streams overlapped as in the product step).  The window is the last `steps` optimizer steps (delimited by
adam_kernel dispatches); per symbol (and, with --grid, per symbol x grid size = per GEMM shape) the summed duration per step is printed for
each arm with the difference to the first arm, plus the window's wall time and the union of busy time per stream.
usage: python tools/step_attrib.py --steps 12 name_a=trace_a.csv name_b=trace_b.csv [--grid] [--top 40]"""
import argparse, csv, re, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("arms", nargs="+")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--grid", action="store_true")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0")),
                     r.get("Grid_Size_X", r.get("Grid_Size", "0")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "0"))))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    assert len(adam) > a.steps, f"{path}: {len(adam)} adam_kernel dispatches, need > {a.steps}"
    t0, t1 = rows[adam[-a.steps - 1]][1], rows[adam[-1]][1]
    win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    per = defaultdict(lambda: [0, 0])
    for s, e, n, q, g, w in win:
        k = short(n) + (f" g{g}" if a.grid else "")
        per[k][0] += e - s
        per[k][1] += 1
    # union of the intervals in which ANY kernel runs (the rest of the window the GPU is idle: host / dependency gaps), and the time with exactly one / more kernels
    ev = sorted([(s_, 1) for s_, e_, *_ in win] + [(e_, -1) for s_, e_, *_ in win])
    busy = one = 0
    depth, last = 0, t0
    for t_, d_ in ev:
        if depth > 0:
            busy += t_ - last
        if depth == 1:
            one += t_ - last
        depth += d_; last = t_
    load.extra = (busy / a.steps / 1e6, one / a.steps / 1e6)
    return per, (t1 - t0) / a.steps / 1e6, sum(e - s for s, e, *_ in win) / a.steps / 1e6


arms = []
for spec in a.arms:
    name, path = spec.split("=", 1)
    per, wall, ksum = load(path)
    arms.append((name, per, wall, ksum))
    print(f"{name}: wall {wall:.3f} ms/step, some kernel running {load.extra[0]:.3f} ms ({100 * load.extra[0] / wall:.1f} %), exactly one kernel running {load.extra[1]:.3f} ms, idle {wall - load.extra[0]:.3f} ms")
base = arms[0]
print(f"window = last {a.steps} optimizer steps of each trace (streams overlapped: a kernel's duration includes what concurrent kernels cost it)")
print("wall ms/step: " + "   ".join(f"{n} {w:.3f}" for n, _, w, _ in arms) + "     summed kernel ms/step: " + "   ".join(f"{n} {k:.3f}" for n, _, _, k in arms))
keys = sorted(set().union(*[set(p) for _, p, _, _ in arms]), key=lambda k: -max(p.get(k, [0, 0])[0] for _, p, _, _ in arms))
hdr = f"{'kernel':78s}" + "".join(f" {n[:14]:>14s} {'n':>5s}" for n, *_ in arms) + "".join(f" {'d(' + n[:10] + ')':>13s}" for n, *_ in arms[1:])
print(hdr)
tot = [0.0] * len(arms)
rest = [0.0] * len(arms)
for i, k in enumerate(keys):
    v = [p.get(k, [0, 0]) for _, p, _, _ in arms]
    ms = [x[0] / a.steps / 1e6 for x in v]
    for j, m in enumerate(ms):
        tot[j] += m
    if i < a.top:
        print(f"{k:78s}" + "".join(f" {m:14.3f} {x[1] / a.steps:5.0f}" for m, x in zip(ms, v)) + "".join(f" {m - ms[0]:+13.3f}" for m in ms[1:]))
    else:
        for j, m in enumerate(ms):
            rest[j] += m
print(f"{'(all other kernels)':78s}" + "".join(f" {m:14.3f} {'':5s}" for m in rest) + "".join(f" {m - rest[0]:+13.3f}" for m in rest[1:]))
print(f"{'TOTAL':78s}" + "".join(f" {m:14.3f} {'':5s}" for m in tot) + "".join(f" {m - tot[0]:+13.3f}" for m in tot[1:]))
