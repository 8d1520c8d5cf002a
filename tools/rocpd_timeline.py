"""Print the kernels between the last two dispatches whose name contains MARK (default: argmax) from a rocprofv3 rocpd database, in
launch order with durations -- one decode step.  Usage: python tools/rocpd_timeline.py results.db [MARK]"""
import re
import sqlite3
import sys
from collections import OrderedDict


def main(path, mark="argmax"):
    db = sqlite3.connect(path)
    rows = list(db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from "
                           "rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if mark in r[0]]
    a, b = idx[-2], idx[-1]
    agg = OrderedDict()
    for r in rows[a + 1:b + 1]:
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", r[0])[:44]
        key = (n, r[3] // max(r[5], 1), r[4], r[5])
        c, t = agg.get(key, (0, 0))
        agg[key] = (c + 1, t + r[2] - r[1])
    tot = 0
    for (n, gx, gy, wg), (c, t) in agg.items():
        print(f"{n:46s} grid {gx:5d} x {gy:2d}  wg {wg:4d}  calls {c:3d}  avg {t / c / 1e3:7.2f} us  total {t / 1e3:8.2f} us")
        tot += t
    print(f"sum of kernel durations {tot / 1e3:.1f} us; wall {(rows[b][2] - rows[a][2]) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
