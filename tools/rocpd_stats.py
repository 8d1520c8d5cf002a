"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
Usage: python tools/rocpd_stats.py results.db [top_n]"""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) " \
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for n, c, t, mn, mx in rows[:top]:
        n = re.sub(r"\(anonymous namespace\)::", "", n or "?")
        print(f"{n[:90]:90s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100.0 * t / tot:6.2f}")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
