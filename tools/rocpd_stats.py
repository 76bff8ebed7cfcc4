#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average) from a rocprofv3 rocpd .db (--kernel-trace without --output-format csv)."""
import sqlite3
import sys


def main(path, top=25):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({sym})")]
    name = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute(f"select s.{name}, count(*), sum(d.end - d.start), min(d.end - d.start) from {disp} d join {sym} s "
                     f"on d.kernel_id = s.id group by s.{name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':90s} {'calls':>7s} {'total ms':>9s} {'avg us':>8s} {'min us':>8s} {'%':>6s}")
    for n, k, t, mn in rows[:top]:
        print(f"{n[:90]:90s} {k:7d} {t / 1e6:9.3f} {t / k / 1e3:8.2f} {mn / 1e3:8.2f} {100 * t / tot:6.2f}")
    print(f"total {tot / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
