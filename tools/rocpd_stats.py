#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a per-kernel table
(name, calls, total ms, avg us, share) — the same numbers `rocprofv3 --stats` prints."""
import sqlite3
import sys


def main(path, top=40, skip_first_ms=0.0):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print("%-96s %7s %10s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, mn, mx in rows[:top]:
        print("%-96s %7d %10.3f %10.2f %10.2f %10.2f %6.2f" % (name[:96], n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3,
                                                             100.0 * tot / total))
    print("TOTAL kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
