#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 kernel trace (rocpd SQLite): kernels in start order with their
start offset, duration and the gap to the previous kernel's end; busy time, idle time and overlap of the step.
    python tools/timeline.py trace.db [anchor-kernel-substring] [which occurrence]"""
import sqlite3
import sys


def main(path, anchor="k_radius", which=-3):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                          % (name_col, kd, ks)))
    anchors = [i for i, r in enumerate(rows) if anchor in r[0]]
    a0, a1 = anchors[which], anchors[which + 1]
    step = rows[a0:a1]
    t0 = step[0][1]
    print("step of %d kernels, %.3f ms wall" % (len(step), (step[-1][2] - t0) / 1e6))
    busy_end, idle, total = t0, 0, 0
    for name, s, e in step:
        gap = s - busy_end
        if gap > 0:
            idle += gap
        short = name.split("(")[0][-60:]
        print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short))
        busy_end = max(busy_end, e)
        total += e - s
    print("sum of kernel durations %.3f ms, idle (no kernel running) %.3f ms" % (total / 1e6, idle / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *(int(x) for x in sys.argv[3:4]))
