#!/usr/bin/env python3
"""Ordered launch list of ONE replayed step from a rocprofv3 kernel trace (rocpd SQLite): start offset, duration and
the gap to the previous kernel's end.  A step is delimited by the optimiser kernel (k_adam).
    rocprofv3 --kernel-trace -d out -o t -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary
    python tools/step_timeline.py out/*.db [steps back from the end, default 2]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+|multi_tensor_apply_kernel\w*|[A-Za-z_]+Functor[A-Za-z_]*|CatArray\w*|distribution\w*|copyBuffer|radixSort\w*|\w*scan\w*)", name)
    return m.group(1) if m else name[:40]


def main(path, back=2, marker="k_adam"):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                          % (name_col, kd, ks)))
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    # a marker kernel that is launched several times in a row (multi-tensor optimiser kernels): the last of each run
    ends = [i for n_, i in enumerate(ends) if n_ + 1 == len(ends) or ends[n_ + 1] - i > 8]
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    prev_end = rows[lo - 1][2]
    busy = gaps = 0.0
    print("# %d launches, %.1f us from the end of the previous step's optimiser kernel to the end of this one's"
          % (len(step), (step[-1][2] - prev_end) / 1e3))
    print("%9s %8s %7s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    run_end = prev_end
    for name, s, e in step:
        gap = (s - run_end) / 1e3
        print("%9.1f %8.2f %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(name)))
        busy += (e - s) / 1e3
        gaps += max(gap, 0.0)
        run_end = max(run_end, e)
    print("# kernel time %.1f us, idle gaps %.1f us" % (busy, gaps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, sys.argv[3] if len(sys.argv) > 3 else "k_adam")
