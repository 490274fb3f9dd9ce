# per-step periods (k_adam end to k_adam end) and the largest idle gaps of a rocprofv3 --kernel-trace run: python3 step_periods.py <dir>
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (col, kd, ks)))
ends = [e for n, s, e in rows if "k_adam" in n]
per = [(b - a) / 1e3 for a, b in zip(ends, ends[1:])]
per_s = sorted(per)
print("steps %d  period us: min %.0f median %.0f mean %.0f p90 %.0f max %.0f" % (len(per), per_s[0], per_s[len(per) // 2], sum(per) / len(per), per_s[int(0.9 * len(per))], per_s[-1]))
print("periods of steps 100..130:", " ".join("%.0f" % p for p in per[100:130]))
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps.append(((s1 - e0) / 1e3, n0.split("(")[0][-40:], n1.split("(")[0][-40:]))
gaps.sort(reverse=True)
from collections import Counter
big = [g for g in gaps if g[0] > 100]
print("gaps > 100 us: %d, total %.1f ms" % (len(big), sum(g[0] for g in big) / 1e3))
cnt = Counter((g[1], g[2]) for g in big)
for (a, b), k in cnt.most_common(8):
    tot = sum(g[0] for g in big if (g[1], g[2]) == (a, b))
    print("  %4d x  %-42s -> %-42s total %.1f ms" % (k, a, b, tot / 1e3))
