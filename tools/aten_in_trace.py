# kernels of a rocprofv3 --kernel-trace result (rocpd .db under <dir>) that are NOT this library's (every kernel of
# libgeossl_hip.so is named k_*; the namespace is anonymous): python tools/aten_in_trace.py <dir>
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute("select s.%s, count(*), sum(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s "
                      "order by 2 desc" % (col, kd, ks, col)))
mine = lambda n: "k_" in n.split("(")[0] and "at::" not in n
ours = [r for r in rows if mine(r[0])]
other = [r for r in rows if not mine(r[0])]
print("library kernels: %d names, %d launches, %.3f ms" % (len(ours), sum(r[1] for r in ours), sum(r[2] for r in ours) / 1e6))
print("other kernels:   %d names, %d launches, %.3f ms" % (len(other), sum(r[1] for r in other), sum(r[2] for r in other) / 1e6))
for n, k, t in other:
    print("  %6d %9.3f ms  %s" % (k, t / 1e6, n[:180]))
