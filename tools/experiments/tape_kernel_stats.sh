#!/bin/bash
# per-kernel time of the second-order route (tape) for one backbone: bash tools/experiments/tape_kernel_stats.sh painn 1024
cd "$(dirname "$0")/../.." || exit 1
REPO=$PWD
export TMPDIR=/tmp
D=/tmp/tape_stats; rm -rf $D
( cd /tmp && rocprofv3 --kernel-trace --stats -d $D -- python3 $REPO/tools/experiments/tape_host_profile.py ${1:-painn} ${2:-1024} ) 2>&1 | grep molecules
python3 - $D <<'PY'
import glob, os, sqlite3, sys
db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
col = "kernel_name" if "kernel_name" in scols else "display_name"
rows = list(c.execute("select s.%s, count(*), sum(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s "
                      "order by 3 desc" % (col, kd, ks, col)))
tot = sum(r[2] for r in rows)
for n, k, t in rows[:30]:
    print("%8d %9.3f ms %5.1f%% avg %8.1f us  %s" % (k, t / 1e6, 100.0 * t / tot, t / k / 1e3, n[:110]))
PY
