#!/usr/bin/env python3
"""Static instruction mix of one kernel of a csrc translation unit, per basic block (gfx950 assembly from -save-temps):
    python tools/isa_mix.py filter_bwd.hip k_filter_bwd_hILi4ELb0 [min block size]
Blocks with a backward branch are the loops; the mix of the tile loop is what a wave issues per tile."""
import collections, glob, os, re, subprocess, sys, tempfile
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tmp = tempfile.mkdtemp(prefix="isa_")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", here + "/include", "-I",
                here + "/geossl_amd/csrc", "-Wno-pass-failed", "-c", here + "/geossl_amd/csrc/" + src, "-o", "x.o",
                "-save-temps=obj"], cwd=tmp, stderr=subprocess.DEVNULL, check=True)
s = open(glob.glob(tmp + "/*gfx950*.s")[0]).read()
m = re.search(r"^(_ZN\S*" + re.escape(pat) + r"\S*):", s, re.M)
body = s[m.end():s.index("s_endpgm", m.end())]
blocks, cur, name = [], [], "entry"
for l in body.split("\n"):
    l = l.strip()
    if re.match(r"^\.?[A-Za-z_0-9$]+:", l):
        blocks.append((name, cur)); cur = []; name = l.split(":")[0]
    elif l and not l.startswith((".", ";")):
        cur.append(l)
blocks.append((name, cur))


def mix(ins):
    c = collections.Counter()
    for l in ins:
        x = l.split()[0]
        if x.startswith("v_mfma"): c["mfma"] += 1
        elif x.startswith(("v_exp", "v_log", "v_rcp", "v_sqrt", "v_rsq")): c["trans"] += 1
        elif x.startswith("v_"): c["valu"] += 1
        elif x.startswith("ds_"): c["lds"] += 1
        elif x.startswith(("global_", "buffer_", "scratch_")): c["vmem"] += 1
        elif x.startswith("s_waitcnt"): c["wait"] += 1
        elif x.startswith("s_"): c["salu"] += 1
        else: c["other"] += 1
    return dict(c)


tot = collections.Counter()
for n, b in blocks:
    if len(b) >= minsz:
        print("%-14s %5d  %s" % (n, len(b), mix(b)))
for n, b in blocks:
    tot.update(mix(b))
print("whole kernel:", dict(tot))
if len(sys.argv) > 4:  # top VALU opcodes
    c = collections.Counter(l.split()[0] for _, b in blocks for l in b if l.startswith("v_"))
    print(c.most_common(30))
