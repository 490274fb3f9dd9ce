"""Build libgeossl_hip.so (gfx950) in-tree with hipcc.  `python -m geossl_amd.build` or build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgeossl_hip.so")


def _sources():
    """Every .hip file of csrc/ is a translation unit of the library; every header (csrc/*.h, include/*.h) is a
    dependency of all of them."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


# Extra compiler flags of single translation units.  painn_mma.hip: no packed fp32 arithmetic (the SLP vectoriser is what
# forms v_pk_*_f32 here).  With it the mu-zero form of k_painn_fwd_mma contained `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` (both
# results read the HIGH half of src1), and on MI355X that instruction now and then returned a low result of 0 in lanes
# 48-63 when a second wave shared the SIMD: one term of a sum of four missing, 5-10 atoms of 18 432 per launch, never with
# one block per CU, never without packed ops (tools/scan_packed_opsel.py).  Same speed either way (140 / 173 us).
SOURCE_FLAGS = {"painn_mma.hip": ["-fno-slp-vectorize"]}


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(REPO, "include")
    return hs + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]


def source_hash():
    """sha256 over every translation unit and header of the library (names and contents, sorted): what a measurement
    of the kernels is valid for.  tools/pmc_traffic.py stamps its summaries with it and bench.py reports the traffic of a
    summary only while the stamp still matches."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr(sorted(SOURCE_FLAGS.items())).encode())
    for path in sorted([os.path.join(CSRC, s_) for s_ in _sources()] + _headers()):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers() + [os.path.abspath(__file__)]   # (SOURCE_FLAGS)
    return _newest(deps) > os.path.getmtime(LIB_PATH)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in _sources():
        obj = os.path.join(LIB_DIR, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"),
               "-I", CSRC, "-Wno-pass-failed"] + SOURCE_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
