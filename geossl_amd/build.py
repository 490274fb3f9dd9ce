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
    deps = [os.path.join(CSRC, s) for s in _sources()] + _headers()
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
               "-I", CSRC, "-Wno-pass-failed", "-c", os.path.join(CSRC, s), "-o", obj]
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
