# shader clock during the filter forward (debug build with -DFF_TIMING, ping-pong kernel): cycles and 100 MHz ticks of one wave
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEOSSL_HIP_LIB"] = sys.argv[1]
import subprocess
import numpy as np, torch
from geossl_amd import _lib
lib = _lib.load()
src = open(os.path.join(os.path.dirname(__file__), "bench_filter_fwd.py")).read().split("# reference value")[0]
sys.argv = sys.argv[:1]
exec(src[src.index("P = int("):])
buf = (C.c_longlong * (2 * 64 * 8 + 128))()
lib.geossl_filter_fwd_debug_read.argtypes = [C.c_void_p]
assert lib.geossl_filter_fwd_debug_read(buf) == 0
a = np.array(list(buf), dtype=np.int64)
for r, nm in ((0, "wave 0"), (1, "wave 4")):
    cyc = a[r * 512 + 1] - a[r * 512]
    ticks = a[1024 + r * 64 + 1] - a[1024 + r * 64]
    print(nm, "cycles", int(cyc), "ticks", int(ticks), "clock %.2f GHz" % (cyc / (ticks * 10.0)), "trips", int(a[r * 512 + 2]), "cycles per trip %.0f" % (cyc / max(1, a[r * 512 + 2])))
try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
