# per-phase marks of one block of k_layer_loop (debug build of chain.hip with -DLOOP_TIMING):
#   python tools/loop_timing.py <lib with LOOP_TIMING> [mols] [set]
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEOSSL_HIP_LIB"] = sys.argv[1]
mols = int(sys.argv[2]) if len(sys.argv) > 2 else 128
molset = sys.argv[3] if len(sys.argv) > 3 else "A"
import torch
import bench
from geossl_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
wl = bench.Workload(dev, 0, 1, mols=mols, molset=molset, n_batches=2)
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
buf, n = (C.c_longlong * 2048)(), C.c_int(0)
lib.geossl_loop_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.geossl_loop_debug_read(buf, C.byref(n), 1) == 0
wl.step(6)
torch.cuda.synchronize()
assert lib.geossl_loop_debug_read(buf, C.byref(n), 1) == 0
marks = [(buf[2 * i], buf[2 * i + 1]) for i in range(n.value)]
names = {1: "loop start", 2: "loop end", 10: "op: chain", 11: "op: aggregate", 20: "op done (before store wait)", 21: "stores complete",
         100: "  input rows split into LDS", 30: "  x rows / flags requested + written to LDS", 31: "  barrier",
         32: "  sums done", 33: "  barrier"}
t0 = marks[0][1]
prev = t0
for tag, t in marks:
    print("%9.2f us  +%6.2f  %s" % ((t - t0) / 100.0, (t - prev) / 100.0, names.get(tag, "  stage %d done" % (tag - 110) if tag >= 110 else str(tag))))
    prev = t
