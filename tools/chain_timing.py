# per-phase clock64 marks of one wave of geossl_linear_chain (debug build with -DCHAIN_TIMING)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEOSSL_HIP_LIB"] = sys.argv[1]
import torch
from geossl_amd import _lib, ops
dev, F, R = "cuda:0", 128, 36864
X = torch.randn(R, F, device=dev)
Ws = [torch.randn(F, F, device=dev) / F ** 0.5 for _ in range(4)]
b = torch.randn(F, device=dev); res = torch.randn(R, F, device=dev)
imgs = ops.prepare_chain(Ws)
st = [dict(image=imgs[0], bias=b, flags=_lib.EPI_SSP), dict(image=imgs[1], bias=b, res=res), dict(image=imgs[2])]
for _ in range(3):
    ops.linear_chain(X, st)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_longlong * 512)()
lib.geossl_chain_debug_read.argtypes = [C.c_void_p]
assert lib.geossl_chain_debug_read(buf) == 0
v = list(buf)
rows = [v[8 * i: 8 * i + 8] for i in range(64)]
t0 = rows[0][0]
flat = []
for i, r in enumerate(rows):
    for slot, x in enumerate(r):
        if x:
            flat.append((x - t0, slot))
flat.sort()
prev = 0
for tt, slot in flat[:80]:
    print("%8d  +%6d  mark %d" % (tt, tt - prev, slot))
    prev = tt
