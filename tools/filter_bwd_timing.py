# per-phase clock64 marks of the filter backward (debug build with -DFB_TIMING): wave A_0 and wave B_0 of block (3, 0)
#   tools/build_variant.sh filter_bwd.hip /path/variant.so -DFB_TIMING;  python tools/filter_bwd_timing.py /path/variant.so
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEOSSL_HIP_LIB"] = sys.argv[1]
import numpy as np, torch
from geossl_amd import _lib
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.synthetic import make_batch, draw_noise
from geossl_amd.Geom3D.models import SchNet
from geossl_amd.NCSN import NCSN_version_03
dev = "cuda:0"
torch.manual_seed(0)
model = SchNet(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9).to(dev)
n1 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2.0).to(dev)
n2 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2.0).to(dev)
tr = pg.DDMTrainer(model, n1, n2)
b = make_batch(1024, seed=1)
batch = pg.Batch.from_numpy(b, dev)
noise = {k: torch.from_numpy(v).to(dev) for k, v in draw_noise(b, 2).items()}
for _ in range(3):
    tr.step(batch, noise)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_longlong * (2 * 64 * 8))()
lib.geossl_filter_bwd_debug_read.argtypes = [C.c_void_p]
assert lib.geossl_filter_bwd_debug_read(buf) == 0
v = np.array(list(buf), dtype=np.int64).reshape(2, 64, 8)
buf2 = (C.c_longlong * (2 * 64))()
lib.geossl_filter_bwd_debug_read2.argtypes = [C.c_void_p]
assert lib.geossl_filter_bwd_debug_read2(buf2) == 0
v2 = np.array(list(buf2), dtype=np.int64).reshape(2, 64)
names = ["wait at barrier 1", "build", "wait at barrier 2", "publish + requests", "   of which publish (incl. wait for its data)", "first MFMA part (A: dt, dU; B: transposes)",
         "second MFMA part (A: dW1; B: dW2)"]
for role, nm in ((0, "A_0"), (1, "B_0")):
    m = v[role]
    ok = [t for t in range(2, 40) if m[t, 0] and m[t + 1, 0]]
    d = np.array([[m[t, 1] - m[t, 0], m[t, 2] - m[t, 1], m[t, 3] - m[t, 2], m[t, 4] - m[t, 3], m[t, 7] - m[t, 3], m[t, 6] - m[t, 4],
                   m[t, 5] - m[t, 6], m[t + 1, 0] - m[t, 0]] for t in ok])
    d = np.concatenate([d[:, :7], d[:, 7:]], axis=1)
    print(nm, "tiles", len(ok), "cycles per tile (mean):", int(d[:, 7].mean()))
    print("   %-46s %7.0f" % ("(barrier 2 -> all earlier global requests done)", np.mean([v2[role, t] - m[t, 3] for t in ok])))
    for i, n in enumerate(names):
        print("   %-46s %7.0f" % (n, d[:, i].mean()))

if hasattr(lib, "geossl_filter_bwd_debug_read3"):
    b3 = (C.c_longlong * 256)()
    lib.geossl_filter_bwd_debug_read3.argtypes = [C.c_void_p]
    assert lib.geossl_filter_bwd_debug_read3(b3) == 0
    m3 = np.array(list(b3), dtype=np.int64).reshape(64, 4)
    ok = [t for t in range(3, 38) if m3[t, 0] and m3[t, 3]]
    print("inside publish (wave B_0): descriptors %.0f, window stores + maxima %.0f, wave maxima %.0f" %
          (np.mean([m3[t, 1] - m3[t, 0] for t in ok]), np.mean([m3[t, 2] - m3[t, 1] for t in ok]), np.mean([m3[t, 3] - m3[t, 2] for t in ok])))
