# per-phase clock64 marks of the filter forward (debug build with -DFF_TIMING): waves 0 and 4 (one SIMD) of block (3, 0), and the
# shader clock during the kernel (s_memtime against the 100 MHz wall clock).  The marks sit in the three-bf16-piece kernel
# (GEOSSL_FILTER_FWD_BF16X3, set below); the two-piece kernel has the same phase structure with half the MFMAs.
#   tools/build_variant.sh filter_fwd.hip /path/variant.so -DFF_TIMING;  python tools/filter_fwd_timing.py /path/variant.so
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GEOSSL_HIP_LIB"] = sys.argv[1]
os.environ["GEOSSL_FILTER_FWD_BF16X3"] = "1"
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
buf = (C.c_longlong * (2 * 64 * 8 + 128))()
lib.geossl_filter_fwd_debug_read.argtypes = [C.c_void_p]
assert lib.geossl_filter_fwd_debug_read(buf) == 0
allv = np.array(list(buf), dtype=np.int64)
v = allv[:1024].reshape(2, 64, 8)
wall = allv[1024:].reshape(2, 64)
names = ["rbf + split", "first GEMM (96 MFMA) + ssp/split/store of blocks 0-2", "ssp/split/store of block 3", "second GEMM pair 0 (96 MFMA)",
         "epilogue pair 0", "second GEMM pair 1 (96 MFMA)", "epilogue pair 1"]
for role, nm in ((0, "wave 0"), (1, "wave 4")):
    m = v[role]
    ok = [t for t in range(2, 26) if m[t, 0] and m[t + 1, 0]]
    d = np.array([[m[t, i + 1] - m[t, i] for i in range(7)] + [m[t + 1, 0] - m[t, 0]] for t in ok])
    print(nm, "tiles", len(ok), "cycles per tile (mean):", int(d[:, 7].mean()))
    for i, n in enumerate(names):
        print("   %-56s %7.0f" % (n, d[:, i].mean()))
# phase offsets between the two waves of the SIMD
m0, m1 = v[0], v[1]
print("tile start of wave 4 minus tile start of wave 0, tiles 2..12:", [int(m1[t, 0] - m0[t, 0]) for t in range(2, 13)])
for role, nm in ((0, "wave 0"), (1, "wave 4")):
    last = max(t for t in range(64) if v[role][t, 0])
    cyc, ticks = v[role][last, 0] - v[role][1, 0], wall[role][last] - wall[role][1]
    print(nm, "tiles", last, "shader cycles", int(cyc), "wall 100 MHz ticks", int(ticks), "=> clock %.2f GHz" % (cyc / (ticks * 10.0)),
          "; cycles per tile over the whole run %.0f" % (cyc / (last - 1)))
    print("   per-tile cycles:", [int(v[role][t + 1, 0] - v[role][t, 0]) for t in range(1, last)])
