# per-phase clock64 marks of the one-pass NCSN backward (debug build with -DNB_TIMING): waves A_0 and B_0 of block 5
#   tools/build_variant.sh ncsn_bwd.hip /path/variant.so -DNB_TIMING;  python tools/ncsn_bwd_timing.py /path/variant.so
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
buf = (C.c_longlong * (2 * 32 * 8))()
lib.geossl_ncsn_bwd_debug_read.argtypes = [C.c_void_p]
assert lib.geossl_ncsn_bwd_debug_read(buf) == 0
v = np.array(list(buf), dtype=np.int64).reshape(2, 32, 8)
for role, nm, names in ((0, "A_0", ["wait at barrier X", "dz1 (24 MFMAs)", "mask, a1^T publish, split, transposes, demb", "dW1 (48 MFMAs)", "wait at barrier Y", "phase 2 (narrow gradients, requests)"]),
                        (1, "B_0", ["wait at barrier X", "dz2^T transposes (6 MFMAs)", "build dz2 fragments", "build (h_u+h_v)^T", "wait at barrier Y", "requests", "dfeat (48) + dW2 (24 MFMAs), stores"])):
    m = v[role]
    ok = [t for t in range(2, 16) if m[t, 0] and m[t + 1, 0]]
    rows = []
    for t in ok:
        if role == 0:
            rows.append([m[t, 1] - m[t, 0], m[t, 5] - m[t, 1], m[t, 6] - m[t, 5], m[t, 2] - m[t, 6], m[t, 3] - m[t, 2], m[t, 4] - m[t, 3], m[t + 1, 0] - m[t, 0]])
        else:
            rows.append([m[t, 1] - m[t, 0], m[t, 5] - m[t, 1], m[t, 6] - m[t, 5], m[t, 2] - m[t, 6], m[t, 3] - m[t, 2], m[t, 7] - m[t, 3], m[t, 4] - m[t, 7], m[t + 1, 0] - m[t, 0]])
    d = np.array(rows)
    print(nm, "tiles", len(ok), "cycles per tile (mean):", int(d[:, -1].mean()))
    for i, n in enumerate(names):
        print("   %-52s %7.0f" % (n, d[:, i].mean()))
