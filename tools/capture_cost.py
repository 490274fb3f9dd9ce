# what the first step on a ragged loader costs (bucket creation, warm-up passes, capture): python tools/capture_cost.py [mols]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd import bucket as bk
from geossl_amd.Geom3D.models import SchNet
from geossl_amd.NCSN import NCSN_version_03
from geossl_amd.synthetic import collate_subset, make_batch
dev = "cuda:0"
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = SchNet(128, 128, 6, 51, 5.0, node_class=9).to(dev)
n1 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
n2 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
tr = pg.DDMTrainer(model, n1, n2, use_graph=True)
pool = make_batch(max(4 * mols, 2048), seed=1, mode="B")
rng = np.random.default_rng(5)
bts = [pg.Batch.from_numpy(collate_subset(pool, rng.permutation(len(pool["sizes"]))[:mols]), dev, prepare=False) for _ in range(40)]
torch.cuda.synchronize()
marks = {}
orig = {}
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
wrap(tr.step_graphs, "_capture")
wrap(tr.step_graphs, "_capture_bucket")
orig_bucket = bk.Bucket
class TB(orig_bucket):
    def __init__(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        super().__init__(*a, **k)
        torch.cuda.synchronize(); marks["Bucket()"] = time.perf_counter() - t0
bk.Bucket = TB
for i, bt in enumerate(bts):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(bt)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if i < 3 or i == len(bts) - 1:
        print("step %d: %.2f ms" % (i, 1e3 * dt), {k: round(1e3 * v, 2) for k, v in marks.items()})
