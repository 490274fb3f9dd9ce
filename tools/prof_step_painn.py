# one-process profile target for rocprofv3: a few eager PaiNN DDM steps at BASELINE config 5
# (bs = 1024 molecules x 18 atoms, PaiNN F=128 L=3 rbf=20 cutoff 5 A + two NCSN heads, fp32)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import ops, pretrain_GeoSSL as pg
from geossl_amd.Geom3D.models import PaiNN
from geossl_amd.NCSN import NCSN_version_03
from geossl_amd.synthetic import make_batch, draw_noise
dev = 'cuda:0'
torch.manual_seed(0)
model = PaiNN(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add").to(dev)
n1 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
n2 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
tr = pg.DDMTrainer(model, n1, n2, model_3d="painn")
b = make_batch(1024, seed=0)
bt = pg.Batch.from_numpy(b, dev)
bt.radius_edge_index = ops.radius_graph(bt.positions, 5.0, bt.batch)
nz = {k: torch.from_numpy(v).to(dev) for k, v in draw_noise(b, 1).items()}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(steps):
    loss = tr.step(bt, nz)
torch.cuda.synchronize()
print("loss", float(loss))
