# one model's force evaluation (finetune_md17.py:85-105: energy head, pred_force = -grad(E, positions)) for a kernel
# trace:  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/force_trace.py schnet|painn [mols] [steps]
# then tools/aten_in_trace.py <dir> lists every kernel of the trace that is not one of this library's.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import ops
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.Geom3D.models import PaiNN, SchNet
from geossl_amd.synthetic import make_batch
which = sys.argv[1] if len(sys.argv) > 1 else "painn"
mols = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
torch.manual_seed(0)
bt = pg.Batch.from_numpy(make_batch(mols, seed=3, mode="B"), dev)
bt.x[:, 0].clamp_(max=8)
if which == "painn":
    model = PaiNN(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add").to(dev)
    head = model.create_output_layers().to(dev)
    bt.radius_edge_index = ops.radius_graph(bt.positions, 5.0, bt.batch)
    rep_of = lambda pos: model(bt.x, pos, bt.radius_edge_index, bt.batch)
else:
    model = SchNet(128, 128, 6, 51, 5.0, node_class=9, readout="add").to(dev)
    from geossl_amd.Geom3D.models.painn import Dense
    head = Dense(128, 1).to(dev)
    rep_of = lambda pos: model(bt.x[:, 0], pos, bt.batch)
minus_one = torch.full((mols, 1), -1.0, device=dev)
import time
for step in range(steps):
    if step == steps // 2:  # second half timed (the first pays the one-time preparation)
        torch.cuda.synchronize(); t0 = time.perf_counter()
    pos = bt.positions.detach().requires_grad_(True)
    energy = head(rep_of(pos))                                   # finetune_md17.py:38-44
    # pred_force = -grad(E, pos, grad_outputs=ones) (:46): the sign rides on grad_outputs
    force = torch.autograd.grad(energy, pos, grad_outputs=minus_one, create_graph=True, retain_graph=True)[0].detach_()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / (steps - steps // 2)
print(which, "force", tuple(force.shape), float(force.abs().max()), "%.3f ms per evaluation = %.1f k molecules/s" % (ms, mols / ms))
