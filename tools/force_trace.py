# one model's force evaluation (finetune_md17.py:85-105: energy head, pred_force = -grad(E, positions)) or - mode "train" /
# "train-reference" - train-on-forces steps (finetune_md17.py:30-54: loss on energy and force, backward, optimiser) for a
# kernel trace:  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/force_trace.py schnet|painn [mols] [steps] [mode]
# then tools/aten_in_trace.py <dir> lists every kernel of the trace that is not one of this library's.
#   train            the step on the library's pieces end to end: Dense energy head, ops.energy_force_loss (:46-51 as one
#                    node), parameters in one flat buffer with their gradients added in place (_lib.direct_grads), fused Adam
#   train-reference  the reference's lines as written: torch L1Loss arithmetic, loss.backward(), torch.optim.Adam - what is
#                    left of ATen is the caller's own arithmetic, the engine's gradient accumulation and the optimiser
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
mode = sys.argv[4] if len(sys.argv) > 4 else "eval"
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
    head = (torch.nn.Linear(128, 1) if mode == "train-reference" else Dense(128, 1)).to(dev)   # finetune_md17.py:33
    rep_of = lambda pos: model(bt.x[:, 0], pos, bt.batch)
minus_one = torch.full((mols, 1), -1.0, device=dev)
import time
if mode != "eval":
    from geossl_amd import _lib
    from geossl_amd.optim import FlatParams, FusedAdam
    g = torch.Generator().manual_seed(1)
    y_e = torch.randn(mols, generator=g).to(dev)
    y_f = torch.randn(bt.positions.shape, generator=g).to(dev)
    ones = torch.ones(mols, device=dev)
    import contextlib
    X = os.environ.get("FT_VARIANT", "")   # experiments: "noflat" (torch Adam on the parameters as they are), "nodirect", "allinputs"
    if mode == "train" and "noflat" not in X:
        flat = FlatParams([model, head])
        opt = FusedAdam(flat, lr=5e-4)
        train_params = flat.trainable
    elif mode == "train":
        train_params = [p for p in list(model.parameters()) + list(head.parameters()) if p.requires_grad]
        opt = torch.optim.Adam(train_params, lr=5e-4)
    else:
        params = list(model.parameters()) + list(head.parameters())
        opt = torch.optim.Adam(params, lr=5e-4)
        criterion = torch.nn.L1Loss()                                # finetune_md17.py:236
    for step in range(steps):
        if step == steps // 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        pos = bt.positions.detach().requires_grad_(True)             # :33
        energy = head(rep_of(pos)).squeeze(1)                        # :36-44
        if mode == "train":
            dE = torch.autograd.grad(energy, pos, grad_outputs=ones, create_graph=True, retain_graph=True)[0]   # :46
            loss = ops.energy_force_loss(energy, y_e, dE, y_f, 0.05, 0.95, "l1")                                # :46-51
            opt.zero_grad()
            with (contextlib.nullcontext() if "nodirect" in X else _lib.direct_grads()):
                loss.backward(inputs=None if "allinputs" in X else train_params)   # :54 (the positions are not trained)
            opt.step()
        else:
            force = -torch.autograd.grad(outputs=energy, inputs=pos, grad_outputs=torch.ones_like(energy),
                                         create_graph=True, retain_graph=True)[0]
            loss = 0.05 * criterion(energy, y_e) + 0.95 * criterion(force, y_f)
            opt.zero_grad()
            loss.backward()
            opt.step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (steps - steps // 2)
    print(which, mode, "loss %.6f" % float(loss), "%.3f ms per step = %.1f k molecules/s" % (ms, mols / ms))
    sys.exit(0)
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
