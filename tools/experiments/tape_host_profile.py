# host-side cost of the second-order route: tape.second_order called directly (main thread) under cProfile, and timed
# against the torch-graph route on the same inputs:  python3 tools/experiments/tape_host_profile.py schnet|painn [mols]
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import higher_order as ho, ops, tape as tp
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.Geom3D.models import PaiNN, SchNet
from geossl_amd.synthetic import make_batch
which = sys.argv[1] if len(sys.argv) > 1 else "schnet"
mols = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
torch.manual_seed(0)
bt = pg.Batch.from_numpy(make_batch(mols, seed=3, mode="B"), dev)
bt.x[:, 0].clamp_(min=1, max=8)
pos = bt.positions.detach().requires_grad_(True)
if which == "painn":
    model = PaiNN(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add").to(dev)
    bt.radius_edge_index = ops.radius_graph(bt.positions, 5.0, bt.batch)
    out = model(bt.x, pos, bt.radius_edge_index, bt.batch)
else:
    model = SchNet(128, 128, 6, 51, 5.0, node_class=9, readout="add").to(dev)
    out = model(bt.x[:, 0], pos, bt.batch)
fctx = out.grad_fn
while fctx is not None and not hasattr(fctx, "cfg"):
    fctx = fctx.next_functions[0][0] if fctx.next_functions else None
params = [p.detach() for p in fctx.params]
N = pos.size(0)
if which == "painn":
    el = fctx.el
    feats_tape = lambda x, ps: tp.painn_atom_features(fctx.z, x, el.idx_i, el.idx_j, fctx.cfg, ps)
    feats_torch = lambda x, ps: ho.painn_atom_features(fctx.z, x, el.idx_i, el.idx_j, fctx.cfg, ps)
else:
    feats_tape = lambda x, ps: tp.schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps)
    feats_torch = lambda x, ps: ho.schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps)
dh = torch.randn(N, 128, device=dev)
mask = [True] + [False] * len(params)
need = [True, True] + [True] * len(params)
cot = [torch.randn(N, 3, device=dev)]
p0 = bt.positions.detach()


def run(route):
    os.environ["GEOSSL_SECOND_ORDER"] = route
    with torch.no_grad():
        return ho._second_order(feats_torch, feats_tape, mask, need, cot, dh, p0, params)


for route in ("tape", "torch"):
    for _ in range(3):
        run(route)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        run(route)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s %s %d molecules: %.2f ms per call (host %.2f ms)" % (which, route, mols, 1e2 * (t2 - t0), 1e2 * (t1 - t0)))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    run("tape")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
