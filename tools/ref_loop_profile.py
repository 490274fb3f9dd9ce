# where the wall time of the reference-shaped loop goes (host side), one GPU:
#   python tools/ref_loop_profile.py [mols] [A|B] [distinct]
# phases are timed on the host WITHOUT synchronising (the loop's own loss.item() is its only sync, as in the reference)
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.Geom3D.models import SchNet
from geossl_amd.NCSN import NCSN_version_03
from geossl_amd.synthetic import collate_subset, make_batch
dev = "cuda:0"
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
molset = sys.argv[2] if len(sys.argv) > 2 else "A"
distinct = len(sys.argv) > 3 and sys.argv[3] == "distinct"
torch.manual_seed(0)
model = SchNet(128, 128, 6, 51, 5.0, node_class=9).to(dev)
n1 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
n2 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(dev)
pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
args = types.SimpleNamespace(model_3d="schnet", lr=5e-4, decay=0.0)
opt = torch.optim.Adam([{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()},
                        {"params": n2.parameters()}], lr=args.lr, weight_decay=0.0)
steps, warm = 60, 10
if distinct:
    pool = make_batch(max(4 * mols, 2048), seed=1, mode=molset)
    rng = np.random.default_rng(5)
    bts = [pg.Batch.from_numpy(collate_subset(pool, rng.permutation(len(pool["sizes"]))[:mols]), dev, prepare=False)
           for _ in range(steps + warm)]
else:
    bts = [pg.Batch.from_numpy(make_batch(mols, seed=i, mode=molset), dev) for i in range(4)]
acc = {}
def mark(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
torch.cuda.synchronize()
if os.environ.get("SINGLE_THREAD_BACKWARD"):
    torch.autograd.set_multithreading_enabled(False)
for step in range(steps + warm):
    if step == warm:
        acc.clear()
        torch.cuda.synchronize()
        t_all = time.perf_counter()
    bt = bts[step % len(bts)].to(dev)
    t0 = time.perf_counter(); loss, _ = pg.do_DDM(args, bt, model, mu=0.0, sigma=0.3); mark("do_DDM (host)", t0)
    t0 = time.perf_counter(); v = loss.detach().item(); mark("loss.item() (waits for the forward)", t0)
    t0 = time.perf_counter(); opt.zero_grad(); mark("zero_grad", t0)
    t0 = time.perf_counter(); loss.backward(); mark("loss.backward() (host)", t0)
    t0 = time.perf_counter(); opt.step(); mark("optimizer.step() (host)", t0)
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
for k, v in acc.items():
    print("%-40s %.3f ms/step" % (k, 1e3 * v / steps))
print("%-40s %.3f ms/step  (%.1f k molecules/s)" % ("wall", 1e3 * wall / steps, mols * steps / wall / 1e3))
eng = model.__dict__["_geossl_autograd_step"]
print("captures", sum(sg.captures for sg in eng.graphs.values()), "fused adam plan", bool(opt.__dict__.get("_geossl_plan")))
# the pieces of do_DDM's host side
sg = list(eng.graphs.values())[0]
bt = bts[0]
g = sg.lookup(bt)
def timeit(name, fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("  %-38s host %.3f ms  (with GPU %.3f ms)" % (name, 1e3 * (t1 - t0) / n, 1e3 * (time.perf_counter() - t0) / n))
timeit("lookup", lambda: sg.lookup(bt))
timeit("refresh (inputs into the graph)", lambda: sg.refresh(g, bt))
timeit("draw_step_noise host pos noise", lambda: pg.draw_step_noise(bt, n1, n2, 0.0, 0.3, False, None, into=sg.noise_views(g)))
timeit("graph replay fwd", lambda: g["graph"].replay())
timeit("graph replay bwd", lambda: g["graph_bwd"].replay())
timeit("_ReplayedLoss.apply", lambda: pg._ReplayedLoss.apply(g["loss"].clone(), eng, pg._Ticket(serial=0, event=None, g=eng.gflat, used=True), *eng.params))
