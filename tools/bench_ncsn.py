# Microbenchmark of the NCSN head (forward, backward) on synthetic batches of set A / set B (HIP events, same box).
#   python tools/bench_ncsn.py [mols]
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd.NCSN import NCSN_version_03
from geossl_amd.synthetic import make_batch
from geossl_amd.Geom3D.dataloaders.dataloaders_AtomTuple import BatchAtomTuple
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
out = {}
for mode in ("A", "B"):
    b = make_batch(mols, seed=3, mode=mode)
    x, pos = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["positions"]).to(dev)
    data = BatchAtomTuple.from_sizes(x, pos, b["sizes"].tolist(), option="combination")
    sei = data.super_edge_index
    S, N = sei.size(1), x.size(0)
    torch.manual_seed(0)
    head = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2.0).to(dev)
    h = (float(os.environ.get("H_SCALE", "1")) * torch.randn(N, 128, device=dev)).requires_grad_()
    dist = (pos[sei[0]] - pos[sei[1]]).norm(dim=-1, keepdim=True)
    nl = torch.randint(0, 50, (mols,), device=dev)
    dn = torch.randn(S, 1, device=dev)

    def fwd():
        return head(data, h, dist, noise_level=nl, distance_noise=dn)

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / n * 1e3, 1)

    def both():
        loss = fwd()
        loss.backward()

    with torch.no_grad():
        t_f = timeit(fwd)
    t_fb = timeit(both)
    out[mode] = {"S": S, "N": N, "fwd_nograd_us": t_f, "fwd_bwd_us": t_fb}
print(json.dumps(out))
