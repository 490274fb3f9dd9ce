# one-time cost of the first steps of a never-repeating run (bucket capture):  python3 tools/experiments/capture_cost.py [mols] [set] [model]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from geossl_amd import _lib
_lib.load()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
molset = sys.argv[2] if len(sys.argv) > 2 else "B"
model = sys.argv[3] if len(sys.argv) > 3 else "schnet"
for rep in range(2):
    wl = bench.Workload(dev, 0, 1, mols=mols, molset=molset, api="trainer", model=model, n_batches=40, distinct=True)
    torch.cuda.synchronize()
    ts = []
    for i in range(40):
        t0 = time.perf_counter()
        wl.step(i)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print("rep %d: first steps ms: %s ... median of the rest %.2f" % (rep, " ".join("%.1f" % t for t in ts[:6]), float(np.median(ts[6:]))))
