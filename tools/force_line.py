import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
from geossl_amd import _lib
_lib.load()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
for bb in ("schnet", "painn"):
    for ug in (True, False):
        r = bench.force_training_line(dev, bb, use_graph=ug, steps=20 if ug else 8)
        print(bb, ug, {k: v for k, v in r.items() if k != "workload"}, flush=True)
