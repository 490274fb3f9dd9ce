# kernel trace of the captured train-on-forces step: python3 tools/force_graph_trace.py <schnet|painn> [mols] [steps]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from geossl_amd import _lib
_lib.load()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
bb = sys.argv[1] if len(sys.argv) > 1 else "schnet"
mols = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = bench.force_training_line(dev, bb, mols=mols, steps=steps, warmup=4)
print({k: v for k, v in r.items() if k != "workload"})
