# Aggregation kernel on a ragged (set B) two-view batch: HIP-event time per launch, bytes of filter rows per launch.
#   python tools/bench_aggregate.py [mols]       (GEOSSL_HIP_LIB selects an alternative build)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import ops
from geossl_amd.layout import MolLayout
from geossl_amd.synthetic import make_batch
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev, F = "cuda:0", 128
out = {}
import numpy as np
for mode in ("A", "B", "U22", "U26", "M18_22"):
    if mode in ("A", "B"):
        sizes = list(make_batch(mols, seed=3, mode=mode)["sizes"]) * 2       # two views
    elif mode == "M18_22":   # two classes only, interleaved
        sizes = [18, 22] * mols
    else:                    # uniform molecules of one large class (one code path, like set A)
        sizes = [int(mode[1:])] * (2 * mols)
    batch = torch.arange(len(sizes), device=dev).repeat_interleave(torch.tensor(sizes, device=dev))
    lay = MolLayout(batch, len(sizes), sizes=sizes)
    x = torch.randn(lay.N, F, device=dev)
    W = torch.randn(lay.P, F, device=dev)
    flag = torch.randint(0, 4, (lay.P,), device=dev, dtype=torch.uint8)
    o = torch.empty_like(x)
    Ws = [torch.randn(lay.P, F, device=dev) for _ in range(6)]           # six layers: no reuse from cache
    def run():
        for w in Ws:
            ops.aggregate(x, w, flag, lay, out=o)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 120 * 1e3
    out[mode] = {"us": round(us, 1), "work_items": None if lay.agg_work is None else int(lay.agg_work.numel()),
                 "MB": round(lay.P * F * 4 / 1e6, 1), "TBps": round(lay.P * F * 4 / us / 1e6, 2)}
print(json.dumps(out))
