# Does the 256 MB Infinity Cache keep a filter tensor (160 MB) between the kernel that wrote or read it and the aggregation that
# reads it next?  Aggregation over six different buffers, over one buffer six times, and right after a copy into the buffer.
#   python tools/probes/agg_cache_probe.py
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import ops
from geossl_amd.layout import MolLayout
from geossl_amd.synthetic import make_batch
dev, F, mols = "cuda:0", 128, 1024
sizes = list(make_batch(mols, seed=3, mode="A")["sizes"]) * 2
batch = torch.arange(len(sizes), device=dev).repeat_interleave(torch.tensor(sizes, device=dev))
lay = MolLayout(batch, len(sizes), sizes=sizes)
x = torch.randn(lay.N, F, device=dev); flag = torch.randint(0, 4, (lay.P,), device=dev, dtype=torch.uint8); o = torch.empty_like(x)
Ws = [torch.randn(lay.P, F, device=dev) for _ in range(6)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def six_diff():
    for w in Ws: ops.aggregate(x, w, flag, lay, out=o)
def six_same():
    for _ in range(6): ops.aggregate(x, Ws[0], flag, lay, out=o)
def write_then_read():   # write Wf (160 MB) with a copy kernel, then aggregate from it
    for w in Ws:
        Ws[0].copy_(w)   # reads 160 MB, writes 160 MB
        ops.aggregate(x, Ws[0], flag, lay, out=o)
def write_only():
    for w in Ws: Ws[0].copy_(w)
print(json.dumps({"six_different_us_per_launch": timeit(six_diff) / 6, "same_buffer_us_per_launch": timeit(six_same) / 6,
                  "copy_then_aggregate_us_per_pair": timeit(write_then_read) / 6, "copy_only_us": timeit(write_only) / 6}))
