# Does the aggregation run faster when its filter rows (160 MB per layer at the bench shape) are still in the 256 MB
# Infinity Cache?  hot: the SAME layer's rows every launch; cold: six layers in turn (963 MB between two visits);
# written-then-read: a device copy into the rows (what a per-layer filter forward would leave behind) right before.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import ops
from geossl_amd.layout import MolLayout
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev, F = "cuda:0", 128
sizes = [18] * (2 * mols)
batch = torch.arange(len(sizes), device=dev).repeat_interleave(torch.tensor(sizes, device=dev))
lay = MolLayout(batch, len(sizes), sizes=sizes)
x = torch.randn(lay.N, F, device=dev)
flag = torch.randint(0, 4, (lay.P,), device=dev, dtype=torch.uint8)
o = torch.empty_like(x)
Ws = [torch.randn(lay.P, F, device=dev) for _ in range(6)]
src = torch.randn(lay.P, F, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cold():
    for w in Ws:
        ops.aggregate(x, w, flag, lay, out=o)


def hot():
    for _ in range(6):
        ops.aggregate(x, Ws[0], flag, lay, out=o)


def copy_only():
    for w in Ws:
        w.copy_(src)


def written_then_read():
    for w in Ws:
        w.copy_(src)
        ops.aggregate(x, w, flag, lay, out=o)


c, h, co, wr = timed(cold) / 6, timed(hot) / 6, timed(copy_only) / 6, timed(written_then_read) / 6
print("rows per layer %.0f MB; aggregation per launch: cold %.1f us, hot (same rows) %.1f us; copy into the rows %.1f us, "
      "copy + aggregation %.1f us (aggregation after a fresh write: %.1f us)" % (lay.P * F * 4 / 1e6, c, h, co, wr, wr - co))
