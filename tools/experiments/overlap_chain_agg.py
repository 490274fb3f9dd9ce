# Does a latency-bound chain launch on half of the rows run BESIDE an HBM-bound aggregation launch on the other half of
# the molecules (two streams), and what does the pair cost against the two full-size launches back to back?
#   GEOSSL_CHAIN_SLOTS=256 python tools/experiments/overlap_chain_agg.py      (chain capped to one block per CU)
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import _lib, ops
from geossl_amd.layout import MolLayout
dev, F, B, n = "cuda:0", 128, 2048, 18
N = B * n
torch.manual_seed(0)
batch = torch.arange(B, device=dev).repeat_interleave(n)
lay = MolLayout(batch, B)
P = lay.P
X = torch.randn(N, F, device=dev)
Wf = torch.randn(P, F, device=dev)
flag = torch.full((P,), 3, dtype=torch.uint8, device=dev)
Ws = [torch.randn(F, F, device=dev) / F ** 0.5 for _ in range(3)]
b = torch.randn(F, device=dev)
res = torch.randn(N, F, device=dev)
img = ops.prepare_chain(Ws)
outs = [torch.empty(N, F, device=dev) for _ in range(4)]
agg_out = torch.empty(N, F, device=dev)
h = N // 2


def chain(a0, a1):
    r = lambda t: t[a0:a1]
    ops.linear_chain(r(X), [dict(image=img[0], bias=b, flags=_lib.EPI_SSP, out=r(outs[0])),
                            dict(image=img[1], bias=b, res=r(res), out=r(outs[1])), dict(image=img[2], out=r(outs[2]))])


def agg(m0, m1):
    ops.aggregate(X, Wf, flag, lay, out=agg_out, mols=None if (m0, m1) == (0, B) else (m0, m1, None))


side = torch.cuda.Stream()


def timed(fn, n=40):
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


def serial_full():
    chain(0, N)
    agg(0, B)


def serial_halves():
    chain(0, h); agg(0, B // 2); chain(h, N); agg(B // 2, B)


def overlapped():  # phase 1: chain(A) || agg(B);  phase 2: agg(A) || chain(B)
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    chain(0, h)
    with torch.cuda.stream(side):
        agg(B // 2, B)
    main.wait_stream(side)
    side.wait_stream(main)
    agg(0, B // 2)
    with torch.cuda.stream(side):
        chain(h, N)
    main.wait_stream(side)


out = {"chain_full_us": timed(lambda: chain(0, N)), "agg_full_us": timed(lambda: agg(0, B)),
       "chain_half_us": timed(lambda: chain(0, h)), "agg_half_us": timed(lambda: agg(0, B // 2)),
       "serial_full_us": timed(serial_full), "serial_halves_us": timed(serial_halves), "overlapped_us": timed(overlapped),
       "slots": os.environ.get("GEOSSL_CHAIN_SLOTS", "512")}
print(json.dumps(out))
