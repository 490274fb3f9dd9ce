# Where a full-size chain launch spends its time: the same three stages with parts of the epilogue traffic removed.
#   python tools/chain_probe.py [R]
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import _lib, ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 36864
dev, F = "cuda:0", 128
torch.manual_seed(0)
X = torch.randn(R, F, device=dev)
Ws = [torch.randn(F, F, device=dev) / F ** 0.5 for _ in range(3)]
b = torch.randn(F, device=dev)
res, tp = torch.randn(R, F, device=dev), torch.randn(R, F, device=dev)
imgs = ops.prepare_chain(Ws)
outs = [torch.empty(R, F, device=dev) for _ in range(3)]


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


def stages(ssp=True, use_res=True, store=(True, True, True), tprev=False):
    st = [dict(image=imgs[0], bias=b, flags=_lib.EPI_SSP if ssp else 0), dict(image=imgs[1], bias=b), dict(image=imgs[2])]
    if use_res:
        st[1]["res"] = res
    if tprev:
        st[0]["tprev"] = tp
        st[2]["tprev"] = tp
    for s in range(3):
        if store[s]:
            st[s]["out"] = outs[s]
        else:
            st[s]["store"] = False
    return st


cases = {
    "all": stages(),
    "no_ssp": stages(ssp=False),
    "no_res": stages(use_res=False),
    "store_last_only": stages(store=(False, False, True)),
    "no_res_store_last_only": stages(use_res=False, store=(False, False, True)),
    "no_res_no_ssp_store_last_only": stages(ssp=False, use_res=False, store=(False, False, True)),
    "tprev_twice": stages(tprev=True),
}
out = {"R": R}
for k, st in cases.items():
    out[k] = timeit(lambda: ops.linear_chain(X, st))
# two-stage and one-stage forms of the lightest variant
light = stages(ssp=False, use_res=False, store=(False, False, True))
out["light2"] = timeit(lambda: ops.linear_chain(X, [light[0], dict(light[2])]))
out["light1"] = timeit(lambda: ops.linear_chain(X, [dict(light[2])]))
print(json.dumps(out), flush=True)
