# Microbenchmark of geossl_linear_chain against the one-launch-per-layer path (same box, HIP events).
#   python tools/bench_chain.py [R] [lib ...]      libs: alternative builds of libgeossl_hip.so (ablation variants)
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 36864
libs = sys.argv[2:]
if libs and not os.environ.get("_CHAIN_CHILD"):
    for lib in [None] + libs:
        env = dict(os.environ, _CHAIN_CHILD="1")
        if lib:
            env["GEOSSL_HIP_LIB"] = lib
        print("==", lib or "default", flush=True)
        subprocess.run([sys.executable, __file__, str(R)], env=env)
    sys.exit(0)
import torch
from geossl_amd import _lib, ops
dev, F = "cuda:0", 128
torch.manual_seed(0)
X = torch.randn(R, F, device=dev)
Ws = [torch.randn(F, F, device=dev) / F ** 0.5 for _ in range(4)]
b = torch.randn(F, device=dev)
res, tp = torch.randn(R, F, device=dev), torch.randn(R, F, device=dev)
imgs = ops.prepare_chain(Ws)
pw = ops.prepare_linear(Ws)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st = [dict(image=imgs[0], bias=b, flags=_lib.EPI_SSP), dict(image=imgs[1], bias=b, res=res), dict(image=imgs[2]),
      dict(image=imgs[3], bias=b)]
out = {}
for n in (1, 2, 3):
    out["chain%d_us" % n] = round(timeit(lambda: ops.linear_chain(X, st[:n])), 1)


def per_layer():
    t = ops.linear(X, pw[0], bias=b, flags=_lib.EPI_SSP)
    h = ops.linear(t, pw[1], bias=b, res=res)
    return ops.linear(h, pw[2])


out["three_launches_us"] = round(timeit(per_layer), 1)
out["one_launch_us"] = round(timeit(lambda: ops.linear(X, pw[2])), 1)
print(json.dumps(out), flush=True)
