# Microbenchmark of geossl_linear_wgrad at the bench shape (20 problems of 36 864 x 128 x 128), optionally against
# variant builds:  python tools/bench_wgrad.py [lib ...]
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = sys.argv[1:]
if libs and not os.environ.get("_WG_CHILD"):
    for lib in [None] + libs:
        env = dict(os.environ, _WG_CHILD="1")
        if lib:
            env["GEOSSL_HIP_LIB"] = lib
        print("==", lib or "default", flush=True)
        subprocess.run([sys.executable, __file__], env=env)
    sys.exit(0)
import torch
from geossl_amd import ops
dev, R, F, NP = "cuda:0", 36864, 128, 20
torch.manual_seed(0)
A = [torch.randn(R, F, device=dev) for _ in range(NP)]
B = [torch.randn(R, F, device=dev) for _ in range(NP)]
dW = [torch.zeros(F, F, device=dev) for _ in range(NP)]
db = [torch.zeros(F, device=dev) for _ in range(NP)]
probs = [(A[i], B[i], dW[i], db[i]) for i in range(NP)]
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t = timeit(lambda: ops.linear_wgrad(probs, R, F, F))
ref = A[3].double().T @ B[3].double()
err = float((dW[3].double() - ref).abs().max() / ref.abs().max())
print(json.dumps({"wgrad20_us": round(t, 1), "GBps": round(NP * R * 2 * F * 4 / t / 1e3, 1), "max_rel_err": err}))
