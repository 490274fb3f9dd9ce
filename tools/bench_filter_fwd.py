# Filter-network forward kernel alone at the bench size (two views x 1024 molecules x 153 pair slots, L = 6, F = 128, G = 51):
# HIP-event time per launch with and without the saved hidden rows T.   (GEOSSL_HIP_LIB selects an alternative build)
#   python tools/bench_filter_fwd.py [P]
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geossl_amd import _lib
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * 1024 * 153
L, F, G, dev = 6, 128, 51, "cuda:0"
lib = _lib.load()
torch.manual_seed(0)
ZERO = os.environ.get("FF_ZERO_DATA") is not None   # same instruction stream on all-zero operands: is the clock data dependent?
w1 = [torch.randn(F, G, device=dev) * 0.2 for _ in range(L)]
b1 = [torch.randn(F, device=dev) * 0.1 for _ in range(L)]
w2 = [torch.randn(F, F, device=dev) * 0.1 for _ in range(L)]
b2 = [torch.randn(F, device=dev) * 0.1 for _ in range(L)]
if ZERO:
    for t in w1 + b1 + w2 + b2:
        t.zero_()
fw = _lib.FilterWeights()
for l in range(L):
    fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = w1[l].data_ptr(), b1[l].data_ptr(), w2[l].data_ptr(), b2[l].data_ptr()
d = torch.rand(P, device=dev) * 5.0 if not ZERO else torch.full((P,), 100.0, device=dev)
c = 0.5 * (torch.cos(d * 3.14159265 / 5.0) + 1.0)
offset = torch.linspace(0, 5.0, G, device=dev)
coeff = -0.5 / float(offset[1] - offset[0]) ** 2
Wf = torch.empty(L, P, F, device=dev)
T = torch.empty(L, P, F, device=dev)
st = torch.cuda.current_stream().cuda_stream
out = {}
for name, tp in (("with_T", T.data_ptr()), ("without_T", None)):
    run = lambda: lib.geossl_cfconv_filter_fwd(d.data_ptr(), c.data_ptr(), P, C.byref(fw), L, F, G, offset.data_ptr(), coeff, tp,
                                               Wf.data_ptr(), st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = L * P * F * 4 * (2 if tp else 1)
    out[name] = {"us": round(us, 1), "written_TBps": round(nbytes / us / 1e6, 2)}
# reference value of one row for a sanity check against fp64
i = 12345 % P
rbf = torch.exp(coeff * (d[i].double() - offset.double()) ** 2)
u = w1[2].double() @ rbf + b1[2].double()
t = torch.nn.functional.softplus(u) - 0.6931471805599453
ref = (w2[2].double() @ t + b2[2].double()) * c[i].double()
out["max_rel_err_row"] = float((Wf[2, i].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
out["checksum"] = float(Wf.double().sum())
print(json.dumps(out))
