# Fused CFConv (filter network + aggregation of one block, geossl_cfconv_fused) against geossl_cfconv_filter_fwd +
# geossl_cfconv_aggregate at the bench size (two views x 1024 molecules x 18 atoms, F = 128, G = 51), HIP-event time per
# layer.   python tools/bench_fused.py [mols] [set]      (GEOSSL_HIP_LIB selects an alternative build)
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from geossl_amd import _lib, ops
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.layout import MolLayout
from geossl_amd.synthetic import make_batch
mols = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mode = sys.argv[2] if len(sys.argv) > 2 else "A"
L, F, G, dev = 6, 128, 51, "cuda:0"
lib = _lib.load()
torch.manual_seed(0)
b = make_batch(mols, seed=0, mode=mode)
bt = pg.Batch.from_numpy(b, dev)
lay = MolLayout(bt.batch, mols, sizes=list(b["sizes"]))
w = [(torch.randn(F, G, device=dev) * 0.2, torch.randn(F, device=dev) * 0.1, torch.randn(F, F, device=dev) * 0.1,
      torch.randn(F, device=dev) * 0.1) for _ in range(L)]
fw = _lib.FilterWeights()
for l in range(L):
    fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = (t.data_ptr() for t in w[l])
offset = torch.linspace(0, 5.0, G, device=dev)
coeff = -0.5 / float(offset[1] - offset[0]) ** 2
pair_d, pair_c, pair_flag = ops.pair_geometry(bt.positions, lay, 5.0)
P, N = lay.P, bt.positions.size(0)
T = torch.empty(L, P, F, device=dev)
Wf = torch.empty(L, P, F, device=dev)
x = torch.randn(N, F, device=dev)
out = torch.empty(N, F, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {"P": P, "N": N}
res["filter_fwd_all_layers_us"] = timed(lambda: lib.geossl_cfconv_filter_fwd(
    pair_d.data_ptr(), pair_c.data_ptr(), P, C.byref(fw), L, F, G, offset.data_ptr(), coeff, T.data_ptr(), Wf.data_ptr(), st))
res["aggregate_us"] = timed(lambda: ops.aggregate(x, Wf[0], pair_flag, lay, out=out))
ref = ops.aggregate(x, Wf[0], pair_flag, lay).clone()
res["unfused_per_layer_us"] = res["filter_fwd_all_layers_us"] / L + res["aggregate_us"]
images = ops.cfconv_fused_prepare(fw, L, F, G, offset, dev)
res["prepare_us"] = timed(lambda: ops.cfconv_fused_prepare(fw, L, F, G, offset, dev))
T2 = torch.empty(P, F, device=dev)
W2 = torch.empty(P, F, device=dev)
for name, kw in (("fused_noT", {}), ("fused_T", dict(T_l=T2)), ("fused_T_Wf", dict(T_l=T2, Wf_l=W2)),
                 ("fused_from_t", dict(T_l=T[0], from_t=True))):
    res[name + "_us"] = timed(lambda: ops.cfconv_fused(x, images[0], pair_d, pair_c, pair_flag, lay, coeff, out, **kw))
    res[name + "_err"] = float((out - ref).abs().max() / ref.abs().max())
print(json.dumps(res))
