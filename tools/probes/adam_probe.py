"""tools/probes/adam_probe.hip against torch.optim.Adam (default foreach path, single-tensor, fused=True) on the GPU, bit
for bit, state tensor by state tensor."""
import ctypes as C, math, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "adam_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "adam_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.adam_probe.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 5 + [C.c_int, C.c_float, C.c_void_p]
torch.manual_seed(0)
n, lr, b1, b2, eps = 520324, 5e-4, 0.9, 0.999, 1e-8
p0 = torch.randn(n, device="cuda") * 0.1
gs = [torch.randn(n, device="cuda") * (10.0 ** (-k)) for k in range(4)]
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, kw in (("foreach", dict(foreach=True)), ("single", dict(foreach=False)), ("fused", dict(fused=True))):
    ps = [torch.nn.Parameter(p0.clone())]
    opt = torch.optim.Adam(ps, lr=lr, betas=(b1, b2), eps=eps, **kw)
    for g in gs[:nsteps]:
        ps[0].grad = g.clone()
        opt.step()
    st = opt.state[ps[0]]
    ref = (ps[0].detach(), st["exp_avg"], st["exp_avg_sq"])
    for mode in (5, 7, 21, 23, 17, 19):
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for step, g in enumerate(gs[:nsteps], 1):
            bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
            lib.adam_probe(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr / bc1, 1 - b1, b2, eps, math.sqrt(bc2), mode, 1 - b2,
                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print(name, "steps", nsteps, "mode", mode, "mismatches p/m/v", int((p != ref[0]).sum()), int((m != ref[1]).sum()), int((v != ref[2]).sum()))
