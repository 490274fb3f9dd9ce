# PaiNN interaction kernels alone at the bench size (two views x 1024 molecules x 18 atoms, F = 128, R = 20): HIP-event time
# per launch, vector kernels (painn.hip) against the matrix-pipe kernels (painn_mma.hip).   python tools/bench_painn_inter.py [set]
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from geossl_amd import _lib
import test_gpu_round3 as t3
from geossl_amd.synthetic import molecule_sizes
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "A"
sizes = molecule_sizes(2048, mode, np.random.default_rng(0)).tolist()
c = t3._painn_edge_case(sizes, seed=1, R=20)
lay, el, N, Fd, R = c["lay"], c["el"], c["N"], 128, 20
st = _lib.stream()
inc_ptr, inc_idx = el.inc["i"]
q2, mu2 = torch.empty_like(c["q"]), torch.empty_like(c["mu"])
row_edge, grp_atom, grp_ptr, mol_grp = el.groups("i", lay.mol_ptr)
P = lambda t: t.data_ptr()


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


res = {"N": N, "E": el.E, "groups_rows_per_edge": float(4 * int(grp_ptr[-1]) / max(el.E, 1))}
res["fwd_mol_us"] = timed(lambda: _lib.call("geossl_painn_interaction_fwd_mol", P(c["q"]), P(c["mu"]), P(c["xc"]), P(el.idx_j),
    P(inc_ptr), P(inc_idx), P(c["phi"]), P(c["fcut"]), P(c["dirv"]), P(c["Wf"]), P(c["bf"]), P(lay.mol_ptr), lay.B, lay.max_n, N,
    Fd, R, P(q2), P(mu2), st))
res["fwd_mma_us"] = timed(lambda: _lib.call("geossl_painn_interaction_fwd_mma", P(c["q"]), P(c["mu"]), P(c["xc"]), P(el.idx_j),
    P(row_edge), P(grp_atom), P(mol_grp), P(c["phi"]), P(c["fcut"]), P(c["dirv"]), P(c["Wf"]), P(c["bf"]), P(lay.mol_ptr), lay.B,
    lay.max_n, N, Fd, R, P(q2), P(mu2), st))
res["fwd_mma_mu_zero_us"] = timed(lambda: _lib.call("geossl_painn_interaction_fwd_mma", P(c["q"]), None, P(c["xc"]), P(el.idx_j),
    P(row_edge), P(grp_atom), P(mol_grp), P(c["phi"]), P(c["fcut"]), P(c["dirv"]), P(c["Wf"]), P(c["bf"]), P(lay.mol_ptr), lay.B,
    lay.max_n, N, Fd, R, P(q2), P(mu2), st))   # the first interaction: mu identically zero (NULL)
print(json.dumps(res))
