"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) the committed golden vectors produced by the unmodified reference and (2) the CPU oracle on the
same seeded inputs.  Tolerances (BASELINE.json north_star): radius-graph edge indices bit-exact;
fp32 outputs / loss <= 1e-5 relative; parameter gradients <= 1e-4 relative (fp32 reductions over up
to ~3e5 rows in a different order)."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, max_rel, rel_err
from helpers import (cfg_of, grad_summary, ncsn_oracle_params, product_ncsn, product_schnet, schnet_oracle_params, t,
                     unique_named_grads)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_OUT = 1e-5
TOL_GRAD = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from geossl_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "GPU tests need an MI355X"


# ------------------------------------------------------------------------------------------- graph (K1)
@pytest.mark.parametrize("r", [5.0, 10.0, 1.5])
def test_radius_graph_golden_bit_exact(r):
    from geossl_amd import ops
    g = load_golden("g3_radius_graph")
    e, w = ops.radius_graph(t(g["positions"], DEV), r, t(g["batch"], DEV), return_weight=True)
    assert e.dtype == torch.int64
    assert torch.equal(e.cpu(), t(g["edge_index_%g" % r]))
    assert max_rel(w.cpu(), g["edge_weight_%g" % r]) < 1e-6


@pytest.mark.parametrize("mode,r", [("A", 5.0), ("B", 5.0), ("B", 10.0)])
def test_radius_graph_full_batch_vs_oracle(mode, r):
    from geossl_amd import ops
    from geossl_amd.synthetic import make_batch
    from oracle.graph import radius_graph_np
    b = make_batch(1024, seed=7, mode=mode)
    e = ops.radius_graph(t(b["positions"], DEV), r, t(b["batch"], DEV))
    ref = radius_graph_np(b["positions"], r, b["batch"])
    assert np.array_equal(e.cpu().numpy(), ref)


def test_radius_graph_edge_cases():
    from geossl_amd import ops
    pos = torch.zeros(1, 3, device=DEV)
    assert ops.radius_graph(pos, 5.0, torch.zeros(1, dtype=torch.long, device=DEV)).shape == (2, 0)
    # exactly-at-cutoff pair is excluded (strict <), batch=None means one graph
    pos = torch.tensor([[0.0, 0, 0], [5.0, 0, 0], [0, 4.999999, 0]], device=DEV)
    e = ops.radius_graph(pos, 5.0)
    assert e.cpu().tolist() == [[2, 0], [0, 2]]
    with pytest.raises(ValueError):
        ops.radius_graph(torch.zeros(3, 3, device=DEV), 5.0, torch.tensor([1, 0, 0], device=DEV))


@pytest.mark.parametrize("option", ["combination", "permutation"])
def test_device_atom_tuple_extractor_bit_exact(option):
    """Device-side AtomTupleExtractor + collate offsets (N1) vs the oracle restatement of
    dataloaders_AtomTuple.py:15-37,64-65: same tuples, same order, incl. 0/1-atom molecules."""
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, BatchAtomTuple
    from oracle import graph
    sizes = [1, 2, 5, 18, 0, 29, 33, 1, 3]
    rng = np.random.default_rng(5)
    mols = [(rng.integers(0, 9, (n, 2)), rng.normal(size=(n, 3)).astype(np.float32)) for n in sizes if n > 0]
    ref = graph.collate_np(mols, option=option)
    nz = [n for n in sizes if n > 0]
    bt = BatchAtomTuple.from_sizes(t(ref["x"], DEV), t(ref["positions"], DEV), nz, option=option)
    assert torch.equal(bt.batch.cpu(), t(ref["batch"]))
    assert torch.equal(bt.super_edge_index.cpu(), t(ref["super_edge_index"]))
    assert bt.num_graphs == len(nz)
    sei = AtomTupleExtractor(option=option)(bt.batch)
    assert torch.equal(sei, bt.super_edge_index)
    assert bt.positions.is_cuda and bt["x"] is bt.x  # item access like torch_geometric.data.Data


def test_layout_from_host_sizes_equals_layout_from_device_scan():
    """Collation-time layouts (layout.prepare_batch: molecule sizes known on the host, nothing read back from the
    device) are the same structures the lazy path derives from the batch vector, and the step finds them."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.layout import MolLayout, get_super_edge_layout
    from geossl_amd.synthetic import make_batch
    b = make_batch(0, seed=9, sizes=[1, 2, 5, 18, 29, 33, 3, 1, 40])
    bat = t(b["batch"], DEV)
    a, c = MolLayout(bat), MolLayout(bat, sizes=list(b["sizes"]))
    assert (a.N, a.B, a.P, a.max_n) == (c.N, c.B, c.P, c.max_n)
    for k in ("mol_ptr", "pair_ptr", "pair_i", "pair_j", "order"):
        assert torch.equal(getattr(a, k), getattr(c, k)), k
    sizes_sorted = torch.tensor(b["sizes"])[a.order.cpu().long()]
    assert bool((sizes_sorted[:-1] >= sizes_sorted[1:]).all())  # largest molecule first
    with pytest.raises(ValueError):
        MolLayout(bat, sizes=list(b["sizes"])[:-1])
    batch = pg.Batch.from_numpy(b, DEV)       # builds the two-view layout and the super-edge lists from b["sizes"]
    b2, lay2 = pg._two_view_batch(batch.batch, batch.num_graphs)
    ref2 = MolLayout(b2)
    assert lay2.P == ref2.P and torch.equal(lay2.pair_i, ref2.pair_i) and torch.equal(lay2.pair_ptr, ref2.pair_ptr)
    se = get_super_edge_layout(batch.batch, batch.super_edge_index, batch.num_graphs)
    from geossl_amd.layout import SuperEdgeLayout
    ref = SuperEdgeLayout(batch.batch, batch.super_edge_index, batch.num_graphs)
    assert torch.equal(se.se_ptr, ref.se_ptr) and torch.equal(se.inc_ptr, ref.inc_ptr) and torch.equal(se.inc_idx, ref.inc_idx)


def test_pair_geometry_matches_edge_list():
    """The pair-slot form used inside SchNet carries exactly the canonical edge set."""
    from geossl_amd import ops
    from geossl_amd.layout import MolLayout
    g = load_golden("g3_radius_graph")
    pos, bat = t(g["positions"], DEV), t(g["batch"], DEV)
    lay = MolLayout(bat)
    for r in (5.0, 10.0):
        d, c, fl = ops.pair_geometry(pos, lay, r)
        pi, pj, fl = lay.pair_i.cpu().long(), lay.pair_j.cpu().long(), fl.cpu()
        src = torch.cat([pj[(fl & 1) > 0], pi[(fl & 2) > 0]])
        dst = torch.cat([pi[(fl & 1) > 0], pj[(fl & 2) > 0]])
        key = torch.argsort(dst * 100000 + src)
        e = torch.stack([src[key], dst[key]])
        assert torch.equal(e, t(g["edge_index_%g" % r]))
        dd = (pos[lay.pair_i.long()] - pos[lay.pair_j.long()]).norm(dim=-1)
        assert max_rel(d, dd) < 1e-6
        cc = 0.5 * (torch.cos(dd * np.pi / r) + 1.0)
        assert float((c - cc).abs().max()) < 2e-6


# ------------------------------------------------------------------------------------ element kernels
def test_gaussian_smearing_golden():
    from geossl_amd import ops
    g = load_golden("g1_g2_smearing_ssp")
    for r, G in [(10.0, 51), (5.0, 51), (10.0, 50)]:
        key = "%g_%d" % (r, G)
        y = ops.gaussian_smearing(t(g["d_" + key], DEV), t(g["offset_" + key], DEV), float(g["coeff_" + key]))
        assert float((y.cpu() - t(g["y_" + key])).abs().max()) < 2e-6


def test_linear_and_ssp_epilogues_vs_torch():
    from geossl_amd import _lib, ops
    torch.manual_seed(0)
    for R, K, NO in [(1000, 128, 128), (77, 32, 32), (300, 64, 64), (5, 128, 128)]:
        x = torch.randn(R, K, device=DEV)
        w = torch.randn(NO, K, device=DEV) / K ** 0.5
        b = torch.randn(NO, device=DEV)
        res = torch.randn(R, NO, device=DEV)
        ref = x.double() @ w.double().t() + b.double()
        assert rel_err(ops.linear(x, w, bias=b).cpu(), ref.cpu()) < 1e-6
        sp = torch.nn.functional.softplus(ref) - float(torch.log(torch.tensor(2.0)))
        assert rel_err(ops.linear(x, w, bias=b, flags=_lib.EPI_SSP).cpu(), sp.cpu()) < 1e-6
        assert rel_err(ops.linear(x, w, bias=b, res=res).cpu(), (ref + res.double()).cpu()) < 1e-6
        # backward-input form: dX = dY @ W, optionally times ssp'(pre) recovered from the ssp output
        dy = torch.randn(R, NO, device=DEV)
        dx_ref = dy.double() @ w.double()
        assert rel_err(ops.linear(dy, w, transB=False).cpu(), dx_ref.cpu()) < 1e-6
        pre = torch.randn(R, K, device=DEV) * 3
        tout = (torch.nn.functional.softplus(pre) - float(torch.log(torch.tensor(2.0))))
        got = ops.linear(dy, w, transB=False, tprev=tout)
        assert rel_err(got.cpu(), (dx_ref * torch.sigmoid(pre.double())).cpu()) < 2e-6
        # weight gradient
        dW = torch.empty(NO, K, device=DEV)
        db = torch.empty(NO, device=DEV)
        ops.linear_wgrad([(dy, x, dW, db)], R, NO, K)
        assert rel_err(dW.cpu(), (dy.double().t() @ x.double()).cpu()) < 1e-6
        assert rel_err(db.cpu(), dy.double().sum(0).cpu()) < 1e-6


def test_prepared_linear_is_bit_identical():
    """geossl_linear_prepare + geossl_linear_prepared = geossl_linear (same arithmetic on a precomputed image)."""
    from geossl_amd import ops, _lib
    gen = torch.Generator().manual_seed(3)
    for K, NO, R in ((128, 128, 1000), (128, 64, 77), (64, 128, 333), (32, 32, 40)):
        x = torch.randn(R, K, generator=gen).to(DEV)
        ws = [(torch.randn(NO, K, generator=gen) / K ** 0.5).to(DEV) for _ in range(3)]
        b = torch.randn(NO, generator=gen).to(DEV)
        res = torch.randn(R, NO, generator=gen).to(DEV)
        for transB in (True, False):
            wl = ws if transB else [w.t().contiguous() for w in ws]
            pw = ops.prepare_linear(wl, transB=transB)
            assert pw is not None
            for w, p in zip(wl, pw):
                a = ops.linear(x, w, bias=b, res=res, transB=transB, flags=_lib.EPI_SSP)
                c = ops.linear(x, p, bias=b, res=res, flags=_lib.EPI_SSP)
                assert torch.equal(a, c)


def test_mfma_tile_layout_asymmetric():
    """A = I against an asymmetric B catches a transposed C-write (cdna guide: always test this)."""
    from geossl_amd import ops
    x = torch.eye(128, device=DEV)
    w = torch.arange(128 * 128, dtype=torch.float32, device=DEV).view(128, 128) * 1e-3
    assert torch.equal(ops.linear(x, w).cpu(), w.t().cpu())
    assert torch.equal(ops.linear(x, w, transB=False).cpu(), w.cpu())


@pytest.mark.parametrize("F", [128, 64, 32])
def test_aggregate_bit_exact_vs_sequential_index_add(F):
    """K4 against the reference's own arithmetic on the canonical edge list: msg = x_j * W rounded to fp32, then a
    sequential index_add per target in ascending source order (schnet.py:190,194-195 on CPU) - bit for bit, for the
    graph and its transpose, ragged molecules incl. 1, 2 and 40 atoms."""
    from geossl_amd import ops
    from geossl_amd.layout import MolLayout
    from geossl_amd.synthetic import make_batch
    sizes = list(make_batch(24, seed=3, mode="B")["sizes"]) + [1, 2, 40, 3]
    batch = torch.arange(len(sizes), device=DEV).repeat_interleave(torch.tensor(sizes, device=DEV))
    lay = MolLayout(batch, len(sizes), sizes=sizes)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(lay.N, F, device=DEV, generator=g)
    W = torch.randn(lay.P, F, device=DEV, generator=g)
    flag = torch.randint(0, 4, (lay.P,), device=DEV, generator=g, dtype=torch.uint8)
    xn, Wn, fn = x.cpu().numpy(), W.cpu().numpy(), flag.cpu().numpy()
    pi, pj = lay.pair_i.cpu().numpy(), lay.pair_j.cpu().numpy()
    for swap in (False, True):
        out = ops.aggregate(x, W, flag, lay, swap=swap).cpu().numpy()
        contrib = {}
        for p in range(lay.P):
            i, j, fl = int(pi[p]), int(pj[p]), int(fn[p])
            if swap:
                fl = ((fl & 1) << 1) | ((fl >> 1) & 1)
            if fl & 1:
                contrib.setdefault(i, []).append((j, p))
            if fl & 2:
                contrib.setdefault(j, []).append((i, p))
        ref = np.zeros_like(xn)
        for tgt, lst in contrib.items():
            acc = np.zeros(F, dtype=np.float32)
            for src, p in sorted(lst):
                acc = (acc + (xn[src] * Wn[p]).astype(np.float32)).astype(np.float32)
            ref[tgt] = acc
        assert np.array_equal(out, ref), swap


@pytest.mark.parametrize("F,G,P", [(128, 51, 1000), (128, 50, 77), (64, 20, 333), (32, 9, 64), (128, 64, 4099)])
def test_filter_network_forward_vs_fp64(F, G, P):
    """geossl_cfconv_filter_fwd through the C ABI against an fp64 evaluation of schnet.py:141-145,186-187,205-207.
    The products run as 3-way bf16 splits on the matrix pipe (csrc/split.h): the bound checked here, 2e-6 of the
    tensor scale, is the accuracy of an fp32 GEMM chain, i.e. the splitting costs no precision."""
    import ctypes as C
    from geossl_amd import _lib
    from geossl_amd._lib import call, ptr, stream
    gen = torch.Generator().manual_seed(F + G + P)
    L, cutoff = 3, 5.0
    d = (torch.rand(P, generator=gen) * 1.1 * cutoff).to(DEV)
    c = (0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)).contiguous()
    offset = torch.linspace(0.0, cutoff, G).to(DEV)
    coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    ws = []
    for l in range(L):
        ws.append([(torch.randn(F, G, generator=gen) / G ** 0.5).to(DEV), (0.3 * torch.randn(F, generator=gen)).to(DEV),
                   (torch.randn(F, F, generator=gen) / F ** 0.5).to(DEV), (0.3 * torch.randn(F, generator=gen)).to(DEV)])
    fw = _lib.FilterWeights()
    for l, w in enumerate(ws):
        fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = (ptr(x) for x in w)
    Wf = torch.full((L, P, F), float("nan"), device=DEV)
    T = torch.full((L, P, F), float("nan"), device=DEV)
    call("geossl_cfconv_filter_fwd", ptr(d), ptr(c), P, C.byref(fw), L, F, G, ptr(offset), coeff, ptr(T), ptr(Wf), stream())
    d64 = d.double()
    rbf = torch.exp(coeff * (d64[:, None] - offset.double()[None, :]) ** 2)
    for l, (w1, b1, w2, b2) in enumerate(ws):
        t64 = torch.nn.functional.softplus(rbf @ w1.double().t() + b1.double()) - math.log(2.0)
        wf64 = (t64 @ w2.double().t() + b2.double()) * c.double()[:, None]
        for got, ref, name in ((T[l], t64, "t"), (Wf[l], wf64, "Wf")):
            err = float((got.double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-6, (name, l, err)


# ------------------------------------------------------------------------------------------- SchNet (K2-K4)
def _schnet_golden_case(tag):
    g = load_golden("g4_schnet_" + tag)
    cfg = cfg_of(g)
    model = product_schnet(cfg, DEV)
    x = t(g["x"], DEV)
    out, h = model(x[:, 0], t(g["positions"], DEV), t(g["batch"], DEV), return_latent=True)
    return g, cfg, model, out, h


@pytest.mark.parametrize("tag", ["reduced", "full_r5", "full_r10"])
def test_schnet_forward_golden(tag):
    g, cfg, model, out, h = _schnet_golden_case(tag)
    assert_close(out.cpu(), g["out"], TOL_OUT, "out")
    assert_close(h.cpu(), g["h"], TOL_OUT, "h")


@pytest.mark.parametrize("tag", ["reduced", "full_r5", "full_r10"])
def test_schnet_param_grads_golden(tag):
    g, cfg, model, out, h = _schnet_golden_case(tag)
    loss = (out ** 2).sum() + (h ** 2).sum() * 0.5
    assert rel_err(loss.detach().cpu(), g["loss"]) < TOL_OUT
    loss.backward()
    grads = unique_named_grads(model)
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(grads[k[5:]].cpu()), g[k]) < TOL_GRAD, k
        if k.startswith("grad/"):
            assert rel_err(grads[k[5:]].cpu(), g[k]) < TOL_GRAD, k


def _energy_and_force(model, x, pos, batch):
    pos = pos.clone().requires_grad_(True)
    out = model(x[:, 0], pos, batch)
    w = torch.cos(torch.arange(out.size(1), dtype=torch.float32, device=out.device))
    energy = (out * w).sum(dim=1)
    # exactly the call of finetune_md17.py:46 / :99 (create_graph=True, then detached by the evaluation loop)
    force = -torch.autograd.grad(outputs=energy, inputs=pos, grad_outputs=torch.ones_like(energy), create_graph=True,
                                 retain_graph=True)[0]
    return energy, force, pos


@pytest.mark.parametrize("tag", ["reduced", "full_r5"])
def test_schnet_forces_golden(tag):
    """SURVEY 8(f) N3, first order: -dE/dpos through the HIP path against the forces of the unmodified reference."""
    g = load_golden("g9_schnet_forces_" + tag)
    model = product_schnet(cfg_of(g), DEV)
    energy, force, _ = _energy_and_force(model, t(g["x"], DEV), t(g["positions"], DEV), t(g["batch"], DEV))
    assert_close(energy.detach().cpu(), g["energy"], TOL_OUT, "energy")
    assert rel_err(force.detach().cpu(), g["force"]) < TOL_GRAD


def test_schnet_forces_vs_fp64_oracle_and_param_grads_unchanged():
    """Ragged synthetic batch (tiny, mid-size and > 32-atom molecules, F = 128, L = 6): forces against the oracle
    evaluated in fp64; asking for the position gradient leaves the parameter gradients bit-identical; the force can
    be differentiated again (training on it, finetune_md17.py:51-54)."""
    from geossl_amd.synthetic import make_batch
    from oracle import nets
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="add")
    b = make_batch(0, seed=21, sizes=[1, 2, 3, 18, 18, 40, 9, 33, 5])
    model = product_schnet(cfg, DEV)
    x, pos, bat = t(b["x"], DEV), t(b["positions"], DEV), t(b["batch"], DEV)
    energy, force, pos_g = _energy_and_force(model, x, pos, bat)
    P64 = {k: v.double() for k, v in schnet_oracle_params(cfg).items()}
    p64 = t(b["positions"]).double().requires_grad_(True)
    out64 = nets.schnet_forward(P64, t(b["x"])[:, 0], p64, t(b["batch"]), 5.0, 6, "add")
    e64 = (out64 * torch.cos(torch.arange(128, dtype=torch.float64))).sum(dim=1)
    f64 = -torch.autograd.grad(e64.sum(), p64)[0]
    assert rel_err(energy.detach().cpu().double(), e64.detach()) < TOL_OUT
    assert rel_err(force.detach().cpu().double(), f64) < TOL_GRAD
    # per-molecule net force vanishes (translation invariance)
    net = torch.zeros(len(b["sizes"]), 3, dtype=torch.float64).index_add_(0, t(b["batch"]), force.detach().cpu().double())
    assert float(net.abs().max()) < 1e-4 * float(f64.abs().max())
    # parameter gradients with and without the position gradient in the same backward
    model.zero_grad()
    energy.sum().backward(inputs=[pos_g] + list(model.parameters()))
    with_pos = {k: v.clone() for k, v in unique_named_grads(model).items()}
    model.zero_grad()
    out = model(x[:, 0], pos, bat)
    ((out * torch.cos(torch.arange(128, dtype=torch.float32, device=DEV))).sum()).backward()
    for k, v in unique_named_grads(model).items():
        assert torch.equal(v, with_pos[k]), k
    # the force is differentiable (training on it, finetune_md17.py:51-54): see test_gpu_round2.py for the parity test
    model.zero_grad()
    (force ** 2).sum().backward()
    assert all(torch.isfinite(v).all() for v in unique_named_grads(model).values())


def test_schnet_no_grad_and_state_dict_roundtrip(tmp_path):
    g = load_golden("g4_schnet_reduced")
    cfg = cfg_of(g)
    model = product_schnet(cfg, DEV)
    x = t(g["x"], DEV)
    with torch.no_grad():
        out = model(x[:, 0], t(g["positions"], DEV), t(g["batch"], DEV))
    assert_close(out.cpu(), g["out"], TOL_OUT, "out")
    # the reference's checkpoint format: {"model": state_dict} (pretrain_GeoSSL.py:54-64)
    path = tmp_path / "model.pth"
    torch.save({"model": model.state_dict()}, path)
    from geossl_amd.Geom3D.models import SchNet
    m2 = SchNet(**cfg).to(DEV)
    m2.load_state_dict(torch.load(path)["model"])
    with torch.no_grad():
        out2 = m2(x[:, 0], t(g["positions"], DEV), t(g["batch"], DEV))
    assert torch.equal(out, out2)


def test_schnet_se3_invariance_full_batch():
    """Domain property at BASELINE size: rotating + translating every molecule leaves h unchanged."""
    from geossl_amd.synthetic import make_batch
    b = make_batch(1024, seed=3, mode="B")
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    model = product_schnet(cfg, DEV)
    x, pos, bat = t(b["x"], DEV), t(b["positions"], DEV), t(b["batch"], DEV)
    q, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))
    pos2 = (pos.double().cpu() @ q + torch.tensor([0.3, -1.2, 2.0], dtype=torch.float64)).float().to(DEV)
    with torch.no_grad():
        out1, h1 = model(x[:, 0], pos, bat, return_latent=True)
        out2, h2 = model(x[:, 0], pos2, bat, return_latent=True)
    assert torch.isfinite(h1).all()
    assert rel_err(h2.cpu(), h1.cpu()) < 5e-5  # distances move by ~1 ulp under an fp32 rotation
    assert out1.shape == (1024, 128)


def test_schnet_vs_oracle_synthetic_128():
    from geossl_amd.synthetic import make_batch
    from oracle import nets
    b = make_batch(128, seed=11, mode="B")
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    model = product_schnet(cfg, DEV)
    P = schnet_oracle_params(cfg)
    x, pos, bat = t(b["x"]), t(b["positions"]), t(b["batch"])
    out_o, h_o = nets.schnet_forward(P, x[:, 0], pos, bat, 5.0, 6, "mean", return_latent=True)
    (h_o ** 2).sum().backward()
    out, h = model(x.to(DEV)[:, 0], pos.to(DEV), bat.to(DEV), return_latent=True)
    (h ** 2).sum().backward()
    assert_close(h.detach().cpu(), h_o.detach(), TOL_OUT, "h")
    assert_close(out.detach().cpu(), out_o.detach(), TOL_OUT, "out")
    grads = unique_named_grads(model)
    for k, gval in grads.items():
        assert rel_err(gval.cpu(), P[k].grad) < TOL_GRAD, k


@pytest.mark.parametrize("cutoff", [5.0, 10.0])
def test_schnet_big_and_tiny_molecules_vs_oracle(cutoff):
    """Paths the Molecule3D-sized fixtures never take: molecules above the 32-neighbour cap (asymmetric graph, flag
    bits differ), pair tiles whose atom window exceeds the LDS stage (60-atom molecules; runs of 2-atom molecules
    packing 128 molecules into one tile -> global-operand fallback of the backward kernels), single atoms."""
    from geossl_amd.synthetic import make_batch
    from oracle import nets
    sizes = [60, 3] + [2] * 150 + [50, 1, 1, 18] + [2] * 40 + [33]
    b = make_batch(0, seed=5, sizes=sizes)
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=3, num_gaussians=51, cutoff=cutoff, node_class=9,
               readout="add")
    model = product_schnet(cfg, DEV)
    P = schnet_oracle_params(cfg)
    x, pos, bat = t(b["x"]), t(b["positions"]), t(b["batch"])
    out_o, h_o = nets.schnet_forward(P, x[:, 0], pos, bat, cutoff, 3, "add", return_latent=True)
    ((h_o ** 2).sum() + out_o.sum()).backward()
    out, h = model(x.to(DEV)[:, 0], pos.to(DEV), bat.to(DEV), return_latent=True)
    ((h ** 2).sum() + out.sum()).backward()
    assert_close(h.detach().cpu(), h_o.detach(), TOL_OUT, "h")
    assert_close(out.detach().cpu(), out_o.detach(), TOL_OUT, "out")
    for k, gval in unique_named_grads(model).items():
        assert rel_err(gval.cpu(), P[k].grad) < TOL_GRAD, k


@pytest.mark.parametrize("F", [32, 64])
def test_schnet_other_widths_vs_oracle(F):
    from geossl_amd.synthetic import make_batch
    from oracle import nets
    b = make_batch(24, seed=8, mode="B")
    cfg = dict(hidden_channels=F, num_filters=F, num_interactions=2, num_gaussians=50, cutoff=10.0, node_class=9,
               readout="mean")
    model = product_schnet(cfg, DEV)
    P = schnet_oracle_params(cfg)
    x, pos, bat = t(b["x"]), t(b["positions"]), t(b["batch"])
    _, h_o = nets.schnet_forward(P, x[:, 0], pos, bat, 10.0, 2, "mean", return_latent=True)
    (h_o ** 2).sum().backward()
    _, h = model(x.to(DEV)[:, 0], pos.to(DEV), bat.to(DEV), return_latent=True)
    (h ** 2).sum().backward()
    assert_close(h.detach().cpu(), h_o.detach(), TOL_OUT, "h")
    for k, gval in unique_named_grads(model).items():
        assert rel_err(gval.cpu(), P[k].grad) < TOL_GRAD, k


def test_schnet_errors():
    from geossl_amd import _lib
    from geossl_amd.Geom3D.models import SchNet
    m = SchNet(32, 32, 2, 8, 5.0, node_class=9).to(DEV)
    with pytest.raises(AssertionError):  # schnet.py:86
        m(torch.zeros(3, 1, dtype=torch.long, device=DEV), torch.zeros(3, 3, device=DEV))
    with pytest.raises(_lib.GeosslHipError):  # no CPU fallback
        SchNet(32, 32, 2, 8, 5.0, node_class=9)(torch.zeros(3, dtype=torch.long), torch.zeros(3, 3))


# --------------------------------------------------------------------------------------------- NCSN (K5)
class _Data:
    def __init__(self, g, dev):
        self.batch = t(g["batch"], dev)
        self.super_edge_index = t(g["super_edge_index"], dev)
        self.x = t(g["x"], dev)
        self.positions = t(g["positions"], dev)

    @property
    def num_graphs(self):
        return self.batch[-1].item() + 1


@pytest.mark.parametrize("tag", ["comb_K50_p2", "comb_K30_p0.05", "comb_K50_p5_last1", "perm_K30_p10"])
def test_ncsn_golden(tag):
    """Loss <= 1e-5 vs the reference's golden value.  Gradients: vs an fp64 evaluation of the oracle <= 1e-4, or, on
    the ill-conditioned case, at least as close as the reference's own fp32 gradient is (with sigma ~ 0.01 and
    anneal_power 0.05 the score - target difference cancels to ~1e-3 of its terms: the reference's fp32 CPU gradient is
    2e-4 off the fp64 value, the HIP path 1.2e-4); and vs the reference's fp32 gradients up to 3x that distance."""
    from oracle import nets
    g = load_golden("g5_ncsn_" + tag)
    K, power = int(g["K"]), float(g["anneal_power"])
    head = product_ncsn(128, K, power, DEV)
    assert torch.equal(head.sigmas.cpu(), t(g["sigmas"]))
    data = _Data(g, DEV)
    h = t(g["h"], DEV).clone().requires_grad_()
    loss = head(data, h, t(g["distance"], DEV), noise_level=t(g["noise_level"], DEV),
                distance_noise=t(g["distance_noise"], DEV))
    assert loss.dim() == 0
    assert rel_err(loss.detach().cpu(), g["loss"]) < TOL_OUT
    loss.backward()
    P64 = {k: v.detach().double().requires_grad_(v.requires_grad) for k, v in ncsn_oracle_params(128, K).items()}
    h64 = t(g["h"]).double().requires_grad_()
    nets.ncsn_v03_forward(P64, t(g["batch"]), t(g["super_edge_index"]), h64, t(g["distance"]).double(),
                          t(g["noise_level"]), t(g["distance_noise"]).double(), power).backward()

    def check(got, gold, truth, what):
        e_truth, e_gold, ref_own = rel_err(got, truth), rel_err(got, gold), rel_err(gold, truth)
        assert e_truth < max(TOL_GRAD, ref_own), "%s vs fp64: %.2e (reference's own fp32 error %.2e)" % (
            what, e_truth, ref_own)
        assert e_gold < max(TOL_GRAD, 3 * ref_own), "%s vs golden: %.2e (reference's own fp32 error %.2e)" % (
            what, e_gold, ref_own)

    check(h.grad.cpu(), g["grad_h"], h64.grad, "grad_h")
    grads = unique_named_grads(head)
    for k in g:
        if k.startswith("grad/"):
            check(grads[k[5:]].cpu(), g[k], P64[k[5:]].grad, k)


def test_ncsn_draws_its_own_noise_like_the_reference():
    g = load_golden("g5_ncsn_comb_K50_p2")
    head = product_ncsn(128, 50, 2, DEV)
    data = _Data(g, DEV)
    torch.manual_seed(3)
    l1 = head(data, t(g["h"], DEV), t(g["distance"], DEV))
    torch.manual_seed(3)
    nl = torch.randint(0, 50, (data.num_graphs,), device=DEV)
    dn = torch.randn_like(t(g["distance"], DEV))
    l2 = head(data, t(g["h"], DEV), t(g["distance"], DEV), noise_level=nl, distance_noise=dn)
    assert torch.equal(l1, l2)


# ---------------------------------------------------------------------------------------------- do_DDM
def _ddm_case(tag, fuse):
    from geossl_amd import pretrain_GeoSSL as pg
    g = load_golden("g6_ddm_" + tag)
    cfg = cfg_of(g)
    F = cfg["hidden_channels"]
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(F, 50, 2, DEV), product_ncsn(F, 50, 2, DEV, scale=0.9)
    batch = pg.Batch(t(g["x"], DEV), t(g["positions"], DEV), t(g["batch"], DEV), t(g["super_edge_index"], DEV))
    noise = {k: t(g[k], DEV) for k in ("pos_noise", "noise_level_1", "dist_noise_1", "noise_level_2", "dist_noise_2")}
    loss, acc = pg.do_DDM(pg.Args("schnet"), batch, model, None, 0.0, 0.3, NCSN_models=(n1, n2), noise=noise,
                          fuse_views=fuse)
    assert acc == 0
    return g, model, n1, n2, loss


@pytest.mark.parametrize("tag", ["reduced", "full"])
@pytest.mark.parametrize("fuse", [True, False])
def test_do_ddm_golden(tag, fuse):
    g, model, n1, n2, loss = _ddm_case(tag, fuse)
    assert rel_err(loss.detach().cpu(), g["loss"]) < TOL_OUT
    loss.backward()
    mods = {"model": unique_named_grads(model), "ncsn1": unique_named_grads(n1), "ncsn2": unique_named_grads(n2)}
    for k in g:
        if k.startswith("gsum/") or k.startswith("grad/"):
            _, m, name = k.split("/", 2)
            got = mods[m][name].cpu()
            got = grad_summary(got) if k.startswith("gsum/") else got
            assert rel_err(got, g[k]) < TOL_GRAD, k


def test_row_normalize_vs_torch():
    """geossl_row_normalize_fwd/bwd against F.normalize (the --normalize branch, pretrain_GeoSSL.py:193-195), incl. an
    all-zero row (clamped norm: y = 0, gradient g / eps like ATen) and a row below eps."""
    from geossl_amd import ops
    gen = torch.Generator().manual_seed(11)
    h = torch.randn(333, 128, generator=gen)
    h[7] = 0.0
    h[9] = 1e-20
    g = torch.randn(333, 128, generator=gen)
    a = h.clone().to(DEV).requires_grad_()
    ya = ops.row_normalize(a)
    ya.backward(g.to(DEV))
    b = h.clone().double().requires_grad_()
    yb = torch.nn.functional.normalize(b, dim=-1)
    yb.backward(g.double())
    assert rel_err(ya.detach().cpu(), yb.detach()) < 1e-6
    keep = torch.ones(333, dtype=torch.bool)
    keep[7] = keep[9] = False  # clamped rows: gradients are g / 1e-12, compared on their own scale
    assert rel_err(a.grad.cpu()[keep], b.grad[keep]) < 1e-6
    assert rel_err(a.grad.cpu()[~keep], b.grad[~keep]) < 1e-5


def test_do_ddm_normalize_vs_oracle():
    """do_DDM with args.normalize (L2-normalised node features before the heads) against the oracle."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    b = make_batch(24, seed=4, mode="B")
    nz = draw_noise(b, seed=5)
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
    batch = pg.Batch.from_numpy(b, DEV)
    noise = {k: t(v, DEV) for k, v in nz.items()}
    loss, _ = pg.do_DDM(pg.Args("schnet", normalize=True), batch, model, None, 0.0, 0.3, NCSN_models=(n1, n2), noise=noise)
    loss.backward()
    Pm, P1, P2 = schnet_oracle_params(cfg), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 5.0, 6, 2, "mean", normalize=True)
    ref.backward()
    assert rel_err(loss.detach().cpu(), ref.detach()) < TOL_OUT
    g = unique_named_grads(model)
    for k in ("lin2.weight", "interactions.0.mlp.0.weight", "interactions.5.conv.lin1.weight", "embedding.weight"):
        assert rel_err(g[k].cpu(), Pm[k].grad) < TOL_GRAD, k


def test_ddm_full_size_vs_oracle_sample_and_determinism():
    """BASELINE config (1024 molecules, n=18, r=5 A): the loss is reproducible bit for bit across two
    runs (no atomics anywhere), and equals the oracle on a 64-molecule slice of the same batch."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)

    def run(nmol):
        b = make_batch(nmol, seed=0, mode="A")
        nz = draw_noise(b, seed=1)
        batch = pg.Batch.from_numpy(b, DEV)
        noise = {k: t(v, DEV) for k, v in nz.items()}
        for m in (model, n1, n2):
            m.zero_grad()
        loss, _ = pg.do_DDM(pg.Args("schnet"), batch, model, None, 0.0, 0.3, NCSN_models=(n1, n2), noise=noise)
        loss.backward()
        return b, nz, loss.detach().clone(), model.lin2.weight.grad.clone()

    _, _, l_a, g_a = run(1024)
    _, _, l_b, g_b = run(1024)
    assert torch.isfinite(l_a) and torch.equal(l_a, l_b) and torch.equal(g_a, g_b)
    b, nz, l_small, _ = run(64)
    Pm, P1, P2 = schnet_oracle_params(cfg, False), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 5.0, 6, 2, "mean")
    assert rel_err(l_small.cpu(), ref.detach()) < TOL_OUT


# ---------------------------------------------------------------------------------------------- PaiNN (config 5)
def _painn_model(cfg):
    from filler import fill_module_
    from geossl_amd.Geom3D.models import PaiNN
    return fill_module_(PaiNN(**cfg)).to(DEV)


def test_painn_forward_and_grads_golden():
    """G7: hydrogen-containing batch (padding_idx row), perturbed geometry so that some precomputed edges lie
    beyond the cutoff (mask path)."""
    g = load_golden("g7_painn")
    cfg = cfg_of(g)
    assert int(g["n_beyond"]) > 0
    model = _painn_model(cfg)
    out, q = model(t(g["x"], DEV), t(g["positions_perturbed"], DEV), t(g["radius_edge_index"], DEV), t(g["batch"], DEV),
                   return_latent=True)
    assert_close(out.cpu(), g["out"], TOL_OUT, "out")
    assert_close(q.cpu(), g["q"], TOL_OUT, "q")
    loss = (out ** 2).sum() + 0.5 * (q ** 2).sum()
    assert rel_err(loss.detach().cpu(), g["loss"]) < TOL_OUT
    loss.backward()
    grads = unique_named_grads(model)
    assert float(grads["embedding.weight"][0].abs().max()) == 0.0  # padding_idx = 0 (painn.py:174)
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(grads[k[5:]].cpu()), g[k]) < TOL_GRAD, k


def test_painn_radius_edge_index_matches_per_molecule_radius_graph():
    """P0 / N4: radius_edge_index = per-molecule radius_graph on the clean geometry (datasets_3D_Radius.py:120)."""
    from geossl_amd import ops
    g = load_golden("g7_painn")
    e = ops.radius_graph(t(g["positions"], DEV), 5.0, t(g["batch"], DEV))
    assert torch.equal(e.cpu(), t(g["radius_edge_index"]))


@pytest.mark.parametrize("fuse", [True, False])
def test_painn_do_ddm_golden(fuse):
    from geossl_amd import pretrain_GeoSSL as pg
    g, d = load_golden("g7_painn"), load_golden("g7_painn_ddm")
    cfg = cfg_of(g)
    model = _painn_model(cfg)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV)
    batch = pg.Batch(t(g["x"], DEV), t(g["positions"], DEV), t(g["batch"], DEV), t(g["super_edge_index"], DEV),
                     radius_edge_index=t(g["radius_edge_index"], DEV))
    noise = {k: t(d[k], DEV) for k in ("pos_noise", "noise_level_1", "dist_noise_1", "noise_level_2", "dist_noise_2")}
    loss, _ = pg.do_DDM(pg.Args("painn"), batch, model, None, 0.0, 0.3, NCSN_models=(n1, n2), noise=noise,
                        fuse_views=fuse)
    assert rel_err(loss.detach().cpu(), d["loss"]) < TOL_OUT
    loss.backward()
    mods = {"model": unique_named_grads(model), "ncsn1": unique_named_grads(n1), "ncsn2": unique_named_grads(n2)}
    for k in d:
        if k.startswith("gsum/"):
            _, m, name = k.split("/", 2)
            assert rel_err(grad_summary(mods[m][name].cpu()), d[k]) < TOL_GRAD, k


# ---------------------------------------------------------------------------------------------- optimizer
def test_fused_adam_matches_torch_adam():
    from geossl_amd.optim import FlatParams, FusedAdam
    torch.manual_seed(0)
    lin_a = torch.nn.Linear(64, 48).to(DEV)
    lin_b = torch.nn.Linear(64, 48).to(DEV)
    lin_b.load_state_dict(lin_a.state_dict())
    ref = torch.optim.Adam(lin_a.parameters(), lr=5e-4)
    flat = FlatParams([lin_b])
    opt = FusedAdam(flat, lr=5e-4)
    for step in range(5):
        x = torch.randn(32, 64, device=DEV)
        for lin, o in ((lin_a, ref), (lin_b, opt)):
            o.zero_grad()
            lin(x).pow(2).sum().backward()
        flat.rebind_grads()
        ref.step()
        opt.step()
    assert rel_err(lin_b.weight.detach().cpu(), lin_a.weight.detach().cpu()) < 1e-6


def test_trainer_graph_replay_matches_eager():
    """HIP-graph replay of forward+backward gives bit-identical losses to eager execution over several steps with
    changing positions and noise (same index structure)."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    losses = {}
    for use_graph in (False, True):
        model = product_schnet(cfg, DEV)
        n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
        tr = pg.DDMTrainer(model, n1, n2, lr=5e-4, use_graph=use_graph)
        out = []
        for step in range(5):
            b = make_batch(64, seed=step, mode="A")
            batch = pg.Batch.from_numpy(b, DEV)
            noise = {k: t(v, DEV) for k, v in draw_noise(b, seed=100 + step).items()}
            out.append(float(tr.step(batch, noise, structure_key=("A", 64, 18))))
        losses[use_graph] = out
    assert losses[True] == losses[False], losses


_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r})
sys.path.insert(0, os.path.join({repo!r}, "tests")); sys.path.insert(0, os.path.join({repo!r}, "tests", "golden"))
import faulthandler
faulthandler.dump_traceback_later(120, exit=True)   # a stuck rank reports where and leaves; nothing lingers on the GPU
import torch, torch.distributed as dist
from geossl_amd import pretrain_GeoSSL as pg
from geossl_amd.parallel import init_distributed, local_device, shard_batch_numpy
from geossl_amd.synthetic import draw_noise, make_batch
from helpers import product_ncsn, product_schnet, t
rank, local_rank, world = init_distributed()          # GEOSSL_DIST_BACKEND=gloo: both ranks on cuda:0
dev = torch.device("cuda", local_device(local_rank))
torch.cuda.set_device(dev)
cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
           readout="mean")
tr = pg.DDMTrainer(product_schnet(cfg, dev), product_ncsn(128, 50, 2, dev), product_ncsn(128, 50, 2, dev, scale=0.9),
                   lr=5e-4, use_graph=True)
losses = []
for step in range(3):
    mine = shard_batch_numpy(make_batch(64, seed=step, mode="A"), rank, world)
    noise = {{k: t(v, dev) for k, v in draw_noise(mine, seed=100 + 10 * step + rank).items()}}
    losses.append(float(tr.step(pg.Batch.from_numpy(mine, dev), noise, structure_key=("A", 32, 18))))
assert tr.use_graph, "capture fell back to eager"
torch.save(dict(losses=losses, params=tr.flat.flat.cpu()), os.path.join({out!r}, "rank%d.pt" % rank))
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_trainer_matches_single_process(tmp_path):
    """The N>1 step on the device path: two ranks (gloo transport, sharing cuda:0 - the box has one GPU) shard each
    batch by whole molecules, replay their captured forward+backward, all-reduce the flat gradient and apply Adam.
    One process doing the same reduction by hand (sum of the two shard gradients, grad_scale 1/2) must land on
    bit-identical parameters: the sum of two floats does not depend on the transport."""
    import os
    import subprocess
    import sys
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.parallel import shard_batch_numpy
    from geossl_amd.synthetic import draw_noise, make_batch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER.format(repo=repo, out=str(tmp_path)))
    import socket
    with socket.socket() as sock:  # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", GEOSSL_DIST_BACKEND="gloo")
    logs = [open(tmp_path / ("rank%d.log" % r), "w") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=logs[r], stderr=subprocess.STDOUT) for r in range(2)]
    try:
        codes = [p.wait(timeout=180) for p in procs]
    except subprocess.TimeoutExpired:
        codes = None
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        for f in logs:
            f.close()
    tails = "\n".join(open(tmp_path / ("rank%d.log" % r)).read()[-1500:] for r in range(2))
    assert codes == [0, 0], tails
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert torch.equal(r0["params"], r1["params"])  # replicas stay in lock step

    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    tr = pg.DDMTrainer(product_schnet(cfg, DEV), product_ncsn(128, 50, 2, DEV),
                       product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4, use_graph=False)
    for step in range(3):
        full = make_batch(64, seed=step, mode="A")
        total, losses = None, []
        for rank in range(2):
            mine = shard_batch_numpy(full, rank, 2)
            noise = {k: t(v, DEV) for k, v in draw_noise(mine, seed=100 + 10 * step + rank).items()}
            losses.append(float(tr._fwd_bwd(pg.Batch.from_numpy(mine, DEV), noise)))
            total = tr.flat.grad.clone() if total is None else total + tr.flat.grad
        tr.flat.grad.copy_(total)
        tr.opt.step(grad_scale=0.5)
        assert losses == [r0["losses"][step], r1["losses"][step]]
    assert torch.equal(tr.flat.flat.cpu(), r0["params"])


def test_trainer_step_reduces_loss():
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    cfg = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
               readout="mean")
    torch.manual_seed(0)
    from geossl_amd.Geom3D.models import SchNet
    from geossl_amd.NCSN import NCSN_version_03
    model = SchNet(**cfg).to(DEV)
    n1 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(DEV)
    n2 = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2).to(DEV)
    tr = pg.DDMTrainer(model, n1, n2, lr=5e-4)
    b = make_batch(256, seed=0)
    batch = pg.Batch.from_numpy(b, DEV)
    noise = {k: t(v, DEV) for k, v in draw_noise(b, seed=1).items()}
    losses = [float(tr.step(batch, noise)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
