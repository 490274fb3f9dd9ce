"""GPU tests of the second-order route on the library's own tape (geossl_amd/tape.py, csrc/tape.hip): training on
forces (examples/finetune_md17.py:46-54) differentiates the position gradient of the backbone again.  The kernels
against fp64 torch, the tape's derivatives (first and second) against torch.autograd on the same expression, the two
backbones' second order on the tape against the round-4 route (a torch autograd graph over the same primitives) and -
in test_gpu_round2.py, unchanged - against fixtures G10 / G13 of the unmodified reference; and that the tape dispatches
no floating-point ATen compute."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from helpers import cfg_of, fill_module_, product_schnet, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_GRAD = 1e-4   # the suite's bound for second-order gradients (fixtures G10 / G13); the routes agree to 1e-5 .. 3e-5


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from geossl_amd import _lib
    _lib.load()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_element_wise_maps_vs_fp64():
    """Every map of geossl_tape_unary against its formula in fp64 (incl. the softplus threshold of F.softplus and each
    family's derivative chain against torch.autograd of the map before it)."""
    from geossl_amd import tape as tp
    x = torch.cat([_rand(4099, seed=1, scale=3.0), torch.tensor([0.0, 19.9, 20.0, 20.1, 25.0, -30.0], device=DEV)])
    xd = x.double()
    sig = torch.sigmoid
    a, b = 0.7, -0.2
    td = a * xd + b
    ref = {
        tp.AFFINE: td, tp.EXP: torch.exp(td), tp.COS: torch.cos(td), tp.SIN: torch.sin(td),
        tp.SSP: torch.nn.functional.softplus(td) - math.log(2.0), tp.SIGMOID: sig(td), tp.DSIGMOID: sig(td) * (1 - sig(td)),
        tp.D2SIGMOID: sig(td) * (1 - sig(td)) * (1 - 2 * sig(td)), tp.SILU: td * sig(td),
        tp.DSILU: sig(td) * (1 + td * (1 - sig(td))), tp.D2SILU: sig(td) * (1 - sig(td)) * (2 + td * (1 - 2 * sig(td))),
    }
    for kind, r in ref.items():
        got = tp._raw_unary(kind, x, a, b)
        assert float((got.double() - r).abs().max() / r.abs().max().clamp_min(1.0)) < 2e-6, kind
    pos = x.abs() + 0.5
    assert rel_err(tp._raw_unary(tp.RECIP, pos, 2.0, 0.1).cpu(), (1.0 / (2.0 * pos.double() + 0.1)).cpu()) < 1e-6
    assert rel_err(tp._raw_unary(tp.SQRT, pos, 2.0, 0.1).cpu(), torch.sqrt(2.0 * pos.double() + 0.1).cpu()) < 1e-6
    tq = 2.0 * pos.double() + 0.1
    for kind, r in ((tp.DRECIP, -1.0 / tq ** 2), (tp.D2RECIP, 2.0 / tq ** 3), (tp.RSQRT, tq ** -0.5), (tp.RSQRT3, tq ** -1.5)):
        assert rel_err(tp._raw_unary(kind, pos, 2.0, 0.1).cpu(), r.cpu()) < 1e-6, kind
    assert rel_err(tp._raw_unary(tp.ABS, x, a, b).cpu(), td.abs().cpu()) < 1e-6
    assert torch.equal(tp._raw_unary(tp.SIGN, x), torch.sign(x))
    c = -0.35
    xs = x.clamp(-6, 6)
    e = torch.exp(c * xs.double() ** 2)
    for kind, r in ((tp.GAUSS, e), (tp.DGAUSS, 2 * c * xs.double() * e), (tp.D2GAUSS, (2 * c + 4 * c * c * xs.double() ** 2) * e)):
        assert float((tp._raw_unary(kind, xs, c).double() - r).abs().max()) < 2e-6, kind
    assert torch.equal(tp._raw_unary(tp.LT, x, 0.25), (x < 0.25).float())
    # an unaligned input (a view one float into a buffer) takes the scalar path
    buf = _rand(1001, seed=2)
    assert torch.equal(tp._raw_unary(tp.AFFINE, buf[1:], 2.0, 1.0), tp._raw_unary(tp.AFFINE, buf[1:].clone(), 2.0, 1.0))


@pytest.mark.parametrize("R,D", [(30, 128), (7, 3), (33, 56), (3, 1)])
def test_broadcast_arithmetic_reductions_gather_scatter_and_copies(R, D):
    from geossl_amd import tape as tp
    full, row, col, third = _rand(R, D, seed=3), _rand(R, 1, seed=4), _rand(1, D, seed=5), _rand(R // 3, D, seed=6)
    view = {tp.FULL: lambda v: v, tp.ROW: lambda v: v.expand(R, D), tp.COL: lambda v: v.expand(R, D),
            tp.THIRD: lambda v: v.repeat_interleave(3, dim=0)}
    operands = {tp.FULL: full, tp.ROW: row, tp.COL: col}
    if R % 3 == 0:
        operands[tp.THIRD] = third
    fns = {tp.ADD: lambda a, b: a + b, tp.SUB: lambda a, b: a - b, tp.MUL: lambda a, b: a * b}
    for am, a in operands.items():
        for bm, b in operands.items():
            for op, fn in fns.items():
                got = tp._raw_binary(op, a, am, b, bm, R, D, 0.5)
                assert torch.equal(got, 0.5 * fn(view[am](a), view[bm](b))), (op, am, bm)
        assert torch.equal(tp._raw_binary(tp.FIRST, a, am, None, tp.FULL, R, D, 2.0), 2.0 * view[am](a).contiguous())
    x64 = full.double()
    assert rel_err(tp._raw_reduce(tp.ROW, full).cpu(), x64.sum(1, keepdim=True).cpu()) < 1e-6
    assert rel_err(tp._raw_reduce(tp.COL, full).cpu(), x64.sum(0, keepdim=True).cpu()) < 1e-6
    if R % 3 == 0:
        assert rel_err(tp._raw_reduce(tp.THIRD, full).cpu(), x64.view(R // 3, 3, D).sum(1).cpu()) < 1e-6
    big = _rand(70001, D, seed=7)   # several blocks of the column sum
    assert rel_err(tp._raw_reduce(tp.COL, big).cpu(), big.double().sum(0, keepdim=True).cpu()) < 1e-5
    assert torch.equal(tp._raw_reduce(tp.COL, big), tp._raw_reduce(tp.COL, big))
    for dtype in (torch.int32, torch.int64):
        idx = torch.randint(0, R, (5 * R + 1,), generator=torch.Generator().manual_seed(8)).to(DEV).to(dtype)
        ix = tp.Index(idx, R)
        g = tp.gather(tp.const(full), ix)
        assert torch.equal(g.t, full[idx.long()])
        s = tp.scatter(g, ix)
        ref = torch.zeros(R, D, dtype=torch.float64, device=DEV).index_add(0, idx.long(), g.t.double())
        assert rel_err(s.t.cpu(), ref.cpu()) < 1e-6
        assert torch.equal(s.t, tp.scatter(g, ix).t)
    if D >= 8:
        sl = tp.slice_cols(tp.const(full), 3, 5)
        assert torch.equal(sl.t, full[:, 3:8])
        pd = tp.pad_cols(sl, 3, D)
        ref = torch.zeros_like(full)
        ref[:, 3:8] = full[:, 3:8]
        assert torch.equal(pd.t, ref)
        assert torch.equal(tp.cat_cols([sl, tp.const(full), sl]).t, torch.cat([sl.t, full, sl.t], dim=1))


def test_mm_forms_at_any_width_without_torch_padding():
    """tape.mm_raw: the three GEMM forms at widths off the kernels' grid (padding / slabbing by the block-copy kernel),
    with and without bias, against fp64."""
    from geossl_amd import tape as tp
    for R, K, NO in ((1000, 51, 128), (333, 20, 1152), (257, 256, 384), (64, 128, 1), (100, 3, 6), (50, 264, 130)):
        a, w, bias = _rand(R, K, seed=9), _rand(NO, K, seed=10, scale=0.2), _rand(NO, seed=11)
        ref = a.double() @ w.double().t()
        assert rel_err(tp.mm_raw(a, w, "nt").cpu(), ref.cpu()) < 2e-6, (R, K, NO)
        assert rel_err(tp.mm_raw(a, w, "nt", bias=bias).cpu(), (ref + bias.double()).cpu()) < 2e-6
        g = _rand(R, NO, seed=12)
        assert rel_err(tp.mm_raw(g, w, "nn").cpu(), (g.double() @ w.double()).cpu()) < 2e-6
        assert rel_err(tp.mm_raw(g, a, "tn").cpu(), (g.double().t() @ a.double()).cpu()) < 2e-6
    assert tp.mm_raw(_rand(0, 8), _rand(4, 8), "nt").shape == (0, 4)
    assert float(tp.mm_raw(_rand(0, 8), _rand(0, 4), "tn").abs().max()) == 0.0


def test_tape_first_and_second_derivatives_vs_torch_autograd():
    """A composite of every primitive family (gather, distance, Gaussian, cosine, Linear + ssp / silu, the xyz
    broadcasts, scatter, slices / concatenation): its gradient and the gradient OF a contraction of that gradient
    against torch.autograd in fp64."""
    from geossl_amd import tape as tp
    N, E, Fd, G = 12, 40, 16, 8
    gen = torch.Generator().manual_seed(13)
    pos = torch.randn(N, 3, generator=gen).to(DEV)
    ii = torch.randint(0, N, (E,), generator=gen).to(DEV)
    ij = (ii + 1 + torch.randint(0, N - 1, (E,), generator=gen).to(DEV)) % N
    w1, b1 = _rand(Fd, G, seed=14, scale=0.5), _rand(Fd, seed=15)
    w2, b2 = _rand(3 * Fd, Fd, seed=16, scale=0.3), _rand(3 * Fd, seed=17)
    off = torch.linspace(0.0, 3.0, G).to(DEV)
    cot_x, cot_w = _rand(N, 3, seed=18), _rand(Fd, G, seed=19)
    dh = _rand(N, 2 * Fd, seed=20)

    def torch_f(x, w1_, b1_, w2_, b2_):
        r = x[ii] - x[ij]
        d = r.norm(dim=1, keepdim=True)
        rbf = torch.exp(-0.8 * (d - off.to(x.dtype).view(1, -1)) ** 2)
        c = 0.5 * (torch.cos(d * math.pi / 6.0) + 1.0)
        h = torch.nn.functional.softplus(rbf @ w1_.t() + b1_) - math.log(2.0)
        y = torch.nn.functional.silu(h) @ w2_.t() + b2_                       # [E, 3F]
        y = y * c
        a, bb, cc = torch.split(y, Fd, dim=1)
        dirv = r / d
        vec = a[:, None, :] * dirv[..., None] + bb[:, None, :] * cc[:, None, :].expand(E, 3, Fd)   # [E, 3, F]
        nrm = torch.sqrt((vec ** 2).sum(1) + 1e-8)                             # [E, F]
        out = torch.cat([nrm, a * (vec * vec).sum(1)], dim=1)                  # [E, 2F]
        return torch.zeros(N, 2 * Fd, dtype=x.dtype, device=x.device).index_add(0, ii, out)

    def tape_f(x, w1_, b1_, w2_, b2_):
        ixi, ixj = tp.Index(ii, N), tp.Index(ij, N)
        r = tp.sub(tp.gather(x, ixi), tp.gather(x, ixj))
        d = tp.unary(tp.SQRT, tp.reduce(tp.ROW, tp.mul(r, r)))
        rbf = tp.unary(tp.GAUSS, tp.binary(tp.SUB, d, tp.ROW, tp.const(off), tp.COL, E, G), -0.8)
        c = tp.unary(tp.AFFINE, tp.unary(tp.COS, d, math.pi / 6.0), 0.5, 0.5)
        h = tp.ssp(tp.linear(rbf, w1_, b1_))
        y = tp.linear(tp.silu(h), w2_, b2_)
        y = tp.binary(tp.MUL, y, tp.FULL, c, tp.ROW, E, 3 * Fd)
        a, bb, cc = (tp.slice_cols(y, k * Fd, Fd) for k in range(3))
        dirv = tp.binary(tp.MUL, r, tp.FULL, tp.unary(tp.RECIP, d), tp.ROW, E, 3)
        vec = tp.add(tp.binary(tp.MUL, a, tp.THIRD, tp.reshape(dirv, 3 * E, 1), tp.ROW, 3 * E, Fd),
                     tp.binary(tp.MUL, bb, tp.THIRD, tp.binary(tp.FIRST, cc, tp.THIRD, None, tp.FULL, 3 * E, Fd), tp.FULL,
                               3 * E, Fd))
        nrm = tp.unary(tp.SQRT, tp.reduce(tp.THIRD, tp.mul(vec, vec)), 1.0, 1e-8)
        out = tp.cat_cols([nrm, tp.mul(a, tp.reduce(tp.THIRD, tp.mul(vec, vec)))])
        return tp.scatter(out, ixi)

    ins64 = [v.double().requires_grad_(True) for v in (pos, w1, b1, w2, b2)]
    dh64 = dh.double().requires_grad_(True)
    y64 = torch_f(*ins64)
    first64 = torch.autograd.grad(y64, [ins64[0], ins64[1]], dh64, create_graph=True)
    s = (first64[0] * cot_x.double()).sum() + (first64[1] * cot_w.double()).sum()
    second64 = torch.autograd.grad(s, [dh64] + ins64)

    with torch.no_grad():
        leaves = [tp.leaf(v) for v in (pos, w1, b1, w2, b2)]
        dhv = tp.leaf(dh)
        y = tape_f(*leaves)
        assert rel_err(y.t.cpu(), y64.detach().cpu()) < 1e-5
        first = tp.grad([y], [dhv], [leaves[0], leaves[1]])
        for f, r in zip(first, first64):
            assert rel_err(f.t.cpu(), r.detach().cpu()) < 2e-5
        second = tp.grad(list(first), [tp.const(cot_x), tp.const(cot_w)], [dhv] + leaves)
    for name, g, r in zip(("dh", "pos", "w1", "b1", "w2", "b2"), second, second64):
        assert rel_err(g.t.reshape(r.shape).cpu(), r.cpu()) < 5e-5, name


def _force_training_grads(model, head, g, painn):
    pos = t(g["positions"], DEV).clone().requires_grad_(True)
    if painn:
        rep = model(t(g["x"], DEV), pos, t(g["radius_edge_index"], DEV), t(g["batch"], DEV))
    else:
        rep = model(t(g["x"], DEV)[:, 0], pos, t(g["batch"], DEV))
    energy = head(rep).squeeze(1)
    force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    crit = torch.nn.MSELoss()
    loss = crit(energy, t(g["actual_energy"], DEV)) + 10.0 * crit(force, t(g["actual_force"], DEV))
    for p in list(model.parameters()) + list(head.parameters()):
        p.grad = None
    loss.backward()
    out = {"pos": pos.grad.clone()}
    for n, p in list(model.named_parameters()) + [("head." + n, p) for n, p in head.named_parameters()]:
        if p.grad is not None:
            out[n] = p.grad.clone()
    return out


@pytest.mark.parametrize("backbone", ["schnet", "painn"])
def test_second_order_on_the_tape_matches_the_torch_graph_route(backbone, monkeypatch):
    """Training on forces (finetune_md17.py:46-54), every gradient of loss.backward(): the tape (default) against the
    round-4 route (GEOSSL_SECOND_ORDER=torch: the same primitives as a torch autograd graph).  Both are fp32 on the
    same GEMM kernels; they differ in the rounding of the element-wise glue only."""
    if backbone == "schnet":
        g = load_golden("g10_schnet_force_training_full_r5")
        model = product_schnet(cfg_of(g), DEV)
        head = fill_module_(torch.nn.Linear(cfg_of(g)["hidden_channels"], 1)).to(DEV)
    else:
        from test_gpu_round2 import _painn
        g = load_golden("g13_painn_force_training")
        model = _painn(cfg_of(g))
        head = fill_module_(model.create_output_layers()).to(DEV)
    res = {}
    for route in ("tape", "torch"):
        monkeypatch.setenv("GEOSSL_SECOND_ORDER", route)
        res[route] = _force_training_grads(model, head, g, backbone == "painn")
    assert set(res["tape"]) == set(res["torch"])
    for k, v in res["torch"].items():
        assert rel_err(res["tape"][k].cpu(), v.cpu()) < TOL_GRAD, k


@pytest.mark.parametrize("backbone", ["schnet", "painn"])
def test_the_tape_dispatches_no_floating_point_aten_compute(backbone):
    """the second-order route on kernels.  Every ATen operator dispatched while tape.second_order runs
    (a TorchDispatchMode sees them all: the tape runs in the calling thread, under no_grad) either allocates / views /
    reshapes, or works on integers (the incidence lists of a gather's adjoint) - none computes on floating-point data."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from geossl_amd import tape as tp
    from geossl_amd.synthetic import make_batch
    b = make_batch(6, seed=5, mode="B")
    pos = t(b["positions"], DEV)
    N = pos.size(0)
    seen = []

    class Recorder(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            outs = out if isinstance(out, (tuple, list)) else (out,)
            floats = any(isinstance(o, torch.Tensor) and o.is_floating_point() for o in outs)
            seen.append((str(func), floats))
            return out

    if backbone == "schnet":
        from geossl_amd.layout import get_layout
        model = product_schnet(dict(hidden_channels=128, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=5.0,
                                    node_class=9, readout="mean"), DEV)
        z, bat = t(b["x"], DEV)[:, 0], t(b["batch"], DEV)
        pos_r = pos.clone().requires_grad_(True)
        out = model(z, pos_r, bat)
        fctx = out.grad_fn
        while fctx is not None and not hasattr(fctx, "cfg"):   # the fused node of the backbone
            fctx = fctx.next_functions[0][0] if fctx.next_functions else None
        assert fctx is not None
        params = list(fctx.params)
        feats = lambda x, ps: tp.schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps)
        dh = _rand(N, 128, seed=21)
    else:
        from test_gpu_round2 import _painn, _painn_batch, PAINN
        model = _painn(PAINN)
        bt = _painn_batch(b)
        pos_r = pos.clone().requires_grad_(True)
        out = model(bt.x, pos_r, bt.radius_edge_index, bt.batch)
        fctx = out.grad_fn
        while fctx is not None and not hasattr(fctx, "cfg"):
            fctx = fctx.next_functions[0][0] if fctx.next_functions else None
        assert fctx is not None
        params = list(fctx.params)
        el = fctx.el
        feats = lambda x, ps: tp.painn_atom_features(fctx.z, x, el.idx_i, el.idx_j, fctx.cfg, ps)
        dh = _rand(N, out.size(1), seed=21)
    mask = [True] + [False] * len(params)
    need = [True, True] + [True] * len(params)
    cot = [_rand(N, 3, seed=22)]
    with Recorder():
        res = tp.second_order(feats, mask, need, cot, dh, pos, [p.detach() for p in params])
    assert all(r is not None for r in res[:2])
    harmless = ("aten.empty", "aten.view", "aten.reshape", "aten._unsafe_view", "aten.detach", "aten.slice", "aten.select",
                "aten.as_strided", "aten.alias", "aten.expand", "aten.t.", "aten.transpose", "aten.unsqueeze",
                "aten.squeeze", "aten._reshape_alias", "aten._local_scalar_dense", "aten.item", "aten.empty_like", "aten.lift_fresh")
    bad = sorted({name for name, floats in seen if floats and not name.startswith(harmless)})
    assert not bad, bad
    assert len(seen) > 100   # the recorder did see the tape's allocations


@pytest.mark.parametrize("backbone", ["schnet", "painn"])
@pytest.mark.parametrize("kind", ["l1", "mse"])
def test_energy_force_loss_with_gradients_added_in_place_matches_the_written_out_step(backbone, kind):
    """finetune_md17.py:46-54 on the library's pieces (ops.energy_force_loss as one node, a Dense energy head, parameter
    gradients of BOTH routes - from the energy and through the force - added straight into a flat gradient buffer under
    _lib.direct_grads) against the same lines written out with torch's loss arithmetic and plain loss.backward()."""
    from geossl_amd import _lib, ops
    from geossl_amd.Geom3D.models.painn import Dense
    from geossl_amd.optim import FlatParams
    if backbone == "schnet":
        g = load_golden("g10_schnet_force_training_full_r5")
        model = product_schnet(cfg_of(g), DEV)
        head = fill_module_(Dense(cfg_of(g)["hidden_channels"], 1)).to(DEV)
        rep_of = lambda pos: model(t(g["x"], DEV)[:, 0], pos, t(g["batch"], DEV))
    else:
        from test_gpu_round2 import _painn
        g = load_golden("g13_painn_force_training")
        model = _painn(cfg_of(g))
        head = fill_module_(model.create_output_layers()).to(DEV)
        rep_of = lambda pos: model(t(g["x"], DEV), pos, t(g["radius_edge_index"], DEV), t(g["batch"], DEV))
    y_e, y_f = t(g["actual_energy"], DEV), t(g["actual_force"], DEV)
    params = [p for p in list(model.parameters()) + list(head.parameters()) if p.requires_grad]
    crit = torch.nn.L1Loss() if kind == "l1" else torch.nn.MSELoss()
    pos = t(g["positions"], DEV).clone().requires_grad_(True)
    energy = head(rep_of(pos)).squeeze(1)
    force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    loss = 0.05 * crit(energy, y_e) + 0.95 * crit(force, y_f)
    for p in params:
        p.grad = None
    loss.backward()
    want = [p.grad.clone() for p in params]

    flat = FlatParams([model, head])
    assert [id(p) for p in flat.trainable] == [id(p) for p in params]
    pos = t(g["positions"], DEV).clone().requires_grad_(True)
    energy = head(rep_of(pos)).squeeze(1)
    dE = torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    loss2 = ops.energy_force_loss(energy, y_e, dE, y_f, 0.05, 0.95, kind)
    assert abs(float(loss2.detach()) - float(loss.detach())) < 1e-6 * abs(float(loss.detach()))
    flat.zero_grad()
    with _lib.direct_grads():
        loss2.backward(inputs=flat.trainable)
    assert pos.grad is None
    for p, w in zip(params, want):
        assert p.grad.data_ptr() >= flat.grad.data_ptr() and p.grad.data_ptr() < flat.grad.data_ptr() + 4 * flat.numel
        assert rel_err(p.grad.cpu(), w.cpu()) < TOL_GRAD


@pytest.mark.parametrize("backbone", ["schnet", "painn"])
def test_training_on_forces_of_a_batch_without_pairs_or_edges(backbone):
    """Single-atom molecules: no pair slots (SchNet), no edges (PaiNN).  The force is zero, and loss.backward() through it
    runs the tape on empty edge tensors (no GEMM is launched on zero rows): every gradient finite, the backbone's filter
    parameters - reached through edges only - untouched, and the energy route's gradients equal to those of a step
    without the force term."""
    from geossl_amd.Geom3D.models.painn import Dense
    if backbone == "schnet":
        from geossl_amd.synthetic import make_batch
        b = make_batch(0, seed=3, sizes=np.array([1, 1, 1, 1], dtype=np.int64))
        model = product_schnet(dict(hidden_channels=128, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=5.0,
                                    node_class=9, readout="add"), DEV)
        z, bat = t(b["x"], DEV)[:, 0], t(b["batch"], DEV)
        rep_of = lambda pos: model(z, pos, bat)
        positions = t(b["positions"], DEV)
    else:
        from test_gpu_round4 import _painn_model_and_batch
        model, bt = _painn_model_and_batch([1, 1, 1, 1], seed=3)
        assert bt.radius_edge_index.size(1) == 0
        rep_of = lambda pos: model(bt.x, pos, bt.radius_edge_index, bt.batch)
        positions = bt.positions
    head = fill_module_(Dense(128, 1)).to(DEV)
    params = [p for p in list(model.parameters()) + list(head.parameters()) if p.requires_grad]
    y_e = torch.linspace(-1.0, 1.0, 4, device=DEV)
    y_f = torch.full_like(positions, 0.25)
    grads = {}
    for with_force in (True, False):
        pos = positions.clone().requires_grad_(True)
        energy = head(rep_of(pos)).squeeze(1)
        loss = ((energy - y_e) ** 2).mean()
        if with_force:
            force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
            assert float(force.detach().abs().max()) == 0.0
            loss = loss + 10.0 * ((force - y_f) ** 2).mean()
        for p in params:
            p.grad = None
        loss.backward()
        grads[with_force] = [None if p.grad is None else p.grad.clone() for p in params]
    for a, b_ in zip(grads[True], grads[False]):
        if a is None or b_ is None:
            assert (a is None or float(a.abs().max()) == 0.0) and (b_ is None or float(b_.abs().max()) == 0.0)
            continue
        assert torch.isfinite(a).all()
        assert rel_err(a.cpu(), b_.cpu()) < 1e-6 or float((a - b_).abs().max()) < 1e-7


def test_painn_second_order_with_padding_atoms_matches_the_torch_graph_route(monkeypatch):
    """Atom type 0 is PaiNN's padding row (painn.py:174 padding_idx=0): its embedding row is read in the forward and gets
    no gradient - also not through the force.  Tape against the torch-graph route on a batch with such atoms, a single
    atom and a two-atom molecule."""
    from test_gpu_round4 import _painn_model_and_batch
    model, bt = _painn_model_and_batch([7, 12, 1, 2, 9], seed=6)
    bt.x[::4, 0] = 0
    head = fill_module_(model.create_output_layers()).to(DEV)
    B = 5
    g = {"positions": bt.positions.cpu().numpy(), "x": bt.x.cpu().numpy(), "batch": bt.batch.cpu().numpy(),
         "radius_edge_index": bt.radius_edge_index.cpu().numpy(),
         "actual_energy": np.linspace(-1.0, 1.0, B).astype(np.float32),
         "actual_force": (0.1 * np.random.default_rng(2).normal(size=tuple(bt.positions.shape))).astype(np.float32)}
    res = {}
    for route in ("tape", "torch"):
        monkeypatch.setenv("GEOSSL_SECOND_ORDER", route)
        res[route] = _force_training_grads(model, head, g, True)
    assert float(res["tape"]["embedding.weight"][0].abs().max()) == 0.0
    assert float(res["tape"]["embedding.weight"].abs().max()) > 0.0
    for k, v in res["torch"].items():
        assert rel_err(res["tape"][k].cpu(), v.cpu()) < TOL_GRAD, k
