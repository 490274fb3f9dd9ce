"""Differentiable-to-any-order restatement of the SchNet message-passing stack, used ONLY when the backward pass itself
has to be differentiated: training on forces (finetune_md17.py:46-54) takes ``pred_force = -grad(E, pos,
create_graph=True)`` and back-propagates a loss on the force, i.e. differentiates the position gradient with respect to
the parameters.  The fused first-order kernels (one autograd node for the whole stack) cannot be differentiated again,
so ``_SchNetCore.backward`` - when autograd runs it with grad mode on (= ``create_graph=True``) - rebuilds the forward
from the primitives below, each of which has a backward written in terms of the same primitives, and returns
``torch.autograd.grad(..., create_graph=True)`` of that graph.

What runs where: every GEMM (``_MM``: the three forms X W^T, X W, A^T B are closed under differentiation) on the split-bf16
row / column GEMM kernels, the neighbour aggregation and its filter gradient (``_Agg`` / ``_PairProd``, closed as well)
on geossl_cfconv_aggregate / geossl_pair_product, the radius graph on geossl_pair_geometry; the element-wise glue of
this path (distance, Gaussian smearing, cosine envelope, softplus, bias adds) is plain torch on the same device -
PyTorch's own derivative formulas carry the higher orders there.  PaiNN works the same way (``painn_atom_features``,
``PaiNNGradNode``): its forward and its first-order gradients - positions included, painn_force.hip - are the fused
kernels, and only a backward through those gradients walks the primitive graph.  Nothing here is on the DDM hot path.
"""
import math
import os

import torch
from .switches import env as _env
import torch.nn.functional as F

from . import ops

SSP_SHIFT = torch.log(torch.tensor(2.0)).item()  # schnet.py:213


from .tape import mm_raw as _mm_raw  # the three GEMM forms on the HIP kernels, padding / slabbing by the block-copy kernel


def _to_leaf(t, g):
    """Inside _lib.direct_grads() a gradient for a leaf that owns a dense fp32 .grad (a parameter of an energy head built
    from Dense layers, reached from the energy AND through the force) is added straight into that .grad and the node
    returns none for it - no AccumulateGrad add per contribution.  Outside the context: g, through autograd."""
    from . import _lib
    if g is None or torch.is_grad_enabled() or _lib._DIRECT["depth"] <= 0:
        return g
    if not (t.is_leaf and t.requires_grad) or not _lib.direct_grads_enabled([t]) or t.grad.shape != g.shape:
        return g
    _lib.call("geossl_axpy", _lib.ptr(t.grad), _lib.ptr(g.contiguous()), 1.0, g.numel(), _lib.ptr(t.grad), _lib.stream())
    return None


class _MM(torch.autograd.Function):
    """a @ b^T ("nt"), a @ b ("nn"), a^T @ b ("tn") with a (and for "tn" also b) a tall row matrix; feature widths are
    multiples of 8 up to 128.  The derivative of each form is made of the other two."""

    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.mode = mode
        ctx.save_for_backward(a, b)
        return _mm_raw(a, b, mode)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        mode, da, db = ctx.mode, None, None
        if mode == "nt":
            if ctx.needs_input_grad[0]:
                da = _MM.apply(g, b, "nn")
            if ctx.needs_input_grad[1]:
                db = _MM.apply(g, a, "tn")
        elif mode == "nn":
            if ctx.needs_input_grad[0]:
                da = _MM.apply(g, b, "nt")
            if ctx.needs_input_grad[1]:
                db = _MM.apply(a, g, "tn")
        else:
            if ctx.needs_input_grad[0]:
                da = _MM.apply(b, g, "nt")
            if ctx.needs_input_grad[1]:
                db = _MM.apply(a, g, "nn")
        return _to_leaf(a, da), _to_leaf(b, db), None


class _Agg(torch.autograd.Function):
    """propagate(aggr="add") of CFConv (schnet.py:190,194-195) in pair-slot form, or its transpose (swap)."""

    @staticmethod
    def forward(ctx, x, Wf, lay, pair_flag, swap):
        ctx.lay, ctx.pair_flag, ctx.swap = lay, pair_flag, swap
        ctx.save_for_backward(x, Wf)
        return ops.aggregate(x.contiguous(), Wf.contiguous(), pair_flag, lay, swap=swap)

    @staticmethod
    def backward(ctx, g):
        x, Wf = ctx.saved_tensors
        dx = _Agg.apply(g, Wf, ctx.lay, ctx.pair_flag, not ctx.swap) if ctx.needs_input_grad[0] else None
        dW = _PairProd.apply(g, x, ctx.lay, ctx.pair_flag, ctx.swap) if ctx.needs_input_grad[1] else None
        return dx, dW, None, None, None


class _PairProd(torch.autograd.Function):
    """out[p] = f0 a[i] b[j] + f1 a[j] b[i] over the pair slots p = (i < j): d aggregate / d filter."""

    @staticmethod
    def forward(ctx, a, b, lay, pair_flag, swap):
        ctx.lay, ctx.pair_flag, ctx.swap = lay, pair_flag, swap
        ctx.save_for_backward(a, b)
        return ops.pair_product(a.contiguous(), b.contiguous(), lay, pair_flag, swap)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _Agg.apply(b, g, ctx.lay, ctx.pair_flag, ctx.swap) if ctx.needs_input_grad[0] else None
        db = _Agg.apply(a, g, ctx.lay, ctx.pair_flag, not ctx.swap) if ctx.needs_input_grad[1] else None
        return da, db, None, None, None


class _LinearBias(torch.autograd.Function):
    """x @ w^T + b with the bias added in the GEMM's epilogue.  Its derivative is made of _MM forms and of the column
    sums of the upstream gradient (_ColSum, closed with its broadcast), so it is differentiable to any order."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.bias = b
        return _mm_raw(x, w, "nt", bias=b.detach())

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = _MM.apply(g, w, "nn") if ctx.needs_input_grad[0] else None
        dw = _MM.apply(g, x, "tn") if ctx.needs_input_grad[1] else None
        db = _ColSum.apply(g) if ctx.needs_input_grad[2] else None
        return dx, _to_leaf(w, dw), _to_leaf(ctx.bias, db)


class _ColSum(torch.autograd.Function):
    """Column sums of a row matrix (the bias gradient of a Linear) in fixed order on geossl_tape_reduce; its derivative
    is the broadcast below and vice versa."""

    @staticmethod
    def forward(ctx, g):
        from . import tape
        ctx.rows = g.size(0)
        return tape._raw_reduce(tape.COL, g.contiguous()).view(-1)

    @staticmethod
    def backward(ctx, c):
        return _RowBroadcast.apply(c, ctx.rows)


class _RowBroadcast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, rows):
        from . import tape
        return tape._raw_binary(tape.FIRST, c.contiguous().view(1, -1), tape.COL, None, tape.FULL, rows, c.numel())

    @staticmethod
    def backward(ctx, g):
        return _ColSum.apply(g), None


class _Silu(torch.autograd.Function):
    """F.silu on geossl_silu_fwd; the derivative is a node of its own (_SiluGrad) so that forces through an activation
    of the energy head stay on kernels under create_graph=True (finetune_md17.py:46,99)."""

    @staticmethod
    def forward(ctx, u):
        from ._lib import call, ptr, stream
        u = u.contiguous()
        ctx.save_for_backward(u)
        y = torch.empty_like(u)
        call("geossl_silu_fwd", ptr(u), u.numel(), ptr(y), stream())
        return y

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        return _SiluGrad.apply(u, g)


class _SiluGrad(torch.autograd.Function):
    """du = g * silu'(u) on geossl_silu_bwd; differentiated (training on forces): silu'' in closed form, torch ops."""

    @staticmethod
    def forward(ctx, u, g):
        from ._lib import call, ptr, stream
        g = g.contiguous()
        ctx.save_for_backward(u, g)
        du = torch.empty_like(u)
        call("geossl_silu_bwd", ptr(u), ptr(g), u.numel(), ptr(du), stream())
        return du

    @staticmethod
    def backward(ctx, c):
        u, g = ctx.saved_tensors
        if not torch.is_grad_enabled():   # second order (training on forces): silu' and silu'' as kernels
            from . import tape
            c, u = c.contiguous(), u.contiguous()   # the raw maps walk flat buffers of numel() elements
            shape, n = u.shape, u.numel()
            flat = lambda t: t.reshape(1, n)
            d1 = tape._raw_unary(tape.DSILU, u)
            d2 = tape._raw_unary(tape.D2SILU, u)
            cg = tape._raw_binary(tape.MUL, flat(c), tape.FULL, flat(g), tape.FULL, 1, n)
            du = tape._raw_binary(tape.MUL, cg, tape.FULL, flat(d2), tape.FULL, 1, n)
            dg = tape._raw_binary(tape.MUL, flat(c), tape.FULL, flat(d1), tape.FULL, 1, n)
            return du.view(shape), dg.view(shape)
        s = torch.sigmoid(u)                                # a third order is asked for: torch's own formulas carry it
        d1 = s * (1.0 + u * (1.0 - s))                      # silu'
        d2 = s * (1.0 - s) * (2.0 + u * (1.0 - 2.0 * s))    # silu''
        return c * g * d2, c * d1


def _silu(x):
    return _Silu.apply(x)


def _linear(x, w, b=None):
    return _MM.apply(x, w, "nt") if b is None else _LinearBias.apply(x, w, b)


def _ssp(x):
    return F.softplus(x) - SSP_SHIFT  # schnet.py:215-216


def schnet_atom_features(z, pos, lay, cfg, params):
    """schnet.py:89-101 (embedding .. head) as a graph of differentiable primitives; `params` in _core_params order."""
    L, Fd, G, cutoff = cfg["L"], cfg["F"], cfg["G"], cfg["cutoff"]
    emb_w, head = params[0], params[1 + 9 * L:]
    layers = [params[1 + 9 * l: 1 + 9 * (l + 1)] for l in range(L)]
    h = F.embedding(z, emb_w)                                                     # :89
    _, _, pair_flag = ops.pair_geometry(pos.detach(), lay, cutoff)                # :91 (the graph carries no gradient)
    pi, pj = lay.pair_i.long(), lay.pair_j.long()
    d = (pos[pi] - pos[pj]).norm(dim=-1)                                          # :93 for every pair slot
    offset = cfg["offset"]
    Gp = (G + 7) // 8 * 8                                                         # contraction width of the row GEMM
    rbf = torch.exp(cfg["coeff"] * (d.view(-1, 1) - offset.view(1, -1)) ** 2)     # :205-207
    rbf = F.pad(rbf, (0, Gp - G))
    C = 0.5 * (torch.cos(d * math.pi / cutoff) + 1.0)                             # :186
    for lp in layers:
        w1, b1, w2, b2, lin1_w, lin2_w, lin2_b, lin_w, lin_b = lp
        W = _linear(_ssp(_linear(rbf, F.pad(w1, (0, Gp - G)), b1)), w2, b2) * C.view(-1, 1)   # :187
        x = _linear(h, lin1_w)                                                    # :189
        x = _Agg.apply(x, W, lay, pair_flag, False)                               # :190
        x = _ssp(_linear(x, lin2_w, lin2_b))                                      # :191,165
        h = h + _linear(x, lin_w, lin_b)                                          # :166,97
    h = _ssp(_linear(h, head[0], head[1]))                                        # :99-100
    return _linear(h, head[2], head[3])                                           # :101


def _linear_wide(x, w, b=None):
    """x @ w^T + b for any widths (the padding / slabbing to the kernels' shapes happens inside _MM)."""
    return _linear(x, w, b)


def painn_atom_features(z, pos, idx_i, idx_j, cfg, params):
    """painn.py:230-255 (edge geometry .. last mixing block) as a graph of differentiable primitives - what
    PaiNNGradNode differentiates twice when a force is back-propagated (training on forces, finetune_md17.py:46-54):
    every Dense layer on the HIP row / column GEMMs, gathers, scatter-adds and element-wise glue in torch on the device.
    `params` in PaiNN._params() order."""
    Fd, L, cutoff = cfg["F"], cfg["L"], cfg["cutoff"]
    emb_w, fw, fb = params[0], params[1], params[2]
    inter = [params[3 + 4 * l: 7 + 4 * l] for l in range(L)]
    mix = [params[3 + 4 * L + 5 * l: 8 + 4 * L + 5 * l] for l in range(L)]
    n_atoms = pos.size(0)
    r_ij = pos[idx_i] - pos[idx_j]                                                # :232
    d_ij = torch.norm(r_ij, dim=1, keepdim=True)                                  # :236
    dir_ij = r_ij / d_ij                                                          # :237
    offsets, widths = cfg["offsets"], cfg["widths"]
    phi = torch.exp((-0.5 / widths ** 2) * (d_ij - offsets) ** 2)                 # painn_utils.py:99-102  [E, R]
    fcut = 0.5 * (torch.cos(d_ij * math.pi / cutoff) + 1.0) * (d_ij < cutoff).to(d_ij.dtype)   # :152-154
    filters = _linear_wide(phi, fw, fb) * fcut                                    # :241  [E, L*3F]
    q = F.embedding(z, emb_w, padding_idx=0)                                      # :247
    mu = torch.zeros(n_atoms, 3, Fd, dtype=q.dtype, device=q.device)              # :249
    for l in range(L):
        c0w, c0b, c1w, c1b = inter[l]
        x = _linear_wide(_silu(_linear_wide(q, c0w, c0b)), c1w, c1b)             # :53
        x = filters[:, l * 3 * Fd:(l + 1) * 3 * Fd] * x[idx_j]                    # :54,56
        dq, dmuR, dmumu = torch.split(x, Fd, dim=-1)                              # :58
        dq = torch.zeros_like(q).index_add(0, idx_i, dq)                          # :59
        dmu = dmuR[:, None, :] * dir_ij[..., None] + dmumu[:, None, :] * mu[idx_j]   # :60
        dmu = torch.zeros_like(mu).index_add(0, idx_i, dmu)                       # :61
        q, mu = q + dq, mu + dmu                                                  # :63-64
        i0w, i0b, i1w, i1b, mw = mix[l]
        mu_mix = _linear_wide(mu.reshape(3 * n_atoms, Fd), mw).reshape(n_atoms, 3, 2 * Fd)   # :100
        mu_V, mu_W = torch.split(mu_mix, Fd, dim=-1)                              # :101
        mu_Vn = torch.sqrt(torch.sum(mu_V ** 2, dim=-2) + cfg["eps"])             # :102
        x = _linear_wide(_silu(_linear_wide(torch.cat([q, mu_Vn], dim=-1), i0w, i0b)), i1w, i1b)   # :104-105
        dq_intra, dmu_intra, dqmu_intra = torch.split(x, Fd, dim=-1)              # :107
        q = q + dq_intra + dqmu_intra * torch.sum(mu_V * mu_W, dim=1)             # :110,112
        mu = mu + dmu_intra[:, None, :] * mu_W                                    # :108,113
    return q


def _deliver(fctx, out):
    """Inside _lib.direct_grads() (the caller owns dense p.grad buffers, e.g. after zero_grad(set_to_none=False)) the
    parameter gradients of the second-order route are added straight into p.grad - like the first-order node's - and the
    node returns none for them: no AccumulateGrad adds by the engine."""
    from . import _lib
    from ._lib import call, ptr, stream
    if torch.is_grad_enabled() or not _lib.direct_grads_enabled(fctx.params):
        return out
    out = list(out)
    for i, p in enumerate(fctx.params):
        g = out[2 + i]
        if g is not None:
            call("geossl_axpy", ptr(p.grad), ptr(g.contiguous()), 1.0, g.numel(), ptr(p.grad), stream())
            out[2 + i] = None
    return out


def _run_grad_node(node, fctx, dhout, want_pos, want_params):
    params = list(fctx.params)
    mask = [bool(want_pos)] + [bool(want_params and p.requires_grad) for p in params]
    outs = node.apply(fctx, mask, dhout, fctx.pos, *params)
    it = iter(outs)
    vals = [next(it) if m else None for m in mask]
    return vals[0], vals[1:]


def _second_order(features, tape_features, mask, need, cot, dhout, pos, params):
    """The derivative of the first-order gradients (d pos, d params given d h), contracted with their cotangents `cot`:
    the primitive restatement of the backbone differentiated twice - on the library's own tape (geossl_amd/tape.py:
    every primitive a HIP kernel, no autograd engine), or, when the caller asks for a graph of THIS derivative too (a
    third order) or GEOSSL_SECOND_ORDER=torch, as a torch autograd graph over `features(pos, params)`.  `need`: which of
    (dhout, pos, *params) get a gradient."""
    higher = torch.is_grad_enabled()  # read OUTSIDE the block below: a third-order graph only if the caller wants one
    if not higher and _env("GEOSSL_SECOND_ORDER", "tape") != "torch":
        from . import tape
        return tape.second_order(tape_features, mask, need, cot, dhout, pos, params)
    with torch.enable_grad():
        dh = dhout.detach().requires_grad_(need[0])
        ps = [p.detach().requires_grad_(True) for p in params]
        x = pos.detach().requires_grad_(True)
        h = features(x, ps)
        wrt = [t for t, m in zip([x] + ps, mask) if m]
        first = torch.autograd.grad(h, wrt, grad_outputs=dh, create_graph=True, allow_unused=True)
        s = None
        for f, c in zip(first, cot):
            if f is not None and c is not None:
                term = (f * c).sum()
                s = term if s is None else s + term
        ins = [t for t, n in zip([dh, x] + ps, need) if n]
        if s is None or not ins:
            second = [None] * len(ins)
        else:
            second = torch.autograd.grad(s, ins, allow_unused=True, create_graph=higher)
    it = iter(second)
    return [next(it) if n else None for n in need]


class SchNetGradNode(torch.autograd.Function):
    """The first-order gradients of the fused SchNet node (d pos, d params given d h) as a node of their own: forward =
    the fused kernels (fast; all an evaluation loop needs), backward = the derivative of those gradients, obtained by
    differentiating the primitive restatement above twice.  ``mask`` says which of (pos, *params) get a gradient."""

    @staticmethod
    def run(fctx, dhout, want_pos, want_params):
        return _run_grad_node(SchNetGradNode, fctx, dhout, want_pos, want_params)

    @staticmethod
    def forward(ctx, fctx, mask, dhout, pos, *params):
        from .Geom3D.models.schnet import _SchNetCore
        dpos, grads = _SchNetCore.fused_backward(fctx, dhout, mask[0], any(mask[1:]), allow_direct=False)
        ctx.fctx, ctx.mask = fctx, mask
        ctx.set_materialize_grads(False)   # an output nobody differentiates arrives as None, not as a zero tensor
        ctx.save_for_backward(dhout, pos, *params)
        vals = [dpos] + list(grads)
        return tuple(v for v, m in zip(vals, mask) if m)

    @staticmethod
    def backward(ctx, *cot):
        fctx = ctx.fctx
        dhout, pos, *params = ctx.saved_tensors
        from . import tape
        out = _second_order(lambda x, ps: schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps),
                            lambda x, ps: tape.schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps), ctx.mask,
                            ctx.needs_input_grad[2:], cot, dhout, pos, params)
        return (None, None) + tuple(_deliver(fctx, out))


class PaiNNGradNode(torch.autograd.Function):
    """The same for the fused PaiNN node: forward = its first-order kernels (painn.hip, painn_force.hip), backward =
    painn_atom_features differentiated twice."""

    @staticmethod
    def run(fctx, dq, want_pos, want_params):
        return _run_grad_node(PaiNNGradNode, fctx, dq, want_pos, want_params)

    @staticmethod
    def forward(ctx, fctx, mask, dq, pos, *params):
        from .Geom3D.models.painn import _PaiNNCore
        dpos, grads = _PaiNNCore.fused_backward(fctx, dq, mask[0], any(mask[1:]), allow_direct=False)
        ctx.fctx, ctx.mask = fctx, mask
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(dq, pos, *params)
        vals = [dpos] + list(grads)
        return tuple(v for v, m in zip(vals, mask) if m)

    @staticmethod
    def backward(ctx, *cot):
        fctx = ctx.fctx
        dq, pos, *params = ctx.saved_tensors
        el = fctx.el
        from . import tape
        out = _second_order(lambda x, ps: painn_atom_features(fctx.z, x, el.idx_i, el.idx_j, fctx.cfg, ps),
                            lambda x, ps: tape.painn_atom_features(fctx.z, x, el.idx_i, el.idx_j, fctx.cfg, ps, el.inc), ctx.mask,
                            ctx.needs_input_grad[2:], cot, dq, pos, params)
        return (None, None) + tuple(_deliver(fctx, out))
