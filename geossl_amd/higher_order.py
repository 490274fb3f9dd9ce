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
PyTorch's own derivative formulas carry the higher orders there.  Nothing here is on the DDM hot path.
"""
import math

import torch
import torch.nn.functional as F

from . import ops

SSP_SHIFT = torch.log(torch.tensor(2.0)).item()  # schnet.py:213


def _mm_raw(a, b, mode):
    a, b = a.contiguous(), b.contiguous()
    if mode == "nt":    # a [R, K] @ b[NO, K]^T
        return ops.linear(a, b, transB=True)
    if mode == "nn":    # a [R, K] @ b[K, NO]
        return ops.linear(a, b, transB=False)
    R, M, N = a.size(0), a.size(1), b.size(1)  # "tn": a[R, M]^T @ b[R, N], tiled to the column GEMM's 128 x 128 limit
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    for m0 in range(0, M, 128):
        for n0 in range(0, N, 128):
            mm, nn = min(128, M - m0), min(128, N - n0)
            ops.linear_wgrad([(a[:, m0:], b[:, n0:], out[m0:, n0:], None)], R, mm, nn, lda=M, ldb=N, ldw=N)
    return out


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
        return da, db, None


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


def _linear(x, w, b=None):
    y = _MM.apply(x, w, "nt")
    return y if b is None else y + b


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


class SchNetGradNode(torch.autograd.Function):
    """The first-order gradients of the fused SchNet node (d pos, d params given d h) as a node of their own: forward =
    the fused kernels (fast; all an evaluation loop needs), backward = the derivative of those gradients, obtained by
    differentiating the primitive restatement above twice.  ``mask`` says which of (pos, *params) get a gradient."""

    @staticmethod
    def run(fctx, dhout, want_pos, want_params):
        params = list(fctx.params)
        mask = [bool(want_pos)] + [bool(want_params and p.requires_grad) for p in params]
        outs = SchNetGradNode.apply(fctx, mask, dhout, fctx.pos, *params)
        it = iter(outs)
        vals = [next(it) if m else None for m in mask]
        return vals[0], vals[1:]

    @staticmethod
    def forward(ctx, fctx, mask, dhout, pos, *params):
        from .Geom3D.models.schnet import _SchNetCore
        dpos, grads = _SchNetCore.fused_backward(fctx, dhout, mask[0], any(mask[1:]), allow_direct=False)
        ctx.fctx, ctx.mask = fctx, mask
        ctx.save_for_backward(dhout, pos, *params)
        vals = [dpos] + list(grads)
        return tuple(v for v, m in zip(vals, mask) if m)

    @staticmethod
    def backward(ctx, *cot):
        fctx, mask = ctx.fctx, ctx.mask
        dhout, pos, *params = ctx.saved_tensors
        need = ctx.needs_input_grad[2:]  # (dhout, pos, *params)
        with torch.enable_grad():
            dh = dhout.detach().requires_grad_(need[0])
            ps = [p.detach().requires_grad_(True) for p in params]
            x = pos.detach().requires_grad_(True)
            h = schnet_atom_features(fctx.z, x, fctx.lay, fctx.cfg, ps)
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
                second = torch.autograd.grad(s, ins, allow_unused=True, create_graph=torch.is_grad_enabled())
        it = iter(second)
        out = [next(it) if n else None for n in need]
        return (None, None) + tuple(out)
