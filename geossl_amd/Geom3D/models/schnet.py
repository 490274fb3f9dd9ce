"""SchNet behind the reference's own interface (Geom3D/models/schnet.py), computed by the HIP path.

Same constructor, ``forward(z, pos, batch=None, return_latent=False)`` signature, parameter
registration order and ``state_dict`` keys as the reference class (schnet.py:16-135), including its
quirks: the filter network is registered twice (``interactions.i.mlp`` and
``interactions.i.conv.nn`` are one module, :141-148), ``mlp[2].bias`` keeps the default Linear
init (:155-158), the head is Linear(F,F) -> ssp -> Linear(F,F) (:62-64).

The arithmetic does not run in PyTorch: one autograd node wraps the whole message-passing stack and
drives the kernels of libgeossl_hip.so (radius graph in pair-slot form, continuous-filter network
for all blocks in one launch, neighbour aggregation, MFMA atom-row Linears, and the matching
backward).  CUDA tensors only — there is no CPU fallback.
"""
import ctypes as C
import math
import os

import torch
from ...switches import env as _env
from torch.nn import Embedding, Linear, ModuleList, Sequential

from ... import _lib, ops
from ..._lib import call, ptr, stream
from ...layout import get_layout
from ._atomic_mass import atomic_masses

SUPPORTED_F = (32, 64, 128)


class _Ssp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        call("geossl_ssp_fwd", ptr(x), x.numel(), ptr(y), stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        call("geossl_ssp_bwd", ptr(y), ptr(dy.contiguous()), y.numel(), ptr(dx), stream())
        return dx


class ShiftedSoftplus(torch.nn.Module):
    """schnet.py:210-216.  Inside SchNet it is evaluated by the fused kernels (GEMM epilogues); the module keeps the
    Sequential indices (mlp.0 / mlp.2) and the module tree of the reference, and called on its own it runs the
    same device function as an element-wise launch."""

    def __init__(self):
        super().__init__()
        self.shift = torch.log(torch.tensor(2.0)).item()

    def forward(self, x):
        _lib.require_cuda(x)
        if x.dtype != torch.float32:
            raise TypeError("expected float32, got %s" % x.dtype)
        return _Ssp.apply(x)


class GaussianSmearing(torch.nn.Module):
    """schnet.py:198-207: offset buffer + Python-double coeff from the fp32 grid spacing."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, dist):
        return ops.gaussian_smearing(dist, self.offset, self.coeff)


class CFConv(torch.nn.Module):
    """Parameter holder with the reference's names (schnet.py:170-183); message passing itself is
    geossl_cfconv_filter_fwd + geossl_cfconv_aggregate."""

    def __init__(self, in_channels, out_channels, num_filters, nn, cutoff):
        super().__init__()
        self.lin1 = Linear(in_channels, num_filters, bias=False)
        self.lin2 = Linear(num_filters, out_channels)
        self.nn = nn
        self.cutoff = cutoff
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        torch.nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)


class InteractionBlock(torch.nn.Module):
    def __init__(self, hidden_channels, num_gaussians, num_filters, cutoff):
        super().__init__()
        self.mlp = Sequential(Linear(num_gaussians, num_filters), ShiftedSoftplus(), Linear(num_filters, num_filters))
        self.conv = CFConv(hidden_channels, hidden_channels, num_filters, self.mlp, cutoff)
        self.act = ShiftedSoftplus()
        self.lin = Linear(hidden_channels, hidden_channels)
        self.reset_parameters()

    def reset_parameters(self):
        # mirrors schnet.py:154-161 including the quirk that mlp[2].bias is never zeroed
        torch.nn.init.xavier_uniform_(self.mlp[0].weight)
        self.mlp[0].bias.data.fill_(0)
        torch.nn.init.xavier_uniform_(self.mlp[2].weight)
        self.conv.reset_parameters()
        torch.nn.init.xavier_uniform_(self.lin.weight)
        self.lin.bias.data.fill_(0)


def _core_params(model):
    ps = [model.embedding.weight]
    for blk in model.interactions:
        ps += [blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias, blk.conv.lin1.weight,
               blk.conv.lin2.weight, blk.conv.lin2.bias, blk.lin.weight, blk.lin.bias]
    ps += [model.lin1.weight, model.lin1.bias, model.lin2.weight, model.lin2.bias]
    return ps


class _SchNetCore(torch.autograd.Function):
    """(z, pos) -> atom features after the head (schnet.py:89-101) as ONE autograd node."""

    @staticmethod
    def forward(ctx, z, pos, lay, cfg, *params):
        L, F, G = cfg["L"], cfg["F"], cfg["G"]
        dev = pos.device
        N = pos.size(0)
        want_params = any(ctx.needs_input_grad[4:])
        training = want_params or ctx.needs_input_grad[1]  # activations are kept for either backward
        ps = [p.detach().contiguous() for p in params]
        emb_w, head = ps[0], ps[1 + 9 * L:]
        layers = [ps[1 + 9 * l: 1 + 9 * (l + 1)] for l in range(L)]
        st = stream()
        # embedding (schnet.py:89); z is usually the strided view x[:, 0]
        h = torch.empty(N, F, dtype=torch.float32, device=dev)
        # an atom type outside the table raises in the reference (Embedding); here the kernel flags it in the model's
        # status word, which forward() polls without draining the stream (synchronously under GEOSSL_DEBUG)
        status = cfg["status"]
        # a layout of a capacity bucket (geossl_amd/bucket.py) carries the device addresses of the batch's real counts:
        # N, P are then capacities and every row-count-driven launch below is told where the real count lives
        dyn = getattr(lay, "dyn", None)
        dN2, dP2 = (dyn.n_atoms2, dyn.n_pairs2) if dyn is not None else (None, None)
        if dyn is not None and (ctx.needs_input_grad[1] or F != 128 or not cfg["chain"]):
            raise _lib.GeosslHipError("a capacity-bucket layout serves the F = 128 chain path without position gradients")
        call("geossl_embedding_fwd_dyn", ptr(z), z.stride(0) if z.numel() else 1, ptr(emb_w), emb_w.size(0), N, F, ptr(h),
             ptr(status.word), dN2, st)
        if cfg["debug"]:
            status.check()
        # radius graph + edge length + envelope (schnet.py:91-93,186)
        pair_d, pair_c, pair_flag = ops.pair_geometry(pos, lay, cfg["cutoff"])
        P = lay.P
        # continuous-filter network of every block in one launch (schnet.py:94,187)
        fw = _lib.FilterWeights()
        for l, lp in enumerate(layers):
            fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = ptr(lp[0]), ptr(lp[1]), ptr(lp[2]), ptr(lp[3])
        Wf = torch.empty(L, P, F, dtype=torch.float32, device=dev)
        # The hidden rows T = ssp(W1 rbf + b1) of the filter network are saved for the backward (0.96 GB written and read
        # again per step at the bench size).  They are a function of the pair's distance alone, and the weight-gradient
        # kernel can rebuild them (GEOSSL_FILTER_RECOMPUTE_T: one K = 64 product per tile by its role-A waves; the forward
        # then stores Wf only) - measured on two boxes: forward 0.41 -> 0.31 ms, backward 0.74 -> 0.88 ms, the step
        # within 0.1 % either way, 21 % less HBM traffic.  Equal speed is not a win: saving stays the default, the
        # rebuild is there for when memory is what is short.  The position gradient
        # (geossl_cfconv_filter_dpos) and the three-bf16-piece backward read the saved rows.
        keep_T = training and (ctx.needs_input_grad[1] or not _env("GEOSSL_FILTER_RECOMPUTE_T")
                               or bool(_env("GEOSSL_FILTER_BWD_BF16X3"))
                               or bool(_env("GEOSSL_ARITH_24BIT")))
        T = torch.empty(L, P, F, dtype=torch.float32, device=dev) if keep_T else None
        if P > 0:
            call("geossl_cfconv_filter_fwd_dyn", ptr(pair_d), ptr(pair_c), P, C.byref(fw), L, F, G, ptr(cfg["offset"]),
                 cfg["coeff"], ptr(T), ptr(Wf), dP2, st)
        hs, xs, aggs, ts = [], [], [], []
        heads = None
        if cfg["chain"]:
            # The row-local layers between two aggregations run as ONE launch each (geossl_linear_chain): conv.lin2 + act,
            # lin + residual and the next block's conv.lin1 (after the last block: the head).  Operand images of all
            # 3L + 2 square weights from one launch.  (Running the two views of a DDM step as two parallel graph branches
            # - lock-step or staggered by one aggregation - was measured and gave nothing: DESIGN.md section 7.)
            chain_w = [lp[k] for lp in layers for k in (4, 5, 7)] + [head[0], head[2]]
            if want_params and 2 * len(chain_w) <= _lib.PREPARE_MAX:  # the backward's images from the same launch
                img, img_bwd = ops.prepare_chain(chain_w, both=True)
            else:
                img, img_bwd = ops.prepare_chain(chain_w, transB=True), None
            i_lin1, i_lin2, i_lin = img[0:3 * L:3], img[1:3 * L:3], img[2:3 * L:3]
            full = lambda: torch.empty(N, F, dtype=torch.float32, device=dev)
            hs, xs, aggs, ts = [h] + [full() for _ in range(L)], [full() for _ in range(L)], [full() for _ in range(L)], \
                [full() for _ in range(L)]
            u, hout = full(), full()

            def run_rows(a0, a1, mols, todo=None):
                """the operations of the layer loop: launched one by one, or collected in `todo`"""
                rows = lambda t_: t_[a0:a1]

                def chain(x_, stages):
                    if todo is not None:
                        todo.append(("chain", x_, stages))
                    else:
                        ops.linear_chain(x_, stages, dyn_rows=dN2)

                chain(rows(hs[0]), [dict(image=i_lin1[0], out=rows(xs[0]))])                           # conv.lin1   :189
                for l, lp in enumerate(layers):
                    if todo is not None:
                        todo.append(("agg", xs[l], Wf[l], aggs[l], False))
                    else:
                        ops.aggregate(xs[l], Wf[l], pair_flag, lay, out=aggs[l], mols=mols)            # propagate   :190
                    stages = [dict(image=i_lin2[l], bias=lp[6], flags=_lib.EPI_SSP, out=rows(ts[l])),  # conv.lin2 + act
                              dict(image=i_lin[l], bias=lp[8], res=rows(hs[l]), out=rows(hs[l + 1]))]  # lin + residual
                    if l + 1 < L:
                        stages.append(dict(image=i_lin1[l + 1], out=rows(xs[l + 1])))                  # next conv.lin1
                    else:
                        stages.append(dict(image=img[3 * L], bias=head[1], flags=_lib.EPI_SSP, out=rows(u)))  # :99-100
                    chain(rows(aggs[l]), stages)
                chain(rows(u), [dict(image=img[3 * L + 1], bias=head[3], out=rows(hout))])             # lin2        :101

            # One launch for the whole loop where the shape allows (every block carries its molecules through all
            # operations, ops.layer_loop), else 14 launches
            todo = [] if (cfg["loop"] and P > 0 and 2 * L + 2 <= _lib.LOOP_MAX_OPS) else None
            if todo is not None:
                run_rows(0, N, None, todo)
                if not ops.layer_loop(todo, lay, pair_flag, N, F, stagger=cfg["loop_stagger"]):
                    todo = None
            if todo is None:
                run_rows(0, N, None)
            h = hs[L]
            hs = hs[:L]
            if not training:
                hs, xs, aggs, ts = [], [], [], []
        else:
            # operand images of the 3L square Linear weights, one launch (each is used by a launch over all atoms of
            # both views; geossl_linear would otherwise re-shape it in every block)
            pw = ops.prepare_linear([lp[k] for lp in layers for k in (4, 5, 7)], transB=True)
            if pw is not None:
                layers = [lp[:4] + [pw[3 * l], pw[3 * l + 1], lp[6], pw[3 * l + 2], lp[8]] for l, lp in enumerate(layers)]
            for l, lp in enumerate(layers):
                x = ops.linear(h, lp[4])                                    # conv.lin1 (no bias)   :189
                agg = ops.aggregate(x, Wf[l], pair_flag, lay)               # propagate(add)        :190
                t = ops.linear(agg, lp[5], bias=lp[6], flags=_lib.EPI_SSP)  # conv.lin2 + act       :191,165
                hn = ops.linear(t, lp[7], bias=lp[8], res=h)                # lin + residual        :166,97
                if training:
                    hs.append(h); xs.append(x); aggs.append(agg); ts.append(t)
                h = hn
            u = ops.linear(h, head[0], bias=head[1], flags=_lib.EPI_SSP)    # lin1 + act            :99-100
            hout = ops.linear(u, head[2], bias=head[3])                     # lin2                  :101
        if training:
            ctx.lay, ctx.cfg = lay, cfg
            ctx.z = z
            ctx.pos = pos
            ctx.want_params = want_params
            ctx.ps = ps
            ctx.params = params
            ctx.saved = dict(pair_d=pair_d, pair_c=pair_c, pair_flag=pair_flag, Wf=Wf, T=T, hs=hs, xs=xs, aggs=aggs,
                             ts=ts, h_last=h, u=u, img_bwd=img_bwd if cfg["chain"] else None)
        return hout

    @staticmethod
    def backward(ctx, dhout):
        want_pos, want_params = ctx.needs_input_grad[1], ctx.want_params
        if torch.is_grad_enabled():
            # autograd runs a backward with grad mode ON only under create_graph=True (finetune_md17.py:46,99): the
            # gradients may be differentiated again (training on forces, :51-54).  They are still computed by the fused
            # first-order kernels, but as the outputs of a node that knows how to be differentiated
            # (geossl_amd/higher_order.py); evaluation loops that detach the force never pay for that.
            from ...higher_order import SchNetGradNode
            dpos, grads = SchNetGradNode.run(ctx, dhout, want_pos, want_params)
            return (None, dpos, None, None) + tuple(grads)
        dpos, grads = _SchNetCore.fused_backward(ctx, dhout, want_pos, want_params, allow_direct=True)
        return (None, dpos, None, None) + tuple(grads)

    @staticmethod
    def fused_backward(ctx, dhout, want_pos, want_params, allow_direct):
        """First-order gradients by the fused kernels -> (dpos or None, [one entry per parameter, None = not returned])."""
        cfg, lay, sv, ps = ctx.cfg, ctx.lay, ctx.saved, ctx.ps
        L, F, G = cfg["L"], cfg["F"], cfg["G"]
        dev = dhout.device
        N = dhout.size(0)
        st = stream()
        emb_w, head = ps[0], ps[1 + 9 * L:]
        layers = [ps[1 + 9 * l: 1 + 9 * (l + 1)] for l in range(L)]
        # Inside _lib.direct_grads() (DDMTrainer, which owns the flat gradient buffer behind every p.grad) the kernels
        # accumulate straight into p.grad and the node returns no parameter gradients: no temporaries, no
        # AccumulateGrad adds.  Everywhere else the gradients go back through autograd like any other node's.
        direct = want_params and allow_direct and _lib.direct_grads_enabled(ctx.params)
        grads = [p.grad for p in ctx.params] if direct else [torch.empty_like(p) if want_params else None for p in ps]
        accum = 1 if direct else 0
        g_emb, g_head = grads[0], grads[1 + 9 * L:]
        g_layers = [grads[1 + 9 * l: 1 + 9 * (l + 1)] for l in range(L)]
        dh_out = dhout.contiguous()
        dyn = getattr(lay, "dyn", None)
        dN2, dP2 = (dyn.n_atoms2, dyn.n_pairs2) if dyn is not None else (None, None)
        probs = []  # (A = dY, B = X, dW, db)
        daggs = [None] * L
        if cfg["chain"]:
            # the same chains walked backwards: [head.lin2 + act', head.lin1], [lin_{L-1} + act', conv.lin2_{L-1}], then per
            # block  dX through conv.lin1_l (+ the residual branch), lin_{l-1} + act', conv.lin2_{l-1}
            img = sv.get("img_bwd")  # converted with the forward's images (the weights have not changed since)
            if img is None:
                img = ops.prepare_chain([lp[k] for lp in layers for k in (4, 5, 7)] + [head[0], head[2]], transB=False)
            i_lin1, i_lin2, i_lin = img[0:3 * L:3], img[1:3 * L:3], img[2:3 * L:3]
            full = lambda: torch.empty(N, F, dtype=torch.float32, device=dev)
            du = full()
            dhs = [full() for _ in range(L + 1)]      # dhs[l] = gradient at the input of block l (dhs[L]: at the head's input)
            dys, dxs = [full() for _ in range(L)], [full() for _ in range(L)]
            daggs = [full() for _ in range(L)]

            def run_rows(a0, a1, mols, todo=None):
                rows = lambda t_: t_[a0:a1]

                def chain(x_, stages):
                    if todo is not None:
                        todo.append(("chain", x_, stages))
                    else:
                        ops.linear_chain(x_, stages, dyn_rows=dN2)

                chain(rows(dh_out), [dict(image=img[3 * L + 1], tprev=rows(sv["u"]), out=rows(du)),
                                     dict(image=img[3 * L], out=rows(dhs[L]))])
                chain(rows(dhs[L]), [dict(image=i_lin[L - 1], tprev=rows(sv["ts"][L - 1]), out=rows(dys[L - 1])),
                                     dict(image=i_lin2[L - 1], out=rows(daggs[L - 1]))])
                for l in reversed(range(L)):
                    if todo is not None:
                        todo.append(("agg", daggs[l], sv["Wf"][l], dxs[l], True))
                    else:
                        ops.aggregate(daggs[l], sv["Wf"][l], sv["pair_flag"], lay, swap=True, out=dxs[l],
                                      mols=mols)                                    # transposed graph
                    stages = [dict(image=i_lin1[l], res=rows(dhs[l + 1]), out=rows(dhs[l]))]      # conv.lin1 + residual
                    if l > 0:
                        stages += [dict(image=i_lin[l - 1], tprev=rows(sv["ts"][l - 1]), out=rows(dys[l - 1])),
                                   dict(image=i_lin2[l - 1], out=rows(daggs[l - 1]))]
                    chain(rows(dxs[l]), stages)

            todo = [] if (cfg["loop"] and lay.P > 0 and 2 * L + 2 <= _lib.LOOP_MAX_OPS) else None
            if todo is not None:
                run_rows(0, N, None, todo)
                if not ops.layer_loop(todo, lay, sv["pair_flag"], N, F, stagger=cfg["loop_stagger"]):
                    todo = None
            if todo is None:
                run_rows(0, N, None)
            probs.append((dh_out, sv["u"], g_head[2], g_head[3]))
            probs.append((du, sv["h_last"], g_head[0], g_head[1]))
            for l in reversed(range(L)):
                gl = g_layers[l]
                probs.append((dhs[l + 1], sv["ts"][l], gl[7], gl[8]))
                probs.append((dys[l], sv["aggs"][l], gl[5], gl[6]))
                probs.append((dxs[l], sv["hs"][l], gl[4], None))
            dh = dhs[0]
        else:
            # head: hout = u W2^T + b2, u = ssp(h W1^T + b1)
            du = ops.linear(dh_out, head[2], transB=False, tprev=sv["u"])
            dh = ops.linear(du, head[0], transB=False)
            probs.append((dh_out, sv["u"], g_head[2], g_head[3]))
            probs.append((du, sv["h_last"], g_head[0], g_head[1]))
            pw = ops.prepare_linear([lp[k] for lp in layers for k in (4, 5, 7)], transB=False)  # backward-input images
            for l in reversed(range(L)):
                lp, gl = layers[l], g_layers[l]
                w_lin1, w_lin2, w_lin = (pw[3 * l], pw[3 * l + 1], pw[3 * l + 2]) if pw is not None else (lp[4], lp[5], lp[7])
                dy = ops.linear(dh, w_lin, transB=False, tprev=sv["ts"][l])        # through lin and act
                dagg = ops.linear(dy, w_lin2, transB=False)                        # through conv.lin2
                dx = ops.aggregate(dagg, sv["Wf"][l], sv["pair_flag"], lay, swap=True)  # transposed graph
                dh_new = ops.linear(dx, w_lin1, transB=False, res=dh)              # through conv.lin1 + residual
                probs.append((dh, sv["ts"][l], gl[7], gl[8]))
                probs.append((dy, sv["aggs"][l], gl[5], gl[6]))
                probs.append((dx, sv["hs"][l], gl[4], None))
                daggs[l] = dagg
                dh = dh_new
        P = lay.P
        fw = _lib.FilterWeights()
        gin = _lib.FilterGradIn()
        for l, lp in enumerate(layers):
            fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = ptr(lp[0]), ptr(lp[1]), ptr(lp[2]), ptr(lp[3])
            gin.x[l], gin.dagg[l] = ptr(sv["xs"][l]), ptr(daggs[l])
        if want_params:
            # every atom-row weight gradient in one batched launch
            ops.linear_wgrad(probs, N, F, F, accumulate=bool(accum), dyn_rows=dN2)
            # embedding table
            nfl = _lib.load().geossl_embedding_bwd_workspace_floats(emb_w.size(0), F)
            ws = torch.empty(nfl, dtype=torch.float32, device=dev)
            z = ctx.z
            call("geossl_embedding_bwd_dyn", ptr(z), z.stride(0) if z.numel() else 1, ptr(dh), emb_w.size(0), N, F,
                 ptr(g_emb), ptr(ws), accum, dN2, st)
            # continuous-filter network weights, all blocks at once
            if P > 0:
                gout = _lib.FilterGradOut()
                for l, gl in enumerate(g_layers):
                    gout.dw1[l], gout.db1[l], gout.dw2[l], gout.db2[l] = ptr(gl[0]), ptr(gl[1]), ptr(gl[2]), ptr(gl[3])
                nfl = _lib.load().geossl_cfconv_filter_bwd_workspace_floats(P, L, F, G)
                ws2 = torch.empty(nfl, dtype=torch.float32, device=dev)
                call("geossl_cfconv_filter_bwd_dyn", ptr(sv["pair_d"]), ptr(sv["pair_c"]), ptr(sv["pair_flag"]),
                     ptr(lay.pair_i), ptr(lay.pair_j), P, N, C.byref(fw), C.byref(gin), L, F, G, ptr(cfg["offset"]),
                     cfg["coeff"], ptr(sv["T"]), C.byref(gout), ptr(ws2), accum, dP2, dN2, st)
            elif not direct:
                for gl in g_layers:
                    for k in range(4):
                        gl[k].zero_()
        dpos = None
        if want_pos:
            # positions enter through the edge lengths only (schnet.py:93): dL/dd per pair slot and block, then the
            # derivative of the norm applied to both atoms of every slot (finetune_md17.py:46, first order)
            dpos = torch.zeros(N, 3, dtype=torch.float32, device=dev)
            if P > 0:
                dd = torch.empty(L, P, dtype=torch.float32, device=dev)
                call("geossl_cfconv_filter_dpos", ptr(sv["pair_d"]), ptr(sv["pair_c"]), ptr(sv["pair_flag"]),
                     ptr(lay.pair_i), ptr(lay.pair_j), P, C.byref(fw), C.byref(gin), L, F, G, ptr(cfg["offset"]),
                     cfg["coeff"], cfg["cutoff"], ptr(sv["T"]), ptr(sv["Wf"]), ptr(dd), st)
                call("geossl_pair_position_grad", ptr(ctx.pos), ptr(sv["pair_d"]), ptr(dd), ptr(lay.mol_ptr),
                     ptr(lay.pair_ptr), lay.B, P, L, ptr(dpos), st)
        # ctx.saved stays: finetune_md17.py:46 differentiates with retain_graph=True and runs this node again
        if direct:
            return dpos, [None] * len(grads)
        return dpos, list(grads)


class _SchNetTapeCore(torch.autograd.Function):
    """schnet.py:89-101 for the widths the fused kernels do not take (hidden_channels != num_filters, widths other than 32
    / 64 / 128, more than 64 gaussians or 12 blocks - the reference's constructor takes any, schnet.py:17-30): the
    backbone restated on the library's reverse-mode tape (geossl_amd/tape.py: every primitive a HIP kernel - row GEMMs at
    any width in padded slabs, the aggregation in column slabs, element-wise maps), differentiated by the tape.  About a
    hundred launches per block instead of three: a correct path for every configuration, not a fast one.  First order
    only (training on forces with such a configuration raises)."""

    @staticmethod
    def forward(ctx, z, pos, lay, cfg, *params):
        from ... import tape as tp
        with torch.no_grad():
            x = tp.leaf(pos, bool(ctx.needs_input_grad[1]))
            ps = [tp.leaf(p, bool(n)) for p, n in zip(params, ctx.needs_input_grad[4:])]
            h = tp.schnet_atom_features(z, x, lay, cfg, ps)
        ctx.tape = (h, x, ps)
        ctx.shapes = [tuple(p.shape) for p in params]
        return h.t

    @staticmethod
    def backward(ctx, dh):
        from ... import tape as tp
        if torch.is_grad_enabled():
            raise NotImplementedError("second-order gradients (training on forces) need the fused widths: "
                                      "hidden_channels == num_filters in %s" % (SUPPORTED_F,))
        h, x, ps = ctx.tape
        wrt = [v for v in [x] + ps if v.req]
        with torch.no_grad():
            got = iter(tp.grad([h], [tp.const(dh.contiguous())], wrt))
        vals = [next(got) if v.req else None for v in [x] + ps]
        dpos = None if vals[0] is None else vals[0].t
        grads = [None if g is None else g.t.reshape(shape) for g, shape in zip(vals[1:], ctx.shapes)]
        # (a parameter the output does not depend on - none here - would come back as None: autograd takes that as zero)
        return (None, dpos, None, None) + tuple(grads)


class _SegmentReduce(torch.autograd.Function):
    """torch_scatter.scatter(h, batch, dim=0, reduce) for a sorted batch (schnet.py:115).  Its backward is the
    expansion below and vice versa, so the pair is differentiable to any order (training on forces differentiates the
    readout's backward with respect to its upstream gradient, i.e. the weights of the energy head)."""

    @staticmethod
    def forward(ctx, h, lay, reduce):
        ctx.lay, ctx.reduce = lay, reduce
        return ops.segment_reduce(h.contiguous(), lay, reduce)

    @staticmethod
    def backward(ctx, dout):
        return _SegmentExpand.apply(dout, ctx.lay, ctx.reduce), None, None


class _SegmentExpand(torch.autograd.Function):
    """dh[a] = dout[molecule of a] (/ atoms of the molecule for "mean"): the adjoint of _SegmentReduce."""

    @staticmethod
    def forward(ctx, dout, lay, reduce):
        ctx.lay, ctx.reduce = lay, reduce
        dh = torch.empty(lay.N, dout.size(1), dtype=torch.float32, device=dout.device)
        call("geossl_segment_reduce_bwd", ptr(dout.contiguous()), ptr(lay.mol_ptr), lay.B, dout.size(1),
             1 if reduce == "mean" else 0, ptr(dh), 0, stream())
        return dh

    @staticmethod
    def backward(ctx, g):
        return _SegmentReduce.apply(g, ctx.lay, ctx.reduce), None, None


class SchNet(torch.nn.Module):
    def __init__(self, hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=50, cutoff=10.0,
                 node_class=None, readout="mean", dipole=False, mean=None, std=None, atomref=None):
        super().__init__()
        assert readout in ["add", "sum", "mean"]
        self.hidden_channels = hidden_channels
        self.num_filters = num_filters
        self.num_interactions = num_interactions
        self.num_gaussians = num_gaussians
        self.cutoff = cutoff
        self.readout = readout
        self.dipole = dipole
        self.readout = "add" if self.dipole else self.readout
        self.mean = mean
        self.std = std
        self.scale = None

        self.register_buffer("atomic_mass", atomic_masses())
        self.embedding = Embedding(node_class, hidden_channels)
        self.distance_expansion = GaussianSmearing(0.0, cutoff, num_gaussians)
        self.interactions = ModuleList()
        for _ in range(num_interactions):
            self.interactions.append(InteractionBlock(hidden_channels, num_gaussians, num_filters, cutoff))
        self.lin1 = Linear(hidden_channels, hidden_channels)
        self.act = ShiftedSoftplus()
        self.lin2 = Linear(hidden_channels, hidden_channels)
        self.register_buffer("initial_atomref", atomref)
        self.atomref = None
        if atomref is not None:
            self.atomref = Embedding(100, 1)
            self.atomref.weight.data.copy_(atomref)
        self.reset_parameters()

    def reset_parameters(self):
        self.embedding.reset_parameters()
        for interaction in self.interactions:
            interaction.reset_parameters()
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        self.lin1.bias.data.fill_(0)
        torch.nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)
        if self.atomref is not None:
            self.atomref.weight.data.copy_(self.initial_atomref)

    def _status_word(self, device):
        return _lib.module_status(self, device, "atom type out of range for the embedding table (node_class=%d)"
                                  % self.embedding.num_embeddings)

    def check_status(self):
        """Synchronous form of the deferred index check (drains the stream)."""
        st = self.__dict__.get("_geossl_status")
        if st is not None:
            st.check()

    def _check_supported(self):
        """-> True when the fused kernels take this configuration (hidden_channels == num_filters in 32 / 64 / 128, at
        most 12 blocks and 64 gaussians: every configuration the reference's scripts use), False when it runs on the
        general-width path (_SchNetTapeCore)."""
        if self.dipole:
            raise NotImplementedError("dipole readout (schnet.py:103-107,117-118) is off the GeoSSL path")
        F = self.hidden_channels
        return (self.num_filters == F and F in SUPPORTED_F and self.num_interactions <= _lib.MAX_L
                and self.num_gaussians <= 64)

    def forward(self, z, pos, batch=None, return_latent=False, layout=None, latent_only=False):
        assert z.dim() == 1 and z.dtype == torch.long
        _lib.require_cuda(z, pos, batch)
        fused = self._check_supported()
        batch = torch.zeros_like(z) if batch is None else batch
        lay = layout if layout is not None else get_layout(batch)
        if lay.N != pos.size(0):
            raise ValueError("layout does not match the number of atoms")
        status = self._status_word(pos.device)
        status.poll()  # an out-of-range atom type seen by an earlier call raises here (IndexError, like Embedding)
        cfg = dict(L=self.num_interactions, F=self.hidden_channels, G=self.num_gaussians, cutoff=float(self.cutoff),
                   offset=self.distance_expansion.offset, coeff=float(self.distance_expansion.coeff),
                   debug=bool(_env("GEOSSL_DEBUG")), status=status,
                   # (GEOSSL_ARITH_24BIT: every dense product at fp32's own 24-bit product width - the atom-row layers then
                   # run as single launches of the three-bf16-piece row GEMM, the chain kernel is a two-fp16-piece kernel)
                   chain=(self.num_interactions >= 1 and not _env("GEOSSL_NO_CHAIN")
                          and not _env("GEOSSL_ARITH_24BIT")),
                   # The layer loop (chains and aggregations between the filter network and the heads) as ONE launch
                   # per pass (ops.layer_loop) - while a graph is being captured.  Launched eagerly it loses: the host
                   # has to describe all 14 operations before the GPU gets the first one, where separate launches
                   # pipeline (forward-only line 1.2 M against 1.4 M molecules/s).  GEOSSL_LAYER_LOOP=1 / =0 force it
                   # on / off (tests, A/B runs).
                   loop=(_env("GEOSSL_LAYER_LOOP") == "1" or
                         (_env("GEOSSL_LAYER_LOOP") is None and not _env("GEOSSL_NO_LAYER_LOOP")
                          and torch.cuda.is_current_stream_capturing())),
                   loop_stagger=int(_env("GEOSSL_LAYER_LOOP_STAGGER") or 0))
        if pos.dtype != torch.float32:
            raise TypeError("positions must be float32")
        if fused:
            h = _SchNetCore.apply(z, pos.contiguous(), lay, cfg, *_core_params(self))
        else:   # any other widths (schnet.py:17-30 takes any): the general path on the library's tape
            if z.numel() and (int(z.min()) < 0 or int(z.max()) >= self.embedding.num_embeddings):   # (Embedding's IndexError)
                raise IndexError("atom type out of range for the embedding table (node_class=%d)"
                                 % self.embedding.num_embeddings)
            h = _SchNetTapeCore.apply(z, pos.contiguous(), lay, cfg, *_core_params(self))
        status.arm()
        if not self.dipole and self.mean is not None and self.std is not None:
            h = h * self.std + self.mean
        if not self.dipole and self.atomref is not None:
            h = h + self.atomref(z)
        if latent_only and return_latent:  # (extension) a caller that discards the readout, e.g. do_DDM: (None, h)
            return None, h
        out = _SegmentReduce.apply(h, lay, self.readout)
        if self.scale is not None:
            out = self.scale * out
        if return_latent:
            return out, h
        return out

    def __repr__(self):
        return (f"{self.__class__.__name__}(hidden_channels={self.hidden_channels}, "
                f"num_filters={self.num_filters}, num_interactions={self.num_interactions}, "
                f"num_gaussians={self.num_gaussians}, cutoff={self.cutoff})")
