"""NCSN_version_03 — the denoising-distance-matching head — behind the reference's interface
(examples/NCSN.py:9-43,168-220), computed by the HIP path (geossl_ddm_loss_*).

``forward(data, node_feature, distance, debug=False)`` keeps the reference signature: ``data`` must
expose ``.batch``, ``.super_edge_index`` and ``.num_graphs`` (NCSN.py:186-190).  The two random
draws of the reference (``torch.randint`` :190, ``torch.randn_like`` :194) are made with the same
torch calls by default, or can be injected with ``noise_level=`` / ``distance_noise=`` (parity
tests, reproducible multi-GPU runs).
"""
import ctypes as C
import os

import numpy as np
import torch
from .switches import env as _env
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import call, ptr, stream
from .layout import get_super_edge_layout


class MultiLayerPerceptron(nn.Module):
    """Parameter holder mirroring NCSN.py:9-31 (``layers`` ModuleList, xavier weights, zero bias)."""

    def __init__(self, input_dim, hidden_dims, activation="relu", dropout=0):
        super().__init__()
        self.dims = [input_dim] + hidden_dims
        self.activation = getattr(F, activation) if isinstance(activation, str) else None
        self.dropout = nn.Dropout(dropout) if dropout else None
        self.layers = nn.ModuleList()
        for i in range(len(self.dims) - 1):
            self.layers.append(nn.Linear(self.dims[i], self.dims[i + 1]))
        self.reset_parameters()

    def reset_parameters(self):
        for layer in self.layers:
            nn.init.xavier_uniform_(layer.weight)
            nn.init.constant_(layer.bias, 0.0)

    def forward(self, input):
        """NCSN.py:33-43 for stand-alone use of the module (inside NCSN_version_03 both MLPs are fused into
        geossl_ddm_loss_fwd/bwd): every layer's product runs on the HIP row GEMM (differentiable to any order,
        geossl_amd/higher_order._linear_wide; widths padded to the kernel's multiples of 8 / 4 with zeros), activation and
        dropout are the torch functions the reference calls."""
        from .higher_order import _linear_wide
        _lib.require_cuda(input)
        lead = input.shape[:-1]
        x = input.reshape(-1, input.shape[-1])
        for i, layer in enumerate(self.layers):
            x = _linear_wide(x, layer.weight, layer.bias)
            if i < len(self.layers) - 1:
                if self.activation:
                    x = self.activation(x)
                if self.dropout:
                    x = self.dropout(x)
        return x.reshape(*lead, x.shape[-1])


def _head_params(m):
    i, o = m.input_distance_mlp.layers, m.output_mlp.layers
    return [i[0].weight, i[0].bias, i[1].weight, i[1].bias, o[0].weight, o[0].bias, o[1].weight, o[1].bias,
            o[2].weight, o[2].bias]


_FIELDS = ("in_w1", "in_b1", "in_w2", "in_b2", "o1_w", "o1_b", "o2_w", "o2_b", "o3_w", "o3_b")

# Optional side stream for the head's weight-gradient kernels in the two-pass form (GEOSSL_NCSN_SPLIT_BWD; the default
# one-pass backward has no separate weight-gradient kernels).  They depend only on the row pass of the head, while
# the backbone's backward (a chain of small, latency-bound launches) needs nothing but dh: with a side stream set
# (DDMTrainer does, and joins it before the optimiser step) the two run concurrently.  Off by default: a caller that
# runs its own optimiser right after loss.backward() must see every gradient on the current stream.
# Process-wide state (one process drives one GPU, parallel.py): set and cleared by DDMTrainer around loss.backward().
_SIDE = {"stream": None, "pending": []}


def set_side_stream(stream):
    _SIDE["stream"] = stream


def join_side_stream():
    """Make the current stream wait for the side-stream work and release the tensors kept alive for it."""
    if _SIDE["pending"]:
        for ev, _keep in _SIDE["pending"]:
            torch.cuda.current_stream().wait_event(ev)
        _SIDE["pending"].clear()


class _NcsnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, distance, noise_level, distance_noise, sel, sigmas, anneal_power, out_scale, *params):
        dev = h.device
        S, Fd = sel.S, h.size(1)
        training = ctx.needs_input_grad[0] or any(ctx.needs_input_grad[8:])
        ps = [p.detach().contiguous() for p in params]
        w = _lib.NcsnWeights()
        for name, p in zip(_FIELDS, ps):
            setattr(w, name, ptr(p))
        w.sigmas = ptr(sigmas)
        ctx.grad_slot = getattr(h, "_geossl_grad_slot", None)  # pretrain_GeoSSL.split_views: where dh may be written
        h = h.detach().contiguous()
        loss_e = torch.empty(S, dtype=torch.float32, device=dev)
        sv = None
        saved = {}
        if training:
            saved = dict(a1=torch.empty(S, Fd, dtype=torch.float32, device=dev),
                         a2=torch.empty(S, Fd // 2, dtype=torch.float32, device=dev),
                         pd=torch.empty(S, dtype=torch.float32, device=dev),
                         emb=torch.empty(S, dtype=torch.float32, device=dev),
                         gscale=torch.empty(S, dtype=torch.float32, device=dev))
            sv = _lib.NcsnSaved(*[ptr(saved[k]) for k in ("a1", "a2", "pd", "emb", "gscale")])
        lib = _lib.load()
        nws = max(int(lib.geossl_ddm_loss_fwd_workspace_floats(Fd)), 256)
        # (no super-edges: the row pass returns before it writes its block partials - the mean below then sums zeros)
        ws = (torch.empty if S > 0 else torch.zeros)(nws, dtype=torch.float32, device=dev)
        st = stream()
        call("geossl_ddm_loss_fwd", ptr(h), ptr(sel.batch), ptr(sel.sei0), ptr(sel.sei1), S,
             ptr(distance.contiguous()), ptr(noise_level.contiguous()), ptr(distance_noise.contiguous()), C.byref(w),
             Fd, float(anneal_power), ptr(loss_e), C.byref(sv) if sv is not None else None, ptr(ws), st)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        # the row pass left one partial sum per block in its workspace: the mean is one small launch (NCSN.py:210-212)
        call("geossl_loss_reduce_partials", ptr(ws), ptr(sel.stats), float(out_scale), ptr(loss), 0, st)
        if training:
            ctx.sel, ctx.ps, ctx.w, ctx.saved, ctx.h = sel, ps, w, saved, h
            ctx.params = params
            ctx.sigmas = sigmas
            ctx.out_scale = float(out_scale)
        return loss

    @staticmethod
    def backward(ctx, gout):
        sel, ps, w, saved, h = ctx.sel, ctx.ps, ctx.w, ctx.saved, ctx.h
        dev = h.device
        S, Fd, N = sel.S, h.size(1), h.size(0)
        st = stream()
        sv = _lib.NcsnSaved(*[ptr(saved[k]) for k in ("a1", "a2", "pd", "emb", "gscale")])
        dfeat = torch.empty(S, Fd, dtype=torch.float32, device=dev)
        demb = torch.empty(S, dtype=torch.float32, device=dev)
        grow = torch.empty(S, dtype=torch.float32, device=dev)
        gout = gout.contiguous().to(torch.float32)
        direct = _lib.direct_grads_enabled(ctx.params)  # opt-in (DDMTrainer); otherwise gradients go through autograd
        grads = [p.grad for p in ctx.params] if direct else [torch.empty_like(p) for p in ps]
        g = _lib.NcsnGrads(*[ptr(t) for t in grads])
        lib = _lib.load()
        acc = 1 if direct else 0
        # the two-pass form (row pass, then column GEMMs): on request, and for tensors of 4 GiB and more, which the
        # one-pass kernel's 32-bit byte offsets cannot address
        # (GEOSSL_ARITH_24BIT: the two-pass form is the one whose products are all 24-bit - row pass on three bf16 pieces,
        # weight gradients on the three-piece column GEMM)
        split = (_env("GEOSSL_NCSN_SPLIT_BWD") is not None or _env("GEOSSL_ARITH_24BIT") is not None
                 or max(S, N) * Fd * 4 >= 2 ** 32)
        if not split:
            # one pass over the rows: dfeat / demb / grow and every weight gradient of the head (ncsn_bwd.hip)
            ws1 = torch.empty(int(lib.geossl_ddm_loss_bwd_fused_workspace_floats(S, Fd)), dtype=torch.float32, device=dev)
            call("geossl_ddm_loss_bwd_fused", ptr(h), ptr(sel.sei0), ptr(sel.sei1), S, N, Fd, C.byref(w), C.byref(sv),
                 ptr(sel.stats), ctx.out_scale, ptr(gout), ptr(dfeat), ptr(demb), ptr(grow), C.byref(g), ptr(ws1), acc, st)
            dz1 = None
        else:
            dz1 = torch.empty(S, Fd, dtype=torch.float32, device=dev)
            call("geossl_ddm_loss_bwd_rows", C.byref(w), C.byref(sv), S, Fd, ptr(sel.stats), ctx.out_scale, ptr(gout),
                 ptr(dz1), ptr(dfeat), ptr(demb), ptr(grow), st)
        dh = None
        if ctx.needs_input_grad[0]:  # first: the backbone's backward waits for nothing else
            slot = ctx.grad_slot
            dh = slot.rows(N, Fd, torch.float32, dev) if slot is not None else None
            if dh is None:
                dh = torch.empty(N, Fd, dtype=torch.float32, device=dev)
            call("geossl_incidence_gather", ptr(dfeat), ptr(sel.inc_ptr), ptr(sel.inc_idx), N, Fd, ptr(dh), 0, st)
        if split:
            nfl = lib.geossl_ddm_loss_bwd_workspace_floats(S, Fd)
            side = _SIDE["stream"] if direct else None
            if side is None:
                ws = torch.empty(nfl, dtype=torch.float32, device=dev)
                call("geossl_ddm_loss_bwd_weights", ptr(h), ptr(sel.sei0), ptr(sel.sei1), S, Fd, C.byref(w), C.byref(sv),
                     ptr(dz1), ptr(demb), ptr(grow), C.byref(g), ptr(ws), acc, st)
            else:
                main = torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record(main)
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    ws = torch.empty(nfl, dtype=torch.float32, device=dev)
                    call("geossl_ddm_loss_bwd_weights", ptr(h), ptr(sel.sei0), ptr(sel.sei1), S, Fd, C.byref(w),
                         C.byref(sv), ptr(dz1), ptr(demb), ptr(grow), C.byref(g), ptr(ws), 1, stream())
                    done = torch.cuda.Event()
                    done.record(side)
                _SIDE["pending"].append((done, (h, saved, dz1, demb, grow, ws, w, sv, ps, grads)))
        if direct:
            return (dh, None, None, None, None, None, None, None) + (None,) * len(grads)
        return (dh, None, None, None, None, None, None, None) + tuple(grads)


class _NcsnLossPair(torch.autograd.Function):
    """Both heads of a DDM step (pretrain_GeoSSL.py:207-210) as ONE autograd node: loss = scale * (l1 + l2), the row
    passes, the one-pass backwards and the incidence gathers of the two heads in the same launches
    (geossl_ddm_loss_fwd2 / _bwd_fused2: 4 launches per step where two _NcsnLoss nodes make 12).  Same arithmetic per
    head; at the reference's batch size a head's row pass fills a quarter of the chip, the pair half of it."""

    @staticmethod
    def forward(ctx, h1, h2, d1, d2, nl1, dn1, nl2, dn2, sel, sigmas, powers, out_scale, npar, *params):
        dev = h1.device
        S, Fd, N = sel.S, h1.size(1), h1.size(0)
        training = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or any(ctx.needs_input_grad[13:])
        ps = [p.detach().contiguous() for p in params]
        pss = (ps[:npar], ps[npar:])
        # capacity bucket (geossl_amd/bucket.py): h1 is the whole [view 0 ; view 1 ; unused] feature tensor and h2 is
        # None - both heads get its base address, head 1's rows start at the real atom count the kernels read from the
        # device (dyn_view); S is the capacity of the super-edge buffers, the real count is read the same way
        dyn = getattr(sel, "dyn", None)
        ctx.dyn = dyn
        if dyn is not None:
            hs = (h1.detach().contiguous(),) * 2
            ctx.grad_slots = (None, None)
        else:
            hs = (h1.detach().contiguous(), h2.detach().contiguous())
            ctx.grad_slots = (getattr(h1, "_geossl_grad_slot", None), getattr(h2, "_geossl_grad_slot", None))
        heads = (_lib.NcsnHeadFwd * 2)()
        keep, saved, ws = [], [], []
        for k, (h, d, nl, dn) in enumerate(((hs[0], d1, nl1, dn1), (hs[1], d2, nl2, dn2))):
            hd = heads[k]
            d, nl, dn = d.contiguous(), nl.contiguous(), dn.contiguous()
            keep += [d, nl, dn]
            for name, p in zip(_FIELDS, pss[k]):
                setattr(hd.w, name, ptr(p))
            hd.w.sigmas = ptr(sigmas[k])
            hd.h, hd.distance, hd.noise_level, hd.distance_noise = ptr(h), ptr(d), ptr(nl), ptr(dn)
            hd.anneal_power = float(powers[k])
            loss_e = torch.empty(S, dtype=torch.float32, device=dev)
            sv = {}
            if training:
                sv = dict(a1=torch.empty(S, Fd, dtype=torch.float32, device=dev),
                          a2=torch.empty(S, Fd // 2, dtype=torch.float32, device=dev),
                          pd=torch.empty(S, dtype=torch.float32, device=dev),
                          emb=torch.empty(S, dtype=torch.float32, device=dev),
                          gscale=torch.empty(S, dtype=torch.float32, device=dev))
                for name in ("a1", "a2", "pd", "emb", "gscale"):
                    setattr(hd.saved, name, ptr(sv[name]))
            w_ = torch.empty(256, dtype=torch.float32, device=dev)
            hd.loss_e, hd.workspace = ptr(loss_e), ptr(w_)
            keep.append(loss_e)
            saved.append(sv)
            ws.append(w_)
        st = stream()
        call("geossl_ddm_loss_fwd2_dyn", C.byref(heads), ptr(sel.batch), ptr(sel.sei0), ptr(sel.sei1), S, Fd,
             None if dyn is None else dyn.n_super, None if dyn is None else dyn.n_atoms, st)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        call("geossl_loss_reduce_partials2", ptr(ws[0]), ptr(ws[1]), ptr(sel.stats), float(out_scale), float(out_scale),
             ptr(loss), st)
        if training:
            ctx.sel, ctx.pss, ctx.saved, ctx.hs, ctx.sigmas = sel, pss, saved, hs, sigmas
            ctx.params, ctx.npar, ctx.out_scale = params, npar, float(out_scale)
        return loss

    @staticmethod
    def backward(ctx, gout):
        sel, pss, saved, hs = ctx.sel, ctx.pss, ctx.saved, ctx.hs
        dev = hs[0].device
        S, Fd, N = sel.S, hs[0].size(1), hs[0].size(0)
        st = stream()
        gout = gout.contiguous().to(torch.float32)
        direct = _lib.direct_grads_enabled(ctx.params)  # opt-in (DDMTrainer); otherwise gradients go through autograd
        grads = [p.grad for p in ctx.params] if direct else [torch.empty_like(p) for p in pss[0] + pss[1]]
        gs = (grads[:ctx.npar], grads[ctx.npar:])
        lib = _lib.load()
        heads = (_lib.NcsnHeadBwd * 2)()
        keep, dhs = [], []
        nws = int(lib.geossl_ddm_loss_bwd_fused_workspace_floats(S, Fd))
        dyn = ctx.dyn
        dh_all = torch.empty(N, Fd, dtype=torch.float32, device=dev) if dyn is not None else None  # both views' rows
        for k in range(2):
            hd = heads[k]
            for name, p in zip(_FIELDS, pss[k]):
                setattr(hd.w, name, ptr(p))
            hd.w.sigmas = ptr(ctx.sigmas[k])
            for name in ("a1", "a2", "pd", "emb", "gscale"):
                setattr(hd.saved, name, ptr(saved[k][name]))
            for name, t_ in zip(_FIELDS, gs[k]):
                setattr(hd.grads, name, ptr(t_))
            dfeat = torch.empty(S, Fd, dtype=torch.float32, device=dev)
            demb = torch.empty(S, dtype=torch.float32, device=dev)
            grow = torch.empty(S, dtype=torch.float32, device=dev)
            ws1 = torch.empty(nws, dtype=torch.float32, device=dev)
            slot = ctx.grad_slots[k]
            dh = dh_all if dyn is not None else (slot.rows(N, Fd, torch.float32, dev) if slot is not None else None)
            if dh is None:
                dh = torch.empty(N, Fd, dtype=torch.float32, device=dev)
            hd.h, hd.out_scale, hd.dfeat, hd.demb, hd.grow, hd.workspace, hd.dh = (
                ptr(hs[k]), ctx.out_scale, ptr(dfeat), ptr(demb), ptr(grow), ptr(ws1), ptr(dh))
            keep += [dfeat, demb, grow, ws1]
            dhs.append(dh)
        # (bucket: N = rows of the fused tensor bounds the byte offsets; the gather walks the real atoms of a view)
        call("geossl_ddm_loss_bwd_fused2_dyn", C.byref(heads), ptr(sel.sei0), ptr(sel.sei1), S, sel.N if dyn is not None else N,
             Fd, ptr(sel.stats), ptr(gout), ptr(sel.inc_ptr), ptr(sel.inc_idx), 1 if direct else 0,
             None if dyn is None else dyn.n_super, None if dyn is None else dyn.n_atoms, st)
        head = ((dh_all, None) if dyn is not None else (dhs[0], dhs[1])) + (None,) * 11
        if direct:
            return head + (None,) * len(grads)
        return head + tuple(grads)


def ddm_heads_loss(n1, n2, data, h1, distance_1, h2, distance_2, noise_level_1=None, distance_noise_1=None,
                   noise_level_2=None, distance_noise_2=None, out_scale=0.5):
    """out_scale * (NCSN_model_01(data, h1, distance_1) + NCSN_model_02(data, h2, distance_2)) - the loss of
    pretrain_GeoSSL.py:207-210 - with both heads in the same launches (_NcsnLossPair) when they allow it, else as two
    calls.  The random draws are the ones the two forward() calls make, in their order."""
    ok = (isinstance(n1, NCSN_version_03) and isinstance(n2, NCSN_version_03) and n1.emb_dim == n2.emb_dim
          and n1.emb_dim in (32, 64, 128) and not _env("GEOSSL_NCSN_SPLIT_BWD")
          and not _env("GEOSSL_ARITH_24BIT")
          and not _env("GEOSSL_NCSN_SEPARATE_HEADS")
          and (h2 is None or h1.shape == h2.shape) and not distance_1.requires_grad and not distance_2.requires_grad
          # the paired kernels write the two heads' gradients from different blocks of ONE launch: the heads must not
          # share a parameter (the same module passed twice, tied weights) - such a pair takes the two single-head calls
          and n1 is not n2 and not ({id(p) for p in _head_params(n1)} & {id(p) for p in _head_params(n2)}))
    if ok:
        S_, N_, Fd = data.super_edge_index.size(1), h1.size(0), h1.size(1)
        ok = S_ > 0 and max(S_, N_) * Fd * 4 < 2 ** 32
    bucket = getattr(data, "_bucket", None)
    if bucket is not None and (not ok or h2 is not None):
        raise _lib.GeosslHipError("a capacity-bucket batch needs the paired NCSN heads on one fused feature tensor")
    if not ok:
        return (n1(data, h1, distance_1, noise_level=noise_level_1, distance_noise=distance_noise_1, out_scale=out_scale) +
                n2(data, h2, distance_2, noise_level=noise_level_2, distance_noise=distance_noise_2, out_scale=out_scale))
    _lib.require_cuda(h1, h2, distance_1, distance_2, n1.sigmas, n2.sigmas)
    n1.device = n2.device = n1.sigmas.device
    num_graphs = data.num_graphs
    sel = bucket.sel if bucket is not None else get_super_edge_layout(data.batch, data.super_edge_index, num_graphs)
    draws = []
    for n_, nl, dn, d in ((n1, noise_level_1, distance_noise_1, distance_1), (n2, noise_level_2, distance_noise_2, distance_2)):
        if nl is None:  # NCSN.py:190
            nl = torch.randint(0, n_.sigmas.size(0), (num_graphs,), device=n_.sigmas.device)
        if dn is None:  # NCSN.py:194
            dn = torch.randn_like(d)
        draws.append((nl, dn))
    p1, p2 = _head_params(n1), _head_params(n2)
    return _NcsnLossPair.apply(h1, h2, distance_1.view(-1), distance_2.view(-1), draws[0][0], draws[0][1].view(-1),
                               draws[1][0], draws[1][1].view(-1), sel, (n1.sigmas.detach(), n2.sigmas.detach()),
                               (n1.anneal_power, n2.anneal_power), out_scale, len(p1), *p1, *p2)


class NCSN_version_03(torch.nn.Module):
    def __init__(self, emb_dim, sigma_begin, sigma_end, num_noise_level, noise_type, anneal_power):
        super().__init__()
        self.anneal_power = anneal_power
        self.noise_type = noise_type
        self.input_distance_mlp = MultiLayerPerceptron(1, [emb_dim, 1], activation="relu")
        self.output_mlp = MultiLayerPerceptron(1 + emb_dim, [emb_dim, emb_dim // 2, 1])
        sigmas = torch.tensor(np.exp(np.linspace(np.log(sigma_begin), np.log(sigma_end), num_noise_level)),
                              dtype=torch.float32)
        self.sigmas = nn.Parameter(sigmas, requires_grad=False)  # (num_noise_level)
        self.emb_dim = emb_dim

    def forward(self, data, node_feature, distance, debug=False, noise_level=None, distance_noise=None,
                out_scale=1.0):
        self.device = self.sigmas.device
        _lib.require_cuda(node_feature, distance, self.sigmas)
        if self.emb_dim not in (32, 64, 128):
            raise NotImplementedError("HIP path supports emb_dim in (32, 64, 128)")
        if distance.requires_grad:
            raise NotImplementedError("gradient w.r.t. distance/positions is not built (SURVEY.md §8(f) N3)")
        num_graphs = data.num_graphs
        sel = get_super_edge_layout(data.batch, data.super_edge_index, num_graphs)
        if noise_level is None:  # NCSN.py:190
            noise_level = torch.randint(0, self.sigmas.size(0), (num_graphs,), device=self.device)
        if distance_noise is None:  # NCSN.py:194
            distance_noise = torch.randn_like(distance)
        if sel.S == 0:
            raise ValueError("batch has no super edges (scatter_add over an empty index fails in the reference too)")
        loss = _NcsnLoss.apply(node_feature, distance.view(-1), noise_level, distance_noise.view(-1), sel,
                               self.sigmas.detach(), self.anneal_power, out_scale, *_head_params(self))
        if debug:
            print("loss", float(loss))
        return loss
