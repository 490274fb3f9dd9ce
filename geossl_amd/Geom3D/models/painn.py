"""PaiNN behind the reference's interface (Geom3D/models/painn.py:117-269) — BASELINE config 5.

Same constructor, ``forward(x, positions, radius_edge_index, batch, return_latent=False)`` signature,
parameter registration order and state_dict keys as the reference.  The arithmetic runs on the HIP path:
edge geometry / radial basis / cutoff, the filter-weighted message sum per target atom (filters recomputed per
edge from the 20 radial values, never stored), the element-wise parts of the mixing block, and the Dense layers
on the MFMA row/column GEMMs — forward and backward, as one autograd node.  CUDA tensors only.
"""
import ctypes as C
import os
from typing import Callable, Optional

import torch
from ...switches import env as _env
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import xavier_uniform_, zeros_

from ... import _lib, ops
from ..._lib import call, ptr, stream
from ...layout import get_edge_layout, get_layout


class Dense(nn.Linear):
    """painn_utils.py:9-35: Linear with xavier weight / zero bias init and an activation attribute."""

    def __init__(self, in_features, out_features, bias=True, activation=None, weight_init=xavier_uniform_,
                 bias_init=zeros_):
        self.weight_init = weight_init
        self.bias_init = bias_init
        super().__init__(in_features, out_features, bias)
        self.activation = activation if activation is not None else nn.Identity()

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)

    def forward(self, input):
        """painn_utils.py:31-35 for stand-alone use (the energy head of create_output_layers(); inside PaiNN the Dense
        layers run in the fused node): the product on the HIP row GEMM, differentiable to any order."""
        from ...higher_order import _linear_wide, _silu
        _lib.require_cuda(input)
        lead = input.shape[:-1]
        y = _linear_wide(input.reshape(-1, input.shape[-1]), self.weight, self.bias)  # bias in the GEMM's epilogue
        act = _silu if self.activation is F.silu else self.activation       # (F.silu: geossl_silu_fwd / _bwd)
        return act(y).reshape(*lead, self.out_features)


class GaussianRBF(nn.Module):
    """painn_utils.py:106-136 (non-trainable): offsets = linspace(start, cutoff, n_rbf), widths = |Δ|."""

    def __init__(self, n_rbf, cutoff, start=0.0):
        super().__init__()
        self.n_rbf = n_rbf
        offset = torch.linspace(start, cutoff, n_rbf)
        widths = torch.FloatTensor(torch.abs(offset[1] - offset[0]) * torch.ones_like(offset))
        self.register_buffer("widths", widths)
        self.register_buffer("offsets", offset)


class CosineCutoff(nn.Module):
    """painn_utils.py:158-177: buffer only; the cutoff function lives in the kernels."""

    def __init__(self, cutoff):
        super().__init__()
        self.register_buffer("cutoff", torch.FloatTensor([cutoff]))


class PaiNNInteraction(nn.Module):
    def __init__(self, n_atom_basis, activation):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.interatomic_context_net = nn.Sequential(
            Dense(n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None))


class PaiNNMixing(nn.Module):
    def __init__(self, n_atom_basis, activation, epsilon=1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.intraatomic_context_net = nn.Sequential(
            Dense(2 * n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None))
        self.mu_channel_mix = Dense(n_atom_basis, 2 * n_atom_basis, activation=None, bias=False)
        self.epsilon = epsilon


def _replicate(factory, n, share):
    if share:
        return nn.ModuleList([factory()] * n)
    return nn.ModuleList([factory() for _ in range(n)])


class PaiNN(nn.Module):
    def __init__(self, n_atom_basis: int, n_interactions: int, n_rbf: int, cutoff: float, n_out: int, readout: str,
                 n_out_hidden: int = None, n_out_layers: int = 2, activation: Optional[Callable] = F.silu,
                 max_z: int = 100, shared_interactions: bool = False, shared_filters: bool = False,
                 epsilon: float = 1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.n_interactions = n_interactions
        self.n_out = n_out
        self.n_out_hidden = n_out_hidden
        self.n_out_layers = n_out_layers
        self.activation = activation
        self.cutoff = cutoff
        self.cutoff_fn = CosineCutoff(cutoff)
        self.radial_basis = GaussianRBF(n_rbf=n_rbf, cutoff=cutoff)
        self.readout = readout
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.share_filters = shared_filters
        if shared_filters:
            self.filter_net = Dense(self.radial_basis.n_rbf, 3 * n_atom_basis, activation=None)
        else:
            self.filter_net = Dense(self.radial_basis.n_rbf, self.n_interactions * n_atom_basis * 3, activation=None)
        self.interactions = _replicate(lambda: PaiNNInteraction(self.n_atom_basis, activation), self.n_interactions,
                                       shared_interactions)
        self.mixing = _replicate(lambda: PaiNNMixing(self.n_atom_basis, activation, epsilon), self.n_interactions,
                                 shared_interactions)

    def create_output_layers(self):
        """build_mlp(n_in, n_out, n_hidden, n_layers, activation), painn_utils.py:38-70."""
        n_in, n_out, n_layers = self.n_atom_basis, self.n_out, self.n_out_layers
        if self.n_out_hidden is None:
            c, neurons = n_in, []
            for _ in range(n_layers):
                neurons.append(c)
                c = max(n_out, c // 2)
            neurons.append(n_out)
        else:
            hid = [self.n_out_hidden] * (n_layers - 1) if isinstance(self.n_out_hidden, int) else list(self.n_out_hidden)
            neurons = [n_in] + hid + [n_out]
        layers = [Dense(neurons[i], neurons[i + 1], activation=self.activation) for i in range(n_layers - 1)]
        layers.append(Dense(neurons[-2], neurons[-1], activation=None))
        return nn.Sequential(*layers)

    def check_status(self):
        """Synchronous form of the deferred index check (drains the stream)."""
        st = self.__dict__.get("_geossl_status")
        if st is not None:
            st.check()

    def _params(self):
        ps = [self.embedding.weight, self.filter_net.weight, self.filter_net.bias]
        for blk in self.interactions:
            n = blk.interatomic_context_net
            ps += [n[0].weight, n[0].bias, n[1].weight, n[1].bias]
        for blk in self.mixing:
            n = blk.intraatomic_context_net
            ps += [n[0].weight, n[0].bias, n[1].weight, n[1].bias, blk.mu_channel_mix.weight]
        return ps

    def forward(self, x, positions, radius_edge_index, batch, return_latent=False, latent_only=False, layout=None,
                edge_layout=None):
        """painn.py:216-269.  `latent_only` (with return_latent): the readout is not evaluated and None is returned in
        its place - the DDM step drops it (pretrain_GeoSSL.py:187).  `layout` / `edge_layout` (extension): the molecule
        and edge structures of the batch when the caller already has them - the static, capacity-sized structures of a
        capacity bucket (geossl_amd/bucket.py), whose real counts are device data."""
        _lib.require_cuda(x, positions, radius_edge_index, batch)
        if self.activation is not F.silu:
            raise NotImplementedError("HIP path implements the reference default activation F.silu")
        # The fused kernels take the configurations the reference's scripts use (n_atom_basis 32 / 64 / 128, 8 / 16 / 20 /
        # 32 radial functions, a filter and a block per interaction); everything else the constructor accepts
        # (painn.py:125-142: any width, shared_filters, shared_interactions) runs on the general path (_PaiNNTapeCore).
        shared = self.share_filters or (self.n_interactions > 1 and self.interactions[0] is self.interactions[1])
        fused = (not shared and self.n_atom_basis in (32, 64, 128) and self.radial_basis.n_rbf in (8, 16, 20, 32))
        atomic_numbers = x[:, 0] if x.dim() == 2 else x  # painn.py:226-229
        lay = layout if layout is not None else get_layout(batch)
        if positions.dtype != torch.float32:
            raise TypeError("positions must be float32")
        el = edge_layout if edge_layout is not None else get_edge_layout(batch, radius_edge_index, lay.B)
        status = _lib.module_status(self, positions.device, "atomic number out of range for the embedding table "
                                    "(max_z=%d)" % self.embedding.num_embeddings)
        status.poll()  # an out-of-range atomic number seen by an earlier call raises here (IndexError, like Embedding)
        cfg = dict(F=self.n_atom_basis, L=self.n_interactions, R=self.radial_basis.n_rbf, cutoff=float(self.cutoff),
                   offsets=self.radial_basis.offsets, widths=self.radial_basis.widths,
                   eps=float(self.mixing[0].epsilon), status=status, debug=bool(_env("GEOSSL_DEBUG")),
                   lay=lay, mma=(self.n_atom_basis == 128 and self.radial_basis.n_rbf in (8, 16, 20)
                                 and not _env("GEOSSL_PAINN_VECTOR")))
        if fused:
            q = _PaiNNCore.apply(atomic_numbers, positions.contiguous(), el, cfg, *self._params())
        else:
            if atomic_numbers.numel() and (int(atomic_numbers.min()) < 0
                                           or int(atomic_numbers.max()) >= self.embedding.num_embeddings):
                raise IndexError("atomic number out of range for the embedding table (max_z=%d)"
                                 % self.embedding.num_embeddings)
            cfg["share_filters"] = bool(self.share_filters)
            q = _PaiNNTapeCore.apply(atomic_numbers, positions.contiguous(), el, cfg, *self._params())
        status.arm()
        if return_latent and latent_only:
            return None, q
        from .schnet import _SegmentReduce
        h = _SegmentReduce.apply(q, lay, self.readout)  # painn.py:266
        if return_latent:
            return h, q
        return h


def _split3(t, F_):
    return [t[:, c * F_:(c + 1) * F_] for c in range(3)]


class _PaiNNTapeCore(torch.autograd.Function):
    """painn.py:230-255 for the configurations the fused kernels do not take (any n_atom_basis / n_rbf, shared_filters,
    shared_interactions: painn.py:125-142,178-202,242-243): the backbone restated on the library's reverse-mode tape
    (geossl_amd/tape.py: every primitive a HIP kernel), differentiated by the tape.  A correct path for every
    configuration, not a fast one; first order only."""

    @staticmethod
    def forward(ctx, z, pos, el, cfg, *params):
        from ... import tape as tp
        with torch.no_grad():
            x = tp.leaf(pos, bool(ctx.needs_input_grad[1]))
            ps = [tp.leaf(p, bool(n)) for p, n in zip(params, ctx.needs_input_grad[4:])]
            q = tp.painn_atom_features(z.contiguous(), x, el.idx_i, el.idx_j, cfg, ps, getattr(el, "inc", None))
        ctx.tape = (q, x, ps)
        ctx.shapes = [tuple(p.shape) for p in params]
        return q.t

    @staticmethod
    def backward(ctx, dq):
        from ... import tape as tp
        if torch.is_grad_enabled():
            raise NotImplementedError("second-order gradients (training on forces) need a configuration of the fused path")
        q, x, ps = ctx.tape
        wrt = [v for v in [x] + ps if v.req]
        with torch.no_grad():
            got = iter(tp.grad([q], [tp.const(dq.contiguous())], wrt))
        vals = [next(got) if v.req else None for v in [x] + ps]
        dpos = None if vals[0] is None else vals[0].t
        grads = [None if g is None else g.t.reshape(shape) for g, shape in zip(vals[1:], ctx.shapes)]
        return (None, dpos, None, None) + tuple(grads)


class _PaiNNCore(torch.autograd.Function):
    """(atomic numbers, positions, radius_edge_index) -> scalar atom features q (painn.py:230-255)."""

    @staticmethod
    def forward(ctx, z, pos, el, cfg, *params):
        F_, L, R = cfg["F"], cfg["L"], cfg["R"]
        dev, N, E = pos.device, pos.size(0), el.E
        st = stream()
        want_params = any(ctx.needs_input_grad[4:])
        training = want_params or ctx.needs_input_grad[1]  # positions: forces (finetune_md17.py:46)
        ps = [p.detach().contiguous() for p in params]
        emb_w, fw, fb = ps[0], ps[1], ps[2]
        inter = [ps[3 + 4 * l: 7 + 4 * l] for l in range(L)]
        mix = [ps[3 + 4 * L + 5 * l: 8 + 4 * L + 5 * l] for l in range(L)]
        f32 = dict(dtype=torch.float32, device=dev)
        # the structures of a capacity bucket (geossl_amd/bucket.py) carry the device addresses of the batch's real counts:
        # N, E are then capacities and every count-driven launch below is told where the real count lives (rows of the
        # scalar features, rows of the vector features viewed as [3 N, F], edges)
        dyn = getattr(el, "dyn", None)
        dN, dN3, dE = (dyn.n_atoms2, dyn.n_atoms2x3, dyn.n_edges2) if dyn is not None else (None, None, None)
        if dyn is not None and (ctx.needs_input_grad[1] or F_ != 128 or _env("GEOSSL_PAINN_NO_CHAIN")
                                or _env("GEOSSL_PAINN_SILU_KERNELS")):
            raise _lib.GeosslHipError("a capacity-bucket layout serves the F = 128 chain path without position gradients")
        dirv, fcut, phi = torch.empty(max(E, 1), 3, **f32), torch.empty(max(E, 1), **f32), torch.empty(max(E, 1), R, **f32)
        call("geossl_painn_edge_geom_dyn", ptr(pos), ptr(el.idx_i), ptr(el.idx_j), E, cfg["cutoff"], ptr(cfg["offsets"]),
             ptr(cfg["widths"]), R, ptr(dirv), ptr(fcut), ptr(phi), dE, st)
        q = torch.empty(N, F_, **f32)
        call("geossl_embedding_fwd_dyn", ptr(z), z.stride(0) if z.numel() else 1, ptr(emb_w), emb_w.size(0), N, F_, ptr(q),
             ptr(cfg["status"].word), dN, st)                               # painn.py:247 (row 0 is the zero padding row)
        if cfg["debug"]:
            cfg["status"].check()
        inc_ptr, inc_idx = el.inc["i"]
        saved = []
        # Every Dense layer here is a set of F x F row GEMMs (3F outputs = three, a 2F contraction = two with the
        # residual operand): they run as one-stage launches of the chained row kernel (weights as fragments in
        # registers, no per-launch weight formatting), on operand images built by one launch per pass.
        blocks = []
        for l in range(L):
            c0w, c0b, c1w, c1b = inter[l]
            i0w, i0b, i1w, i1b, mw = mix[l]
            blocks += [c0w] + [c1w[c * F_:(c + 1) * F_] for c in range(3)] + [mw[:F_], mw[F_:]] + \
                      [i0w[:, :F_], i0w[:, F_:]] + [i1w[c * F_:(c + 1) * F_] for c in range(3)]
        img = ops.prepare_chain(blocks, transB=True) if not _env("GEOSSL_PAINN_NO_CHAIN") else None
        NB = 11  # blocks per layer, in the order above
        # silu inside the chained launches (F = 128): GEOSSL_PAINN_SILU_KERNELS keeps the separate silu launches (A/B runs)
        fused = img is not None and F_ == 128 and not _env("GEOSSL_PAINN_SILU_KERNELS")

        def lin_fan(x, ks, biases, outs):
            """several F x F blocks of one wide Dense applied to the same rows: one launch (F = 128), else one each"""
            if img is None or F_ != 128:
                for k, b_, o_ in zip(ks, biases, outs):
                    lin(x, None, k, bias=b_, out=o_)
                return
            ops.linear_chain(x, [dict(image=img[k], bias=b_, out=o_, same_input=(n_ > 0))
                                 for n_, (k, b_, o_) in enumerate(zip(ks, biases, outs))],
                             dyn_rows=dN if x.size(0) == N else dN3)

        def lin(x, w, k, bias=None, res=None, out=None):
            if img is None:
                return ops.linear(x, blocks[k].contiguous(), bias=bias, res=res, out=out)
            return ops.linear_chain(x, [dict(image=img[k], bias=bias, res=res, out=out)],
                                    dyn_rows=dN if x.size(0) == N else dN3)[0]

        # Which interaction kernel: the matrix-pipe form stages a molecule's rows in LDS (at most `cap_f` atoms).  A
        # layout with larger molecules (Molecule3D with hydrogens) keeps that form for the molecules that fit and covers
        # the atoms of the others with the per-atom kernel (`big_f`: their list) - one oversized molecule does not send
        # the whole batch to the slower kernels.  Without host-side molecule sizes there is no list: one form for all.
        lay = cfg["lay"]
        lib = _lib.load()
        use_mma = bool(cfg["mma"]) and E > 0
        big_f, stage_f = None, lay.max_n      # the atoms left to the per-atom kernel; the molecule size the LDS rows are for
        if use_mma:
            from ...layout import painn_stage_caps
            hard, split_cap = int(lib.geossl_painn_stage_cap(0, F_, R)), painn_stage_caps(F_, R)[0]
            if 0 < split_cap < lay.max_n:
                big_f = lay.big_atoms(split_cap)
                if big_f is not None:
                    stage_f = split_cap
            if big_f is None and lay.max_n > hard:
                use_mma = False
        # mu starts as zeros (:249).  When the matrix-pipe kernel covers every atom, the first interaction is told so (NULL)
        # instead of being handed 3 N F zeros to stage and gather: its mu rows, the dmumu * mu_j products and - in the
        # backward - the whole dmumu third of the filter are exact zeros (csrc: k_painn_fwd_mma<R, true>,
        # k_painn_interaction_bwd_mol<R, true>).  Forces (edge gradients read mu) and GEOSSL_PAINN_NO_MU_ZERO keep the tensor.
        mu_is_zero = (use_mma and big_f is None and not ctx.needs_input_grad[1] and R in (8, 16, 20)
                      and not _env("GEOSSL_PAINN_NO_MU_ZERO"))
        mu = None if mu_is_zero else torch.zeros(N, 3, F_, **f32)            # :249
        for l in range(L):
            c0w, c0b, c1w, c1b = inter[l]
            k0 = NB * l
            xc = torch.empty(N, 3 * F_, **f32)
            if fused:  # Dense(F, F, silu) and Dense(F, 3F) (:27-30,53) in one launch: u and silu(u) are both kept
                u, s = torch.empty(N, F_, **f32), torch.empty(N, F_, **f32)
                ops.linear_chain(q, [dict(image=img[k0], bias=c0b, out=u, out_act=s, flags=_lib.EPI_SILU)] +
                                 [dict(image=img[k0 + 1 + c], bias=c1b[c * F_:(c + 1) * F_], out=o_, same_input=(c > 0))
                                  for c, o_ in enumerate(_split3(xc, F_))], dyn_rows=dN)
            else:
                u = lin(q, c0w, k0, bias=c0b)                                # Dense(F, F, silu)      :27-30,53
                s = torch.empty_like(u)
                call("geossl_silu_fwd", ptr(u), u.numel(), ptr(s), st)
                lin_fan(s, [k0 + 1 + c for c in range(3)], [c1b[c * F_:(c + 1) * F_] for c in range(3)],
                        _split3(xc, F_))                                     # Dense(F, 3F)
            q2, mu2 = torch.empty_like(q), torch.empty(N, 3, F_, **f32)
            # one block per molecule: the rows its edges read are staged in LDS once
            if use_mma:  # filter on the matrix pipe (painn_mma.hip); LDS: 3.5 KB per atom + 3 KB
                row_edge, grp_atom, _, mol_grp = el.groups("i", lay.mol_ptr)
                call("geossl_painn_interaction_fwd_mma_dyn", ptr(q), ptr(mu), ptr(xc), ptr(el.idx_j), ptr(row_edge),
                     ptr(grp_atom), ptr(mol_grp), ptr(phi), ptr(fcut), ptr(dirv), ptr(fw[l * 3 * F_:(l + 1) * 3 * F_]),
                     ptr(fb[l * 3 * F_:(l + 1) * 3 * F_]), ptr(lay.mol_ptr), lay.B, stage_f, N, F_, R, ptr(q2),
                     ptr(mu2), ptr(getattr(el, "mol_grp_end", None)), st)                 # :54-64
                if big_f is not None and big_f[1] > 0:   # the atoms of the molecules above the staged rows
                    call("geossl_painn_interaction_fwd_atoms", ptr(q), ptr(mu), ptr(xc), ptr(el.idx_j), ptr(inc_ptr),
                         ptr(inc_idx), ptr(phi), ptr(fcut), ptr(dirv), ptr(fw[l * 3 * F_:(l + 1) * 3 * F_]),
                         ptr(fb[l * 3 * F_:(l + 1) * 3 * F_]), ptr(big_f[0]), big_f[1], big_f[2], F_, R, ptr(q2), ptr(mu2), st)
            else:
                call("geossl_painn_interaction_fwd_mol", ptr(q), ptr(mu), ptr(xc), ptr(el.idx_j), ptr(inc_ptr),
                     ptr(inc_idx), ptr(phi), ptr(fcut), ptr(dirv), ptr(fw[l * 3 * F_:(l + 1) * 3 * F_]),
                     ptr(fb[l * 3 * F_:(l + 1) * 3 * F_]), ptr(lay.mol_ptr), lay.B, lay.max_n, N, F_, R, ptr(q2), ptr(mu2),
                     st)                                                        # :54-64
            i0w, i0b, i1w, i1b, mw = mix[l]
            mm = torch.empty(3 * N, 2 * F_, **f32)                           # mu_channel_mix        :100
            lin_fan(mu2.view(3 * N, F_), [k0 + 4, k0 + 5], [None, None], [mm[:, :F_], mm[:, F_:]])
            cx, dot = torch.empty(N, 2 * F_, **f32), torch.empty(N, F_, **f32)
            call("geossl_painn_mix_pre_fwd_dyn", ptr(q2), ptr(mm), N, F_, cfg["eps"], ptr(cx), ptr(dot), dN, st)  # :101-104
            # Dense(2F, F, silu) :105 - a contraction over 2F columns is two passes of the F-wide row GEMM (the split
            # kernel holds one K <= 128 weight image in LDS): the second adds onto the first through the residual operand
            xx = torch.empty(N, 3 * F_, **f32)
            if fused:  # both F-wide passes of Dense(2F, F, silu), then Dense(F, 3F) on silu of it: one launch
                u1, s1 = torch.empty(N, F_, **f32), torch.empty(N, F_, **f32)
                ops.linear_chain(cx[:, :F_], [dict(image=img[k0 + 6], bias=i0b, store=False),
                                              dict(image=img[k0 + 7], x=cx[:, F_:], add_prev=True, out=u1, out_act=s1,
                                                   flags=_lib.EPI_SILU)] +
                                 [dict(image=img[k0 + 8 + c], bias=i1b[c * F_:(c + 1) * F_], out=o_, same_input=(c > 0))
                                  for c, o_ in enumerate(_split3(xx, F_))], dyn_rows=dN)
            else:
                if img is not None and F_ == 128:   # both F-wide passes in one launch (the second brings its own input)
                    u1 = ops.linear_chain(cx[:, :F_], [dict(image=img[k0 + 6], bias=i0b, store=False),
                                                       dict(image=img[k0 + 7], x=cx[:, F_:], add_prev=True)])[1]
                else:
                    u1 = lin(cx[:, :F_], None, k0 + 6, bias=i0b)
                    lin(cx[:, F_:], None, k0 + 7, res=u1, out=u1)
                s1 = torch.empty_like(u1)
                call("geossl_silu_fwd", ptr(u1), u1.numel(), ptr(s1), st)
                lin_fan(s1, [k0 + 8 + c for c in range(3)], [i1b[c * F_:(c + 1) * F_] for c in range(3)],
                        _split3(xx, F_))                                     # Dense(F, 3F)
            # (the representation is q alone, :262-269: the LAST block's mu' is nobody's input and is not formed - NULL)
            mu_dead = l == L - 1 and not _env("GEOSSL_PAINN_NO_MU_ZERO")
            q3, mu3 = torch.empty_like(q), (None if mu_dead else torch.empty(N, 3, F_, **f32))
            call("geossl_painn_mix_post_fwd_dyn", ptr(q2), ptr(mu2), ptr(mm), ptr(xx), ptr(dot), N, F_, ptr(q3), ptr(mu3),
                 dN, st)
            if training:
                saved.append(dict(q=q, mu=mu, u=u, s=s, xc=xc, q2=q2, mu2=mu2, mm=mm, cx=cx, dot=dot, u1=u1, s1=s1, xx=xx))
            q, mu = q3, mu3
        if training:
            ctx.el, ctx.cfg, ctx.z, ctx.ps, ctx.params = el, cfg, z, ps, params
            ctx.pos, ctx.want_params = pos, want_params
            ctx.geom = (dirv, fcut, phi)
            ctx.saved = saved
        return q

    @staticmethod
    def backward(ctx, dq):
        want_pos, want_params = ctx.needs_input_grad[1], ctx.want_params
        if torch.is_grad_enabled():
            # a backward with grad mode ON = create_graph=True (finetune_md17.py:46,99): the gradients may be
            # differentiated again (training on forces, :51-54).  The fused first-order kernels still compute them, as
            # the outputs of a node that knows how to be differentiated (geossl_amd/higher_order.py)
            from ...higher_order import PaiNNGradNode
            dpos, grads = PaiNNGradNode.run(ctx, dq, want_pos, want_params)
            return (None, dpos, None, None) + tuple(grads)
        dpos, grads = _PaiNNCore.fused_backward(ctx, dq, want_pos, want_params, allow_direct=True)
        return (None, dpos, None, None) + tuple(grads)

    @staticmethod
    def fused_backward(ctx, dq, want_pos, want_params, allow_direct):
        """First-order gradients by the fused kernels -> (dpos or None, [one entry per parameter, None = not returned])."""
        el, cfg, ps = ctx.el, ctx.cfg, ctx.ps
        F_, L, R = cfg["F"], cfg["L"], cfg["R"]
        dirv, fcut, phi = ctx.geom
        dev, N = dq.device, dq.size(0)
        st = stream()
        f32 = dict(dtype=torch.float32, device=dev)
        # opt-in (DDMTrainer); otherwise gradients go through autograd
        direct = want_params and allow_direct and _lib.direct_grads_enabled(ctx.params)
        grads = [p.grad for p in ctx.params] if direct else [torch.zeros_like(p) for p in ps]
        acc = 1 if direct else 0
        g_emb, g_fw, g_fb = grads[0], grads[1], grads[2]
        inter = [ps[3 + 4 * l: 7 + 4 * l] for l in range(L)]
        mix = [ps[3 + 4 * L + 5 * l: 8 + 4 * L + 5 * l] for l in range(L)]
        g_inter = [grads[3 + 4 * l: 7 + 4 * l] for l in range(L)]
        g_mix = [grads[3 + 4 * L + 5 * l: 8 + 4 * L + 5 * l] for l in range(L)]
        inc_ptr, inc_idx = el.inc["j"]
        dyn = getattr(el, "dyn", None)
        dN, dN3 = (dyn.n_atoms2, dyn.n_atoms2x3) if dyn is not None else (None, None)
        if dyn is not None and want_pos:
            raise _lib.GeosslHipError("a capacity-bucket layout serves the step without position gradients")
        dq_cur = dq.contiguous()
        # d(last block's mu') = 0, as NULL: no 3 N F zeros written, read by two kernels and added as a residual
        dmu_cur = torch.zeros(N, 3, F_, **f32) if _env("GEOSSL_PAINN_NO_MU_ZERO") else None
        groups = {}  # (rows, lda, ldb, ldw) -> list of problems, all with M = N = F

        def add(rows, lda, ldb, ldw, A, Bm, dW, db):
            groups.setdefault((rows, lda, ldb, ldw), []).append((A, Bm, dW, db))

        blocks = []
        for l in range(L):
            c0w, c0b, c1w, c1b = inter[l]
            i0w, i0b, i1w, i1b, mw = mix[l]
            blocks += [c0w] + [c1w[c * F_:(c + 1) * F_] for c in range(3)] + [mw[:F_], mw[F_:]] + \
                      [i0w[:, :F_], i0w[:, F_:]] + [i1w[c * F_:(c + 1) * F_] for c in range(3)]
        img = ops.prepare_chain(blocks, transB=False) if not _env("GEOSSL_PAINN_NO_CHAIN") else None
        NB = 11
        fused = img is not None and F_ == 128 and not _env("GEOSSL_PAINN_SILU_KERNELS")

        def lin_t(x, w, k, res=None, out=None):  # x @ w (the transposed use of a forward weight block)
            if img is None:
                return ops.linear(x, blocks[k].contiguous(), transB=False, res=res, out=out)
            return ops.linear_chain(x, [dict(image=img[k], res=res, out=out)], dyn_rows=dN if x.size(0) == N else dN3)[0]

        lay = cfg["lay"]
        def lin_t_sum(xs_list, ks, res=None):
            """sum_c xs_c @ W_c (+ res): the passes of a contraction over a wide input, one launch at F = 128"""
            if img is None or F_ != 128:
                acc_ = res
                for x_, k_ in zip(xs_list, ks):
                    acc_ = lin_t(x_, None, k_, res=acc_)
                return acc_
            stages = []
            for n_, (x_, k_) in enumerate(zip(xs_list, ks)):
                sd = dict(image=img[k_], store=(n_ == len(ks) - 1))
                if n_ == 0:
                    sd["res"] = res
                    if res is not None and len(ks) > 1:
                        sd["store"] = False
                else:
                    sd.update(x=x_, add_prev=True)
                stages.append(sd)
            if res is not None and len(ks) > 1:
                # the residual belongs to the stage that stores: move it there (the row stride must be the output's)
                stages[0].pop("res")
                stages[-1]["res"] = res
            return ops.linear_chain(xs_list[0], stages, dyn_rows=dN if xs_list[0].size(0) == N else dN3)[-1]

        nfl = _lib.load().geossl_painn_interaction_bwd_mol_workspace_floats(N, lay.B, F_, R)
        ws = torch.empty(max(int(nfl), 1), **f32)
        # molecules above the LDS rows of the molecule-staged backward: skipped there, covered by the per-atom kernel
        from ...layout import painn_stage_caps
        cap_b = painn_stage_caps(F_, R)[1]
        big_b = lay.big_atoms(cap_b) if (0 < cap_b < lay.max_n and F_ in (64, 128)) else None
        keep = []
        E = el.E
        if want_pos:  # dL/d(phi, fcut, dir) per edge, summed over the blocks (painn_force.hip)
            dphi, dfc, ddir = torch.empty(max(E, 1), R, **f32), torch.empty(max(E, 1), **f32), torch.empty(max(E, 1), 3, **f32)
        for l in reversed(range(L)):
            sv = ctx.saved[l]
            c0w, c0b, c1w, c1b = inter[l]
            i0w, i0b, i1w, i1b, mw = mix[l]
            gc0w, gc0b, gc1w, gc1b = g_inter[l]
            gi0w, gi0b, gi1w, gi1b, gmw = g_mix[l]
            k0 = NB * l
            # ---- mixing block
            dxx, dmm = torch.empty(N, 3 * F_, **f32), torch.empty(3 * N, 2 * F_, **f32)
            call("geossl_painn_mix_post_bwd_dyn", ptr(dq_cur), ptr(dmu_cur), ptr(sv["mm"]), ptr(sv["xx"]), ptr(sv["dot"]), N,
                 F_, ptr(dxx), ptr(dmm), dN, st)
            for c, xs_ in enumerate(_split3(dxx, F_)):
                add(N, 3 * F_, F_, F_, xs_, sv["s1"], gi1w[c * F_:(c + 1) * F_], gi1b[c * F_:(c + 1) * F_])
            dctx = torch.empty(N, 2 * F_, **f32)                           # [N][2F] = du1 @ i0w
            if fused:  # (sum_c dxx_c W_c) * silu'(u1) = du1, then du1 @ i0w: one launch
                du1 = torch.empty(N, F_, **f32)
                x3 = _split3(dxx, F_)
                ops.linear_chain(x3[0], [dict(image=img[k0 + 8], store=False),
                                         dict(image=img[k0 + 9], x=x3[1], add_prev=True, store=False),
                                         dict(image=img[k0 + 10], x=x3[2], add_prev=True, tprev=sv["u1"], out=du1,
                                              flags=_lib.EPI_MUL_DSILU),
                                         dict(image=img[k0 + 6], out=dctx[:, :F_]),
                                         dict(image=img[k0 + 7], out=dctx[:, F_:], same_input=True)], dyn_rows=dN)
            else:
                ds1 = lin_t_sum(_split3(dxx, F_), [k0 + 8 + c for c in range(3)])
                du1 = torch.empty_like(ds1)
                call("geossl_silu_bwd", ptr(sv["u1"]), ptr(ds1), ds1.numel(), ptr(du1), st)
                if img is not None and F_ == 128:
                    ops.linear_chain(du1, [dict(image=img[k0 + 6], out=dctx[:, :F_]),
                                           dict(image=img[k0 + 7], out=dctx[:, F_:], same_input=True)])
                else:
                    for c in range(2):
                        lin_t(du1, i0w[:, c * F_:(c + 1) * F_], k0 + 6 + c, out=dctx[:, c * F_:(c + 1) * F_])
            for c in range(2):
                add(N, F_, 2 * F_, 2 * F_, du1, sv["cx"][:, c * F_:(c + 1) * F_], gi0w[:, c * F_:(c + 1) * F_],
                    gi0b if c == 0 else None)
            dq2 = torch.empty(N, F_, **f32)
            call("geossl_painn_mix_pre_bwd_dyn", ptr(dq_cur), ptr(dctx), ptr(sv["cx"]), ptr(sv["mm"]), N, F_, ptr(dq2),
                 ptr(dmm), dN, st)
            # d mu (after interaction): contraction over the 2F columns of dmm in two F-wide passes
            dmu2 = lin_t_sum([dmm[:, :F_], dmm[:, F_:]], [k0 + 4, k0 + 5],
                             res=None if dmu_cur is None else dmu_cur.view(3 * N, F_))
            for c in range(2):
                add(3 * N, 2 * F_, F_, F_, dmm[:, c * F_:(c + 1) * F_], sv["mu2"].view(3 * N, F_),
                    gmw[c * F_:(c + 1) * F_], None)
            # ---- interaction block
            if want_pos:
                call("geossl_painn_edge_grads", ptr(dq2), ptr(dmu2), ptr(sv["mu"]), ptr(sv["xc"]), ptr(el.idx_i),
                     ptr(el.idx_j), ptr(phi), ptr(fcut), ptr(dirv), ptr(ps[1][l * 3 * F_:(l + 1) * 3 * F_]),
                     ptr(ps[2][l * 3 * F_:(l + 1) * 3 * F_]), E, F_, R, ptr(dphi), ptr(dfc), ptr(ddir),
                     0 if l == L - 1 else 1, st)
            mu_l = sv["mu"]
            if mu_l is None and big_b is not None:   # (cannot happen with today's stage caps: the forward's is the smaller)
                mu_l = torch.zeros(N, 3, F_, **f32)
            # (the first interaction's mu is identically zero and its gradient is nobody's input: NULL for both)
            dxc, dmu_in = torch.empty(N, 3 * F_, **f32), (None if mu_l is None else torch.empty(N, 3, F_, **f32))
            call("geossl_painn_interaction_bwd_mol" if big_b is None else "geossl_painn_interaction_bwd_mol_skip",
                 ptr(dq2), ptr(dmu2), ptr(mu_l), ptr(sv["xc"]), ptr(el.idx_i),
                 ptr(inc_ptr), ptr(inc_idx), ptr(phi), ptr(fcut), ptr(dirv), ptr(ps[1][l * 3 * F_:(l + 1) * 3 * F_]),
                 ptr(ps[2][l * 3 * F_:(l + 1) * 3 * F_]), ptr(lay.mol_ptr), lay.B, lay.max_n, N, F_, R, ptr(dxc), ptr(dmu_in),
                 ptr(g_fw[l * 3 * F_:(l + 1) * 3 * F_]), ptr(g_fb[l * 3 * F_:(l + 1) * 3 * F_]), ptr(ws), acc, st)
            if big_b is not None and big_b[1] > 0:
                call("geossl_painn_interaction_bwd_atoms", ptr(dq2), ptr(dmu2), ptr(mu_l), ptr(sv["xc"]), ptr(el.idx_i),
                     ptr(inc_ptr), ptr(inc_idx), ptr(phi), ptr(fcut), ptr(dirv), ptr(ps[1][l * 3 * F_:(l + 1) * 3 * F_]),
                     ptr(ps[2][l * 3 * F_:(l + 1) * 3 * F_]), ptr(big_b[0]), big_b[1], big_b[2], F_, R, ptr(dxc), ptr(dmu_in),
                     ptr(g_fw[l * 3 * F_:(l + 1) * 3 * F_]), ptr(g_fb[l * 3 * F_:(l + 1) * 3 * F_]), ptr(ws), 1, st)
            for c, xs_ in enumerate(_split3(dxc, F_)):
                add(N, 3 * F_, F_, F_, xs_, sv["s"], gc1w[c * F_:(c + 1) * F_], gc1b[c * F_:(c + 1) * F_])
            if fused:  # (sum_c dxc_c W_c) * silu'(u) = du, then du @ c0w + dq2 (residual q2 = q + dq): one launch
                du, dq_in = torch.empty(N, F_, **f32), torch.empty(N, F_, **f32)
                x3 = _split3(dxc, F_)
                ops.linear_chain(x3[0], [dict(image=img[k0 + 1], store=False),
                                         dict(image=img[k0 + 2], x=x3[1], add_prev=True, store=False),
                                         dict(image=img[k0 + 3], x=x3[2], add_prev=True, tprev=sv["u"], out=du,
                                              flags=_lib.EPI_MUL_DSILU),
                                         dict(image=img[k0], res=dq2, out=dq_in)], dyn_rows=dN)
            else:
                ds = lin_t_sum(_split3(dxc, F_), [k0 + 1 + c for c in range(3)])
                du = torch.empty_like(ds)
                call("geossl_silu_bwd", ptr(sv["u"]), ptr(ds), ds.numel(), ptr(du), st)
                dq_in = lin_t(du, c0w, k0, res=dq2)                          # residual q2 = q + dq
            add(N, F_, F_, F_, du, sv["q"], gc0w, gc0b)
            keep += [dxx, dmm, du1, dxc, du, dq2, dmu2]
            dq_cur, dmu_cur = dq_in, dmu_in
        dpos = None
        if want_pos:
            dpos = torch.zeros(N, 3, **f32)
            if E > 0:
                dr = torch.empty(E, 3, **f32)
                call("geossl_painn_edge_geom_bwd", ptr(ctx.pos), ptr(el.idx_i), ptr(el.idx_j), E, cfg["cutoff"],
                     ptr(cfg["offsets"]), ptr(cfg["widths"]), R, ptr(dphi), ptr(dfc), ptr(ddir), ptr(dr), st)
                (pi, ii), (pj, ij) = el.inc["i"], el.inc["j"]
                call("geossl_painn_position_grad", ptr(dr), ptr(pi), ptr(ii), ptr(pj), ptr(ij), N, ptr(dpos), st)
        if not want_params:  # an evaluation of the forces alone (finetune_md17.py:85-105 with frozen weights)
            return dpos, [None] * len(grads)
        for (rows, lda, ldb, ldw), probs in groups.items():
            ops.linear_wgrad(probs, rows, F_, F_, accumulate=bool(acc), lda=lda, ldb=ldb, ldw=ldw,
                             dyn_rows=dN if rows == N else dN3)
        # embedding table; padding_idx = 0 keeps row 0 without gradient (painn.py:174)
        z = ctx.z
        emb_w = ps[0]
        tmp = torch.empty_like(emb_w)
        wsf = torch.empty(int(_lib.load().geossl_embedding_bwd_workspace_floats(emb_w.size(0), F_)), **f32)
        call("geossl_embedding_bwd_dyn", ptr(z), z.stride(0) if z.numel() else 1, ptr(dq_cur), emb_w.size(0), N, F_, ptr(tmp),
             ptr(wsf), 0, dN, st)
        tmp[0].zero_()
        if direct:
            call("geossl_axpy", ptr(g_emb), ptr(tmp), 1.0, tmp.numel(), ptr(g_emb), st)
            return dpos, [None] * len(grads)
        g_emb.copy_(tmp)
        # ctx.saved stays: finetune_md17.py:46 differentiates with retain_graph=True and runs this node again
        return dpos, [g if p.requires_grad else None for g, p in zip(grads, ctx.params)]
