"""Thin functional wrappers over the C ABI (one call = one or a few kernel launches on the current
stream).  Every function takes/returns CUDA tensors; nothing here computes on the host."""
import ctypes as C
import math
import os

import numpy as np
import torch
from .switches import env as _env

from . import _lib
from ._lib import call, ptr, stream
from .layout import MolLayout, get_layout

PI_F32 = float(torch.tensor(math.pi, dtype=torch.float32))
# molecules of the (two-view) batch up to which ragged batches take the layer loop.  Same-box A/B on set B (trainer, one
# bucket graph), loop against separate launches: 128 molecules per view 0.865 -> 0.820 ms (+5.5 %), 256 1.18 -> 1.23 ms,
# 512 1.75 -> 2.06 ms: with one block per molecule an operation of the loop is a 9 us chain of dependent round trips
# (weights from L2, rows through L2 between operations) whatever the batch, and from 256 molecules per view on the
# separate launches fill the chip better.  (First form, every wave the unrolled walk of its molecule's size class: 0.839
# ms at 128 with 31 spilled registers; the block form above has none.)
RAGGED_LOOP_MAX_MOLS = int(_env("GEOSSL_RAGGED_LOOP_MAX", 256))


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return t.contiguous()


def radius_cap(max_num_neighbors, loop=False):
    # torch_cluster.radius_graph searches max_num_neighbors (+1 when loop=False: the self hit) candidates
    return max_num_neighbors if loop else max_num_neighbors + 1


def radius_graph(pos, r, batch=None, loop=False, max_num_neighbors=32, layout=None, return_weight=False):
    """torch_geometric.nn.radius_graph(pos, r, batch) as called at schnet.py:91 and
    datasets_3D_Radius.py:120 -> int64 [2, E] = [source j; target i], target-major, sources ascending."""
    _lib.require_cuda(pos)
    if loop:
        raise NotImplementedError("loop=True is not on the GeoSSL path")
    pos = _f32(pos)
    N = pos.size(0)
    if batch is None:
        batch = torch.zeros(N, dtype=torch.long, device=pos.device)
    lay = layout or get_layout(batch)
    r2 = float(torch.tensor(float(r) * float(r), dtype=torch.float32))
    cap = radius_cap(max_num_neighbors)
    deg = torch.zeros(N, dtype=torch.int32, device=pos.device)
    call("geossl_radius_graph_count", ptr(pos), ptr(lay.mol_ptr), lay.B, lay.max_n, r2, cap, ptr(deg), stream())
    edge_ptr = torch.zeros(N + 1, dtype=torch.int64, device=pos.device)
    edge_ptr[1:] = torch.cumsum(deg, 0, dtype=torch.int64)
    E = int(edge_ptr[-1].item())
    edge_index = torch.empty(2, E, dtype=torch.int64, device=pos.device)
    weight = torch.empty(E, dtype=torch.float32, device=pos.device)
    if E > 0:
        call("geossl_radius_graph_fill", ptr(pos), ptr(lay.mol_ptr), lay.B, lay.max_n, r2, cap, ptr(edge_ptr),
             ptr(edge_index[0]), ptr(edge_index[1]), ptr(weight), stream())
    return (edge_index, weight) if return_weight else edge_index


def pair_geometry(pos, layout, cutoff, max_num_neighbors=32):
    """Radius graph in pair-slot form: (pair_d [P], pair_c [P], pair_flag [P] u8)."""
    pos = _f32(pos)
    dev = pos.device
    P = layout.P
    pair_d = torch.empty(P, dtype=torch.float32, device=dev)
    pair_c = torch.empty(P, dtype=torch.float32, device=dev)
    pair_flag = torch.empty(P, dtype=torch.uint8, device=dev)
    r2 = float(torch.tensor(float(cutoff) * float(cutoff), dtype=torch.float32))
    if P > 0:
        call("geossl_pair_geometry", ptr(pos), ptr(layout.mol_ptr), ptr(layout.pair_ptr), layout.B, layout.max_n, r2,
             radius_cap(max_num_neighbors), float(cutoff), ptr(pair_d), ptr(pair_c), ptr(pair_flag), stream())
    return pair_d, pair_c, pair_flag


def gaussian_smearing(dist, offset, coeff):
    """GaussianSmearing.forward, schnet.py:205-207."""
    _lib.require_cuda(dist)
    d = _f32(dist).view(-1)
    out = torch.empty(d.numel(), offset.numel(), dtype=torch.float32, device=d.device)
    call("geossl_rbf_fwd", ptr(d), d.numel(), ptr(offset), offset.numel(), float(coeff), ptr(out), stream())
    return out


class PreparedWeight:
    """MFMA operand image of a Linear weight (geossl_linear_prepare): `image` is an int32 tensor; K / NO are the
    contraction and output widths of the product it was built for."""
    __slots__ = ("image", "K", "NO")

    def __init__(self, image, K, NO):
        self.image, self.K, self.NO = image, K, NO


def prepare_linear(weights, transB=True):
    """Convert Linear weights (all the same shape) to their operand images in one launch per GEOSSL_TN_MAX weights.
    transB as in `linear`.  Returns a list of PreparedWeight, or None when the shape has no prepared path."""
    w0 = weights[0]
    NO, K = (w0.size(0), w0.size(1)) if transB else (w0.size(1), w0.size(0))
    words = int(_lib.load().geossl_linear_image_words(K, NO))
    if words == 0:
        return None
    out = []
    for lo in range(0, len(weights), _lib.TN_MAX):
        chunk = weights[lo:lo + _lib.TN_MAX]
        images = torch.empty(len(chunk), words, dtype=torch.int32, device=w0.device)
        pb = _lib.PrepareBatch()
        for i, w in enumerate(chunk):
            assert w.shape == w0.shape and w.stride(1) == 1
            pb.W[i], pb.image[i] = ptr(w), ptr(images[i])
            pb.ldw[i] = 0 if w.is_contiguous() else w.stride(0)
        call("geossl_linear_prepare", C.byref(pb), len(chunk), K, NO, 1 if transB else 0, stream())
        out += [PreparedWeight(images[i], K, NO) for i in range(len(chunk))]
    return out


def linear(x, w, bias=None, res=None, tprev=None, transB=True, flags=0, out=None, K=None, NO=None):
    """Y = epi(X @ Bm); transB: w is torch layout [NO][K] (forward) else [K][NO] (dX = dY @ W).
    x / out may be column slices of wider row-major tensors (row stride = x.stride(0) / out.stride(0));
    K / NO default to the slice widths.  `w` may be a PreparedWeight (transB is then fixed by its image)."""
    R = x.size(0)
    prepared = isinstance(w, PreparedWeight)
    if prepared:
        K, NO = w.K, w.NO
    K = x.size(1) if K is None else K
    NO = (w.size(0) if transB else w.size(1)) if NO is None else NO
    if out is None:
        out = torch.empty(R, NO, dtype=torch.float32, device=x.device)
    if bias is not None:
        flags |= _lib.EPI_BIAS
    if res is not None:
        flags |= _lib.EPI_RESIDUAL
    if tprev is not None:
        flags |= _lib.EPI_MUL_DSSP
    assert x.stride(1) == 1 and out.stride(1) == 1
    for aux in (res, tprev):
        assert aux is None or (aux.stride(0) == out.stride(0) and aux.stride(1) == 1)
    if prepared:
        call("geossl_linear_prepared", ptr(x), x.stride(0), ptr(w.image), ptr(bias), ptr(res), ptr(tprev), ptr(out),
             out.stride(0), R, K, NO, flags, stream())
    else:
        call("geossl_linear", ptr(x), x.stride(0), ptr(w), ptr(bias), ptr(res), ptr(tprev), ptr(out), out.stride(0), R,
             K, NO, 1 if transB else 0, flags, stream())
    return out


def prepare_chain(weights, transB=True, both=False):
    """Operand images (geossl_chain_prepare) of square F x F Linear weights for `linear_chain`, one launch per
    GEOSSL_PREPARE_MAX images.  transB as in `linear`.  A weight may be a column block of a wider matrix (unit column
    stride, row stride a multiple of 4, 16-byte aligned): it is converted where it lies.  Returns a list of int32
    tensors, or None if F has no chain path; `both`: (forward images (transB=True), backward images (transB=False)) of
    every weight from the same launch."""
    w0 = weights[0]
    F = w0.size(0)
    words = int(_lib.load().geossl_chain_image_words(F)) if w0.size(1) == F else 0
    if words == 0:
        return (None, None) if both else None
    jobs = [(w, 1) for w in weights] + [(w, 2) for w in weights] if both else [(w, 0) for w in weights]
    out = []
    for lo in range(0, len(jobs), _lib.PREPARE_MAX):
        chunk = jobs[lo:lo + _lib.PREPARE_MAX]
        images = torch.empty(len(chunk), words, dtype=torch.int32, device=w0.device)
        pb = _lib.PrepareBatch()
        for i, (w, tb) in enumerate(chunk):
            assert w.shape == w0.shape and w.stride(1) == 1
            pb.W[i], pb.image[i] = ptr(w), ptr(images[i])
            pb.ldw[i] = 0 if w.is_contiguous() else w.stride(0)
            pb.tb[i] = tb
        call("geossl_chain_prepare", C.byref(pb), len(chunk), F, 1 if transB else 0, stream())
        out += [images[i] for i in range(len(chunk))]
    return (out[:len(weights)], out[len(weights):]) if both else out


def _dyn(layout_or_dyn, field):
    """Device address of a bucket's real count (bucket.DynDims), or None."""
    d = getattr(layout_or_dyn, "dyn", layout_or_dyn)
    return None if d is None else getattr(d, field)


def linear_chain(x, stages, dyn_rows=None):
    """Several F -> F Linear layers applied to the rows of x back to back in one launch (geossl_linear_chain).
    stages: list of dicts with `image` (from prepare_chain) and optional `bias`, `res`, `tprev`, `flags`, `store`,
    `same_input` (F = 128 only: the stage reads the input of the stage before it, not its result), `x` (F = 128 only:
    the stage reads its own input rows) with `add_prev` (and adds the result of the stage before it)
    (default True: the stage's result is written to a new [R, F] tensor).  F = 128 only: flags EPI_SILU (the stage
    stores its result as it is, hands silu of it to the next stage and writes that to `out_act` if given) and
    EPI_MUL_DSILU (tprev is a saved pre-activation: * silu'(tprev)); up to five stages.  Returns the list of stored
    results (None where store is False)."""
    R, F = x.shape  # (x may be a column slice: its row stride is passed)
    assert x.stride(1) == 1 and 1 <= len(stages) <= _lib.CHAIN_MAX
    ch = _lib.Chain()
    ch.nstage = len(stages)
    outs = []
    for s, sd in enumerate(stages):
        st = ch.st[s]
        o = sd.get("out")  # a preallocated [R, F] destination (e.g. a row slice of a larger tensor) ...
        if o is None and sd.get("store", True):  # ... or a new tensor, unless the stage is not stored at all
            o = torch.empty(R, F, dtype=torch.float32, device=x.device)
        # out / res / tprev of a stage share one row stride (they may be column slices of wider row-major tensors)
        rows_ = [a for a in (o, sd.get("res"), sd.get("tprev"), sd.get("out_act")) if a is not None]
        ld = rows_[0].stride(0) if rows_ else F
        for a in rows_:
            assert a.stride(0) == ld and a.stride(1) == 1 and a.size(0) == R and a.size(1) == F
        st.image, st.bias, st.res, st.tprev, st.out = (ptr(sd["image"]), ptr(sd.get("bias")), ptr(sd.get("res")),
                                                       ptr(sd.get("tprev")), ptr(o))
        st.out_act = ptr(sd.get("out_act"))
        st.ld, st.flags = ld, int(sd.get("flags", 0)) | (_lib.CHAIN_SAME_INPUT if sd.get("same_input") else 0)
        xin = sd.get("x")  # the stage's own input rows (F = 128): one of several F-wide passes over a wide input
        if xin is not None:
            assert s > 0 and xin.stride(1) == 1 and xin.size(0) == R and xin.size(1) == F
            st.xin, st.ldxin = ptr(xin), xin.stride(0)
            st.flags |= _lib.CHAIN_NEW_INPUT | (_lib.CHAIN_ADD_PREV if sd.get("add_prev") else 0)
        else:
            st.xin, st.ldxin = None, 0
        outs.append(o)
    # dyn_rows: device address of the real row count when R is a capacity (bucket.DynDims; F = 128 only)
    call("geossl_linear_chain_dyn", ptr(x), x.stride(0), C.byref(ch), R, F, dyn_rows, stream())
    return outs


# GeosslLoopOp as 64-bit words (include/geossl_hip.h; checked against the ctypes layout when the module is imported):
# word 0 = kind | swap << 32, 1 = X, 2 = Wf, 3 = out, 4 = chain.nstage, then 9 words per stage: image, bias, res, tprev,
# out, ld | flags << 32, xin, ldxin, out_act
_LOOP_WORDS, _LOOP_STAGE0, _LOOP_STAGE_WORDS = 50, 5, 9
assert C.sizeof(_lib.LoopOp) == 8 * _LOOP_WORDS and _lib.LoopOp.chain.offset == 32 and _lib.Chain.st.offset == 8 \
    and C.sizeof(_lib.ChainStage) == 8 * _LOOP_STAGE_WORDS and _lib.ChainStage.ld.offset == 40 \
    and _lib.ChainStage.out_act.offset == 64


def layer_loop(ops_list, layout, pair_flag, N, F, stagger=0):
    """The operations of `ops_list` - ("chain", x, stages) / ("agg", x, Wf_l, out, swap) - as ONE launch in which every
    block carries its own molecules through all of them (geossl_schnet_layer_loop).  Returns False when the shape has no
    such path (the caller then launches them one by one).  The operation list is written as plain 64-bit words (a
    field-by-field ctypes fill of 14 operations cost more host time than the 14 calls it replaces)."""
    plan, nblk = layout.loop_plan()
    # RAGGED molecules of a SMALL batch (the reference's batch sizes): every launch of the pass is then a 5 .. 9 us
    # latency and the loop removes 13 of the 14; blocks of one or two molecules found from mol_ptr on the device (a
    # capacity bucket's index structures are device data).  Beyond RAGGED_LOOP_MAX_MOLS the ragged loop loses to the
    # separate launches (a block owns its molecules for the whole pass: 650-670 us per pass against 470 us at 2 x 1024
    # molecules) and is not used.
    ragged = (plan is None and F == 128 and 1 < layout.max_n <= 33 and layout.B <= RAGGED_LOOP_MAX_MOLS
              and not _env("GEOSSL_NO_RAGGED_LOOP") and len(ops_list) <= _lib.LOOP_MAX_OPS)
    if not ragged and (plan is None or F != 128 or not layout.uniform or layout.max_n > 20
                       or len(ops_list) > _lib.LOOP_MAX_OPS):
        return False
    words = [0] * (_LOOP_WORDS * len(ops_list))
    dp = lambda t_: 0 if t_ is None else t_.data_ptr()
    for i, op in enumerate(ops_list):
        b = _LOOP_WORDS * i
        if op[0] == "chain":
            _, x, stages = op
            assert x.shape == (N, F) and x.is_contiguous() and len(stages) <= 3
            words[b + 1], words[b + 4] = x.data_ptr(), len(stages)
            for s, sd in enumerate(stages):
                w = b + _LOOP_STAGE0 + _LOOP_STAGE_WORDS * s
                o, r, tp = sd.get("out"), sd.get("res"), sd.get("tprev")
                for a in (o, r, tp):
                    assert a is None or (a.shape == (N, F) and a.is_contiguous())
                words[w], words[w + 1], words[w + 2], words[w + 3], words[w + 4] = (
                    sd["image"].data_ptr(), dp(sd.get("bias")), dp(r), dp(tp), dp(o))
                words[w + 5] = F | (int(sd.get("flags", 0)) << 32)
        else:
            _, x, Wf_l, out, swap = op
            words[b], words[b + 1], words[b + 2], words[b + 3] = 1 | ((1 if swap else 0) << 32), x.data_ptr(), \
                Wf_l.data_ptr(), out.data_ptr()
    arr = np.array(words, dtype=np.uint64)
    if ragged:
        call("geossl_schnet_layer_loop_ragged", arr.ctypes.data, len(ops_list), ptr(layout.mol_ptr), ptr(layout.pair_ptr),
             ptr(pair_flag), layout.B, 1 if layout.B <= 512 else 2, N, F, stream())
        return True
    call("geossl_schnet_layer_loop", arr.ctypes.data, len(ops_list), ptr(plan), nblk, ptr(layout.mol_ptr),
         ptr(layout.pair_ptr), ptr(pair_flag), layout.max_n, 1 if layout.uniform else 0, N, F, int(stagger), stream())
    return True


def linear_wgrad(problems, R, M, N, accumulate=False, lda=None, ldb=None, ldw=None, dyn_rows=None):
    """Batched weight gradients.  problems: list of (A [R,M], B [R,N], dW [M,N], db [M] or None); lda/ldb/ldw are
    the row strides when A / B / dW are column slices of wider tensors."""
    dev = problems[0][0].device
    lda, ldb, ldw = lda or M, ldb or N, ldw or N
    for lo in range(0, len(problems), _lib.TN_MAX):
        chunk = problems[lo:lo + _lib.TN_MAX]
        tb = _lib.TnBatch()
        for i, (A, Bm, dW, db) in enumerate(chunk):
            tb.A[i], tb.B[i], tb.dW[i], tb.db[i] = ptr(A), ptr(Bm), ptr(dW), ptr(db)
        nfl = _lib.load().geossl_tn_workspace_floats(R, M, N, len(chunk))
        ws = torch.empty(nfl, dtype=torch.float32, device=dev)
        call("geossl_linear_wgrad_dyn", C.byref(tb), len(chunk), R, M, N, lda, ldb, ldw, ptr(ws), 1 if accumulate else 0,
             dyn_rows, stream())


def aggregate(x, Wf_l, pair_flag, layout, swap=False, out=None, mols=None):
    """x, out: full [N, F] tensors (rows are addressed by global atom index).  mols = (m0, m1, order): only molecules
    m0 .. m1-1 are processed (their rows of `out` written), started in the sequence `order` (int32 global molecule ids,
    or None for m0, m0+1, ...): independent sections of a batch can then run on different streams."""
    N, F = x.shape
    if out is None:
        out = torch.empty_like(x)
    work = getattr(layout, "agg_work", None)
    if mols is None and work is not None and 32 < F <= 128:
        call("geossl_cfconv_aggregate_targets_dyn" if getattr(layout, "agg_targets", False)
             else "geossl_cfconv_aggregate_work_dyn", ptr(x), ptr(Wf_l), ptr(pair_flag), ptr(layout.mol_ptr),
             ptr(layout.pair_ptr), ptr(work), work.numel(), layout.max_n, F, 1 if swap else 0, ptr(out),
             _dyn(layout, "n_work"), stream())
        return out
    if getattr(layout, "dyn", None) is not None:
        raise _lib.GeosslHipError("a capacity-bucket layout needs the work-list aggregation (64 or 128 features)")
    if mols is None:
        mp, pp, order, B = ptr(layout.mol_ptr), ptr(layout.pair_ptr), layout.order, layout.B
    else:
        m0, m1, order = mols
        B = m1 - m0
        if order is not None:  # ids are global: the pointer arrays stay whole
            mp, pp = ptr(layout.mol_ptr), ptr(layout.pair_ptr)
        else:                  # molecule blockIdx.x of the launch is m0 + blockIdx.x: shift the (int32) pointer arrays
            mp, pp = ptr(layout.mol_ptr) + 4 * m0, ptr(layout.pair_ptr) + 4 * m0
    call("geossl_cfconv_aggregate", ptr(x), ptr(Wf_l), ptr(pair_flag), mp, pp, ptr(order), B, layout.max_n, F,
         1 if swap else 0, ptr(out), stream())
    return out


def pair_product(a, b, layout, pair_flag, swap=False):
    """d aggregate / d filter rows as a tensor [P, F]: f0 a[i] b[j] + f1 a[j] b[i] per pair slot."""
    N, F = a.shape
    out = torch.empty(layout.P, F, dtype=torch.float32, device=a.device)
    call("geossl_pair_product", ptr(a), ptr(b), ptr(layout.pair_i), ptr(layout.pair_j), ptr(pair_flag), layout.P, F,
         1 if swap else 0, ptr(out), stream())
    return out


def segment_reduce(h, layout, reduce):
    """torch_scatter.scatter(h, batch, dim=0, reduce) for a sorted batch (schnet.py:115)."""
    out = torch.empty(layout.B, h.size(1), dtype=torch.float32, device=h.device)
    call("geossl_segment_reduce_fwd", ptr(h), ptr(layout.mol_ptr), layout.B, h.size(1), 1 if reduce == "mean" else 0,
         ptr(out), stream())
    return out


def pair_distance(pos, sei0, sei1):
    """pretrain_GeoSSL.py:199-205 -> [S, 1]."""
    pos = _f32(pos)
    S = sei0.numel()
    out = torch.empty(S, 1, dtype=torch.float32, device=pos.device)
    call("geossl_pair_distance", ptr(pos), ptr(sei0), ptr(sei1), S, ptr(out), stream())
    return out


def ddm_views(pos, noise, sei0, sei1, z=None, dyn=None):
    """Both views of a DDM step in one launch (pretrain_GeoSSL.py:68-74,199-205): ([pos ; pos + noise] as one [2N, 3]
    tensor, super-edge lengths of the clean view [S, 1], of the perturbed view [S, 1]); with the atom types `z` [N]
    (any stride) also [z ; z]."""
    pos, noise = _f32(pos), _f32(noise)
    N, S = pos.size(0), sei0.numel()
    pos2 = torch.empty(2 * N, 3, dtype=torch.float32, device=pos.device)
    d01 = torch.empty(S, 1, dtype=torch.float32, device=pos.device)
    d02 = torch.empty(S, 1, dtype=torch.float32, device=pos.device)
    z2 = None
    if z is not None:
        assert z.dim() == 1 and z.dtype == torch.long and z.numel() == N
        z2 = torch.empty(2 * N, dtype=torch.long, device=pos.device)
    # dyn (bucket.DynDims): N and S are capacities, the real counts are read on the device; view 1 starts at row dims[N]
    call("geossl_ddm_views_dyn", ptr(pos), ptr(noise), ptr(sei0), ptr(sei1), N, S, ptr(pos2), ptr(d01), ptr(d02), ptr(z),
         z.stride(0) if z is not None and N > 0 else 1, ptr(z2), _dyn(dyn, "n_atoms"), _dyn(dyn, "n_super"), stream())
    return (pos2, d01, d02) if z is None else (pos2, d01, d02, z2)


def add_scaled(a, b, alpha=1.0):
    a, b = _f32(a), _f32(b)
    out = torch.empty_like(a)
    call("geossl_axpy", ptr(a), ptr(b), float(alpha), a.numel(), ptr(out), stream())
    return out


class _RowNormalize(torch.autograd.Function):
    """F.normalize(h, dim=-1) (pretrain_GeoSSL.py:193-195) on the HIP path."""

    @staticmethod
    def forward(ctx, h, eps):
        h = _f32(h)
        N, F = h.shape
        y = torch.empty_like(h)
        norm = torch.empty(N, dtype=torch.float32, device=h.device) if ctx.needs_input_grad[0] else None
        call("geossl_row_normalize_fwd", ptr(h), N, F, float(eps), ptr(y), ptr(norm), stream())
        ctx.save_for_backward(y, norm)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        y, norm = ctx.saved_tensors
        N, F = y.shape
        dh = torch.empty_like(y)
        call("geossl_row_normalize_bwd", ptr(g.contiguous()), ptr(y), ptr(norm), N, F, ctx.eps, ptr(dh), stream())
        return dh, None


def row_normalize(h, eps=1e-12):
    _lib.require_cuda(h)
    return _RowNormalize.apply(h, eps)


class _EnergyForceLoss(torch.autograd.Function):
    """finetune_md17.py:46-51 after the position gradient: pred_force = -dE/dpos and
    loss = c_E * criterion(pred_energy, actual_energy) + c_F * criterion(pred_force, actual_force), criterion = L1Loss
    (:236) or MSELoss, as one node on the element-wise / reduction kernels of csrc/tape.hip (fixed summation order)."""

    @staticmethod
    def forward(ctx, pred_energy, actual_energy, dE_dpos, actual_force, c_e, c_f, kind):
        from . import tape as tp
        pe, ae = _f32(pred_energy).reshape(1, -1), _f32(actual_energy).reshape(1, -1)
        gp, af = _f32(dE_dpos).reshape(1, -1), _f32(actual_force).reshape(1, -1)
        ne, nf = pe.size(1), gp.size(1)
        de = tp._raw_binary(tp.SUB, pe, tp.FULL, ae, tp.FULL, 1, ne)
        df = tp._raw_binary(tp.ADD, gp, tp.FULL, af, tp.FULL, 1, nf, -1.0)       # (-dE/dpos) - actual_force
        if kind == "l1":
            te, tf = tp._raw_unary(tp.ABS, de), tp._raw_unary(tp.ABS, df)
        else:
            te, tf = tp._raw_binary(tp.MUL, de, tp.FULL, de, tp.FULL, 1, ne), tp._raw_binary(tp.MUL, df, tp.FULL, df, tp.FULL, 1, nf)
        se = tp._raw_unary(tp.AFFINE, tp._raw_reduce(tp.ROW, te), c_e / max(ne, 1))
        sf = tp._raw_unary(tp.AFFINE, tp._raw_reduce(tp.ROW, tf), c_f / max(nf, 1))
        ctx.save_for_backward(de, df)
        ctx.meta = (c_e / max(ne, 1), c_f / max(nf, 1), kind, pred_energy.shape, dE_dpos.shape)
        return tp._raw_binary(tp.ADD, se, tp.FULL, sf, tp.FULL, 1, 1).view(())

    @staticmethod
    def backward(ctx, g):
        from . import tape as tp
        de, df = ctx.saved_tensors
        we, wf, kind, shape_e, shape_f = ctx.meta
        g = g.contiguous().view(1, 1)
        if kind == "l1":
            de, df, fe, ff = tp._raw_unary(tp.SIGN, de), tp._raw_unary(tp.SIGN, df), we, -wf
        else:
            fe, ff = 2.0 * we, -2.0 * wf
        d_pe = tp._raw_binary(tp.MUL, de, tp.FULL, g, tp.ROW, 1, de.size(1), fe) if ctx.needs_input_grad[0] else None
        d_gp = tp._raw_binary(tp.MUL, df, tp.FULL, g, tp.ROW, 1, df.size(1), ff) if ctx.needs_input_grad[2] else None
        return (None if d_pe is None else d_pe.view(shape_e), None, None if d_gp is None else d_gp.view(shape_f), None,
                None, None, None)


def energy_force_loss(pred_energy, actual_energy, dE_dpos, actual_force, energy_coeff=0.05, force_coeff=0.95, loss="l1"):
    """The training loss of finetune_md17.py:46-51 from the energies and the position gradient of their sum
    (``dE_dpos = grad(pred_energy, positions, ones, create_graph=True)[0]``; the force is its negative) - on the
    library's kernels, so that a train-on-forces step launches no ATen arithmetic (config.py:59-60 for the defaults)."""
    if loss not in ("l1", "mse"):
        raise ValueError("loss is 'l1' (finetune_md17.py:236) or 'mse'")
    _lib.require_cuda(pred_energy, actual_energy, dE_dpos, actual_force)
    # the kernels walk flat buffers of pred_energy.numel() / dE_dpos.numel() elements: a target of another size would be
    # read past its end (torch's L1Loss / MSELoss raise or broadcast here; broadcasting targets is not supported)
    if actual_energy.numel() != pred_energy.numel():
        raise ValueError("actual_energy has %d elements, pred_energy %d" % (actual_energy.numel(), pred_energy.numel()))
    if tuple(actual_force.shape) != tuple(dE_dpos.shape):
        raise ValueError("actual_force has shape %s, dE_dpos %s" % (tuple(actual_force.shape), tuple(dE_dpos.shape)))
    return _EnergyForceLoss.apply(pred_energy, actual_energy, dE_dpos, actual_force, float(energy_coeff),
                                  float(force_coeff), loss)
