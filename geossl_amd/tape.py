"""A reverse-mode tape over primitives that are closed under differentiation and all run on HIP kernels - the
second-order route of the two backbones without the autograd engine.

Training on forces (examples/finetune_md17.py:46-54) takes ``pred_force = -grad(E, pos, create_graph=True)`` and then
``loss.backward()``: the position gradient of the backbone is differentiated again, with respect to the parameters, the
positions and the upstream gradient.  The fused first-order kernels are one autograd node and cannot be differentiated,
so that node's gradients are the outputs of a second node (higher_order.SchNetGradNode / PaiNNGradNode) whose backward
re-states the backbone from primitives and differentiates the restatement twice.  Until round 4 that restatement was a
torch autograd graph: dense products on the library's kernels, everything else (distance, Gaussian smearing, cosine
envelope, softplus, gathers, index_add, split / cat, and the engine's own gradient accumulation) ATen launches.  Here
the restatement is recorded on a tape of this module's own:

* a value ``V`` is a 2-D fp32 tensor plus the primitive that made it (parents, a VJP rule);
* every VJP rule is written with the same primitives, so the gradients returned by ``grad`` are values on the tape
  and can be differentiated again (that is all ``create_graph`` means here);
* where a value has several consumers its gradient contributions are summed by the tape (``add``, one kernel), not by
  an engine.

Primitives: the three GEMM forms and Linear with its bias (row / column GEMM kernels of the first-order path), the
neighbour aggregation and the pair product (schnet.hip), and the kernels of csrc/tape.hip: element-wise maps (each
with its first and second derivative as maps of their own), broadcast arithmetic over per-row / per-column / xyz-triple
operands with the matching fixed-order reductions, row gather and its adjoint over a sorted incidence list, column
slices / pads / concatenation.  PaiNN's vector features [n, 3, F] live here as the matrix [3 n, F].

Nothing in this module is on the DDM hot path, and nothing in it launches a floating-point ATen kernel (the incidence
lists of a gather's adjoint are built once per index tensor with torch's integer sort).
"""
import math

import torch

from . import ops
from .switches import env as _env
from ._lib import call, ptr, stream

# element-wise maps (csrc/tape.hip: enum Unary)
(AFFINE, EXP, COS, SIN, SSP, SIGMOID, DSIGMOID, D2SIGMOID, RECIP, SQRT, SILU, DSILU, D2SILU, GAUSS, DGAUSS, D2GAUSS, LT, ABS,
 SIGN, DRECIP, D2RECIP, RSQRT, RSQRT3) = range(23)
# operand modes (enum Mode) and operations (enum Binary)
FULL, ROW, COL, THIRD = range(4)
ADD, SUB, MUL, FIRST = range(4)

_DERIV = {EXP: EXP, SIN: COS, SSP: SIGMOID, SIGMOID: DSIGMOID, DSIGMOID: D2SIGMOID, SILU: DSILU, DSILU: D2SILU,
          RECIP: DRECIP, DRECIP: D2RECIP}


# ----------------------------------------------------------------------------------------------------- raw launches
def _empty(R, D, dev):
    return torch.empty(int(R), int(D), dtype=torch.float32, device=dev)


def raw_fill(t, value):
    call("geossl_tape_fill", ptr(t), t.numel(), float(value), stream())
    return t


def raw_full(R, D, dev, value=0.0):
    return raw_fill(_empty(R, D, dev), value)


def raw_copy2d(src, dst, R, C, src_off=0, dst_off=0):
    """dst[:R, dst_off:dst_off+C] = src[:R, src_off:src_off+C] for two row-major matrices (row strides from the tensors)."""
    call("geossl_tape_copy2d", src.data_ptr() + 4 * src_off, src.stride(0), dst.data_ptr() + 4 * dst_off, dst.stride(0),
         int(R), int(C), stream())


def raw_block(t, r0, r1, c0, c1, rows, cols):
    """t[r0:r1, c0:c1] as a contiguous [rows, cols] matrix, zero-padded (t itself when that is the whole of it)."""
    if c0 == 0 and c1 == t.size(1) == cols and rows == r1 - r0 and t.is_contiguous():
        return t if (r0 == 0 and r1 == t.size(0)) else t[r0:r1]   # whole rows of a contiguous matrix: a view
    out = _empty(rows, cols, t.device)
    if rows != r1 - r0 or cols != c1 - c0:
        raw_fill(out, 0.0)
    if r1 > r0 and c1 > c0:
        call("geossl_tape_copy2d", t.data_ptr() + 4 * (r0 * t.stride(0) + c0), t.stride(0), ptr(out), cols, r1 - r0,
             c1 - c0, stream())
    return out


def _up(n, m):
    return (n + m - 1) // m * m


def _rows2d(t):
    if t.dim() != 2 or t.dtype != torch.float32:
        raise ValueError("expected a float32 matrix, got %s %s" % (t.dtype, tuple(t.shape)))
    return t if (t.stride(1) == 1 or t.size(1) <= 1) else t.contiguous()   # (a transposed view: off the tape's own paths)


def mm_raw(a, b, mode, bias=None):
    """a @ b^T ("nt"), a @ b ("nn"), a^T @ b ("tn") on the HIP GEMMs, any widths: operands are zero-padded to what the
    kernels take (contraction: multiples of 8, outputs: multiples of 4; column GEMM tiles of 32 / 64 / 128) by the
    block-copy kernel and the result is cut back by it.  `bias` ("nt" only): added in the epilogue of the first
    contraction pass."""
    a, b = _rows2d(a), _rows2d(b)
    dev = a.device
    if a.size(0) == 0:  # no rows (e.g. a batch without edges): the GEMMs are not launched
        shape = {"nt": (0, b.size(0)), "nn": (0, b.size(1)), "tn": (a.size(1), b.size(1))}[mode]
        return raw_full(shape[0], shape[1], dev, 0.0)
    try:
        return _mm_launch(a, b, mode, bias)
    except Exception as e:
        raise type(e)("%s [mm %s: a %s, b %s]" % (e, mode, tuple(a.shape), tuple(b.shape))) from e


def _mm_launch(a, b, mode, bias):
    dev = a.device
    if mode in ("nt", "nn"):
        R, K = a.shape
        NO = b.size(0) if mode == "nt" else b.size(1)
        Kp, NOp = _up(K, 8), _up(NO, 4)
        a = raw_block(a, 0, R, 0, K, R, Kp)
        if bias is not None and NOp != NO:
            bias = raw_block(bias.view(1, -1), 0, 1, 0, NO, 1, NOp).view(-1)
        y = _empty(R, NOp, dev)
        for c0 in range(0, NOp, 128):  # output slabs of <= 128 columns, written into their columns of y
            c1 = min(c0 + 128, NOp)
            yc = y[:, c0:c1]
            for k0 in range(0, Kp, 256):  # contraction in passes of <= 256, accumulated through the residual operand
                k1 = min(k0 + 256, Kp)
                ak = a if (k0 == 0 and k1 == Kp) else a[:, k0:k1]
                if mode == "nt":   # a [R, K] @ b[NO, K]^T
                    w = raw_block(b, c0, min(c1, NO), k0, min(k1, K), c1 - c0, k1 - k0)
                    ops.linear(ak, w, transB=True, res=yc if k0 else None, K=k1 - k0, NO=c1 - c0, out=yc,
                               bias=bias[c0:c1] if (bias is not None and k0 == 0) else None)
                else:              # a [R, K] @ b[K, NO]
                    w = raw_block(b, k0, min(k1, K), c0, min(c1, NO), k1 - k0, c1 - c0)
                    ops.linear(ak, w, transB=False, res=yc if k0 else None, K=k1 - k0, NO=c1 - c0, out=yc)
        return y if NOp == NO else raw_block(y, 0, R, 0, NO, R, NO)
    R, M, N = a.size(0), a.size(1), b.size(1)  # "tn": a[R, M]^T @ b[R, N], tiled to the column GEMM's 128 x 128 limit
    tile = lambda n: 32 if n <= 32 else (64 if n <= 64 else 128)
    Mp, Np = (_up(M, 128) if M > 128 else tile(M)), (_up(N, 128) if N > 128 else tile(N))
    a, b = raw_block(a, 0, R, 0, M, R, Mp), raw_block(b, 0, R, 0, N, R, Np)
    out = _empty(Mp, Np, dev)
    for m0 in range(0, Mp, 128):
        for n0 in range(0, Np, 128):
            mm_, nn_ = min(128, Mp - m0), min(128, Np - n0)
            ops.linear_wgrad([(a[:, m0:], b[:, n0:], out[m0:, n0:], None)], R, mm_, nn_, lda=Mp, ldb=Np, ldw=Np)
    return out if (Mp == M and Np == N) else raw_block(out, 0, M, 0, N, M, N)


def _raw_unary(kind, x, alpha=1.0, beta=0.0):
    y = torch.empty_like(x)
    call("geossl_tape_unary", kind, ptr(x), x.numel(), float(alpha), float(beta), ptr(y), stream())
    return y


def _raw_binary(op, a, am, b, bm, R, D, scale=1.0):
    y = _empty(R, D, a.device)
    call("geossl_tape_binary", op, ptr(a), am, ptr(b), bm, int(R), int(D), float(scale), ptr(y), stream())
    return y


def _raw_reduce(kind, x):
    R, D = x.shape
    if kind == ROW:
        y, ws = _empty(R, 1, x.device), None
    elif kind == COL:
        y = _empty(1, D, x.device)
        ws = torch.empty(int(call_i64("geossl_tape_colsum_workspace_floats", R, D)), dtype=torch.float32, device=x.device)
    else:
        y, ws = _empty(R // 3, D, x.device), None
    call("geossl_tape_reduce", kind, ptr(x), int(R), int(D), ptr(y), ptr(ws), stream())
    return y


def call_i64(name, *args):
    from . import _lib
    return getattr(_lib.load(), name)(*args)


class Index:
    """Row indices of a gather (int32 or int64, values in [0, nrows)) with, on demand, the incidence lists of its
    adjoint: the positions of each target row in ascending order (torch's stable integer sort, once per index)."""

    def __init__(self, idx, nrows, csr=None):
        assert idx.dtype in (torch.int32, torch.int64)
        self.idx, self.nrows = idx.contiguous(), int(nrows)
        self._csr = csr   # (ptr [nrows + 1] int32 / int64, positions [n] int32) when the caller has the lists already

    @property
    def n(self):
        return self.idx.numel()

    def csr(self):
        if self._csr is None:
            idx = self.idx.long()
            # stable integer sort + a binary search for the list ends: no host read-back (torch.bincount would make one)
            sorted_idx, perm = torch.sort(idx, stable=True)
            ptr_ = torch.searchsorted(sorted_idx, torch.arange(self.nrows + 1, dtype=torch.int64, device=idx.device))
            self._csr = (ptr_, perm.to(torch.int32))
        return self._csr


# ------------------------------------------------------------------------------------------------------------ the tape
class V:
    """A value on the tape: tensor `t` [R, D] fp32, the values it was computed from, and the rule that turns the
    gradient of this value into the gradients of those (`vjp(g, needs) -> tuple`, built from primitives)."""
    __slots__ = ("t", "parents", "vjp", "req")

    def __init__(self, t, parents=(), vjp=None, req=None):
        self.t, self.parents, self.vjp = t, parents, vjp
        self.req = any(p.req for p in parents) if req is None else req

    @property
    def R(self):
        return self.t.size(0)

    @property
    def D(self):
        return self.t.size(1)


class _LazyV(V):
    """A value whose tensor is produced by the deferred column GEMMs of the running backward pass (`deferred_tn`): the
    shape is known, `.t` flushes every pending product in batched launches."""
    __slots__ = ("_t", "shape", "pending")

    def __init__(self, shape, pending, parents=(), vjp=None, req=None):
        self._t, self.shape, self.pending = None, shape, pending
        V.__init__(self, None, parents, vjp, req)

    @property
    def t(self):
        if self._t is None:
            self.pending.flush()
        return self._t

    @t.setter
    def t(self, value):
        self._t = value

    @property
    def R(self):
        return self.shape[0]

    @property
    def D(self):
        return self.shape[1]


class _LazyFn(_LazyV):
    """fn(src.t) of a deferred value, formed when somebody reads it (a column slice of a weight gradient: the gradient of
    a weight that enters the graph zero-padded) - reading it earlier would flush the batch that is still being collected."""
    __slots__ = ("src", "fn")

    @property
    def t(self):
        if self._t is None:
            self._t = self.fn(self.src.t)
            self.src = self.fn = None
        return self._t

    @t.setter
    def t(self, value):
        self._t = value


class _TnNode(_LazyV):
    """a^T @ b (and, with `bias`, the column sums of a) waiting for its batch; a / b already padded to the column GEMM's
    tile widths [R, Mp] / [R, Np] like `_mm_launch` pads them.  `target`: the sum it belongs to."""
    __slots__ = ("a", "b", "key", "bias", "target")


class _SumNode(_LazyV):
    """The sum of deferred products with one shape (contributions to one weight): formed by the batched launches
    themselves - the k-th part of every sum goes into round k, which accumulates onto round k - 1."""
    __slots__ = ("parts",)


class _Deferred:
    """The column GEMMs (`mm(.., "tn")`: gradients of weights) of one backward pass over the tape, collected instead of
    launched: a pass makes ~100 of them one by one - a launch, its reduction of partials and, for a Linear with bias, a
    two-stage column sum each - where a DDM step batches its twenty into one launch.  `flush` groups them by shape and
    issues `geossl_linear_wgrad` batches (up to 32 problems each, bias sums in the same launch); contributions to one
    weight go into successive rounds that accumulate.  Only products whose operands need no padding are deferred."""

    def __init__(self):
        self.nodes = []

    def tn(self, a, b, parents, vjp, bias=False):
        R, M, N = a.t.size(0), a.t.size(1), b.t.size(1)
        tile = lambda n: 32 if n <= 32 else (64 if n <= 64 else 128)
        Mp, Np = (_up(M, 128) if M > 128 else tile(M)), (_up(N, 128) if N > 128 else tile(N))
        n = _TnNode((M, N), self, parents, vjp)
        n.a, n.b = raw_block(_rows2d(a.t), 0, R, 0, M, R, Mp), raw_block(_rows2d(b.t), 0, R, 0, N, R, Np)
        n.key, n.target = (R, Mp, Np), None
        n.bias = _LazyV((1, M), self, (a,), lambda g, needs: (binary(FIRST, g, COL, None, FULL, R, M),)) if bias else None
        self.nodes.append(n)
        return n

    def total(self, x, y):
        """x + y for two deferred values of one shape, without a launch of its own; None if that is not possible."""
        ok = lambda v: isinstance(v, (_TnNode, _SumNode)) and v._t is None and v.pending is self and \
            (not isinstance(v, _TnNode) or v.target is None)
        if not (ok(x) and ok(y)) or x.shape != y.shape:
            return None
        parts = [p for v in (x, y) for p in (v.parts if isinstance(v, _SumNode) else [v])]
        if len({p.key for p in parts}) != 1:
            return None
        s = _SumNode(x.shape, self, (x, y), lambda g, needs: (g, g))
        s.parts = parts
        for p in parts:
            p.target = s
        return s

    def flush(self):
        nodes, self.nodes = self.nodes, []
        groups = {}
        for n in nodes:
            if n._t is None:
                groups.setdefault(n.key, []).append(n)
        for (R, Mp, Np), ns in groups.items():
            dev = ns[0].a.device
            rounds, seen, outs = {}, {}, {}   # round k: the k-th contribution of every sum (free-standing products: round 0)
            for n in sorted(ns, key=lambda n_: n_.bias is None):   # (a product that also yields column sums: an early round)
                tgt = n.target if n.target is not None else n
                k = seen.get(id(tgt), 0)
                seen[id(tgt)] = k + 1
                if id(tgt) not in outs:
                    outs[id(tgt)] = (tgt, _empty(Mp, Np, dev))
                rounds.setdefault(k, []).append((n, outs[id(tgt)][1]))
            for k in sorted(rounds):
                tiles = {}     # (rows, cols) of a tile -> problems: every 128 x 128 tile of every product of the round
                for n, out in rounds[k]:
                    for m0 in range(0, Mp, 128):       # (the column GEMM's 128 x 128 limit, as _mm_launch tiles)
                        for n0 in range(0, Np, 128):
                            db = None
                            if n.bias is not None and n0 == 0:   # (column sums of a: once per column block of a)
                                if n.bias._t is None:            # an accumulating round adds onto db as well: from zero then
                                    n.bias._t = (torch.zeros if k > 0 else torch.empty)(1, Mp, dtype=torch.float32, device=dev)
                                db = n.bias._t[0, m0:]
                            tiles.setdefault((min(128, Mp - m0), min(128, Np - n0)), []).append(
                                (n.a[:, m0:], n.b[:, n0:], out[m0:, n0:], db))
                for (mm_, nn_), probs in tiles.items():
                    ops.linear_wgrad(probs, R, mm_, nn_, accumulate=k > 0, lda=Mp, ldb=Np, ldw=Np)
            for tgt, out in outs.values():
                M, N = tgt.shape
                res = out if (Mp == M and Np == N) else raw_block(out, 0, M, 0, N, M, N)
                tgt._t = res
                if isinstance(tgt, _SumNode):
                    for p_ in tgt.parts:
                        p_._t = res   # (a part read on its own after the sum was formed: not on the tape's paths)
                    tgt.parts = None
            for n in ns:
                if n.bias is not None and n.bias._t is not None and n.bias._t.size(1) != n.bias.shape[1]:
                    n.bias._t = raw_block(n.bias._t, 0, 1, 0, n.bias.shape[1], 1, n.bias.shape[1])
                n.a = n.b = n.target = None   # (operands released; a part -> its sum -> (parents) the part: the cycle cut)


_PENDING = None   # the _Deferred of the backward pass in progress (deferred_tn), else None
_DEFER_MAX_BYTES = 64 << 20


class deferred_tn:
    """with deferred_tn(): the column GEMMs of `grad` calls inside are batched (flushed on exit at the latest)."""

    def __enter__(self):
        global _PENDING
        self.prev, _PENDING = _PENDING, _Deferred()
        return _PENDING

    def __exit__(self, *exc):
        global _PENDING
        mine, _PENDING = _PENDING, self.prev
        if exc[0] is None:
            mine.flush()
        return False


def _tn_deferrable(a, b):
    """A column GEMM that may wait for its batch: a pass is collecting them, both operands are fp32 matrices with rows."""
    if _PENDING is None:
        return False
    ta, tb = a.t, b.t
    # a waiting product keeps its operands alive until the batch runs: only operands of up to 64 MB wait (a launch over
    # more rows than that lasts long enough on its own - the [edges, 3F] and [3 edges, F] operands of PaiNN)
    return (ta.dim() == 2 and tb.dim() == 2 and ta.size(0) > 0 and ta.size(1) > 0 and tb.size(1) > 0
            and ta.dtype == torch.float32 and tb.dtype == torch.float32
            and 4 * ta.size(0) * (ta.size(1) + tb.size(1)) <= _DEFER_MAX_BYTES)


def leaf(t, req=True):
    t = t.detach()
    if t.dim() == 1:
        t = t.view(1, -1)
    if not t.is_contiguous():
        t = raw_block(t, 0, t.size(0), 0, t.size(1), t.size(0), t.size(1)) if t.stride(1) == 1 else t.contiguous()
    return V(t, req=req)


def const(t):
    return leaf(t, req=False)


def _topo(outs):
    """The values reachable from `outs` through values that depend on a leaf with req, parents first."""
    order, seen = [], set()
    stack = [(o, False) for o in outs]
    while stack:
        v, done = stack.pop()
        if done:
            order.append(v)
            continue
        if id(v) in seen:
            continue
        seen.add(id(v))
        stack.append((v, True))
        for p in v.parents:
            if p.req and id(p) not in seen:
                stack.append((p, False))
    return order


def grad(outs, gouts, wrt):
    """d sum_k <outs[k], gouts[k]> / d wrt[i] as values on the tape (None where nothing flows)."""
    wrt_ids = {id(w) for w in wrt}
    order = _topo(outs)
    useful = {}
    for v in order:  # on a path from a wrt value
        useful[id(v)] = id(v) in wrt_ids or any(useful.get(id(p), False) for p in v.parents)
    acc = {}

    def put(v, g):
        have = acc.get(id(v))
        if have is None:
            acc[id(v)] = g
        elif isinstance(g, dict):   # pieces of a split: kept apart until the split node joins them
            for k, gk in g.items():
                have[k] = gk if k not in have else add(have[k], gk)
        else:
            both = _PENDING.total(have, g) if _PENDING is not None else None   # deferred weight gradients: summed by their batch
            acc[id(v)] = both if both is not None else add(have, g)

    for o, g in zip(outs, gouts):
        if g is not None:
            put(o, g)
    for v in reversed(order):
        g = acc.get(id(v))
        if g is None or v.vjp is None or not useful[id(v)]:
            continue
        if id(v) not in wrt_ids:
            del acc[id(v)]
        needs = tuple(p.req and useful.get(id(p), False) for p in v.parents)
        gs = v.vjp(g, needs)
        for p, need, gp in zip(v.parents, needs, gs):
            if need and gp is not None:
                put(p, gp)
    return [acc.get(id(w)) for w in wrt]


# ---------------------------------------------------------------------------------------------------------- primitives
def unary(kind, x, alpha=1.0, beta=0.0):
    def vjp(g, needs):
        if kind == AFFINE:
            return (unary(AFFINE, g, alpha),)
        if kind == LT:
            return (None,)
        if kind == COS:
            return (mul(g, unary(SIN, x, alpha, beta), -alpha),)
        if kind == SQRT:
            return (mul(g, unary(RSQRT, x, alpha, beta), 0.5 * alpha),)
        if kind == RSQRT:
            return (mul(g, unary(RSQRT3, x, alpha, beta), -0.5 * alpha),)
        if kind in (GAUSS, DGAUSS):  # alpha is the coefficient of x^2, the derivative is taken in x
            return (mul(g, unary(kind + 1, x, alpha)),)
        if kind not in _DERIV:
            raise NotImplementedError("no third derivative on the tape for map %d" % kind)
        return (mul(g, unary(_DERIV[kind], x, alpha, beta), alpha),)

    return V(_raw_unary(kind, x.t, alpha, beta), (x,), vjp, req=False if kind == LT else None)


def _mode_shape(mode, R, D):
    return {FULL: (R, D), ROW: (R, 1), COL: (1, D), THIRD: (R // 3, D)}[mode]


def binary(op, a, am, b, bm, R, D, scale=1.0):
    """scale * (A op B) as an [R, D] matrix; a / b are addressed by their modes."""
    assert tuple(a.t.shape) == _mode_shape(am, R, D), (tuple(a.t.shape), am, R, D)
    assert op == FIRST or tuple(b.t.shape) == _mode_shape(bm, R, D), (tuple(b.t.shape), bm, R, D)

    def vjp(g, needs):
        ga = gb = None
        if needs[0]:
            t = binary(MUL, g, FULL, b, bm, R, D, scale) if op == MUL else (g if scale == 1.0 else unary(AFFINE, g, scale))
            ga = reduce_to(am, t)
        if op != FIRST and needs[1]:
            if op == MUL:
                t = binary(MUL, g, FULL, a, am, R, D, scale)
            else:
                s = scale if op == ADD else -scale
                t = g if s == 1.0 else unary(AFFINE, g, s)
            gb = reduce_to(bm, t)
        return (ga, gb) if op != FIRST else (ga,)

    parents = (a,) if op == FIRST else (a, b)
    return V(_raw_binary(op, a.t, am, None if op == FIRST else b.t, bm, R, D, scale), parents, vjp)


def add(a, b):
    return binary(ADD, a, FULL, b, FULL, a.R, a.D)


def sub(a, b):
    return binary(SUB, a, FULL, b, FULL, a.R, a.D)


def mul(a, b, scale=1.0):
    return square(a, scale) if a is b else binary(MUL, a, FULL, b, FULL, a.R, a.D, scale)


def square(x, scale=1.0):
    return V(_raw_binary(MUL, x.t, FULL, x.t, FULL, x.R, x.D, scale), (x,), lambda g, needs: (mul(g, x, 2.0 * scale),))


def reduce(kind, x):
    """Row sums [R, 1], column sums [1, D] or sums over the xyz triple [R / 3, D]: the adjoints of the broadcasts."""
    R, D = x.R, x.D

    def vjp(g, needs):
        return (binary(FIRST, g, kind, None, FULL, R, D),)

    return V(_raw_reduce(kind, x.t), (x,), vjp)


def reduce_to(mode, x):
    return x if mode == FULL else reduce(mode, x)


def gather(src, ix):
    """src[ix] (rows)."""
    D = src.D
    out = _empty(ix.n, D, src.t.device)
    call("geossl_tape_gather_rows", ptr(src.t), ptr(ix.idx), 1 if ix.idx.dtype == torch.int64 else 0, ix.n, D, ptr(out),
         stream())
    return V(out, (src,), lambda g, needs: (scatter(g, ix),))


def scatter(src, ix):
    """zeros(ix.nrows, D).index_add(0, ix, src), summed in ascending position (no atomics)."""
    D = src.D
    ptr_, perm = ix.csr()
    out = _empty(ix.nrows, D, src.t.device)
    call("geossl_tape_scatter_rows", ptr(src.t), ptr(ptr_), 1 if ptr_.dtype == torch.int64 else 0, ptr(perm), ix.nrows, D,
         ptr(out), stream())
    return V(out, (src,), lambda g, needs: (gather(g, ix),))


def slice_cols(x, c0, w):
    R, D = x.R, x.D
    if c0 == 0 and w == D:
        return x
    vjp = lambda g, needs: (pad_cols(g, c0, D),)

    def cut(t):
        out = _empty(R, w, t.device)
        raw_copy2d(t, out, R, w, src_off=c0)
        return out

    if isinstance(x, _LazyV) and x._t is None:   # a deferred weight gradient: sliced when it exists
        node = _LazyFn((R, w), x.pending, (x,), vjp)
        node.src, node.fn = x, cut
        return node
    return V(cut(x.t), (x,), vjp)


def pad_cols(x, c0, D):
    R, w = x.R, x.D
    if c0 == 0 and w == D:
        return x
    out = raw_full(R, D, x.t.device, 0.0)
    raw_copy2d(x.t, out, R, w, dst_off=c0)
    return V(out, (x,), lambda g, needs: (slice_cols(g, c0, w),))


def cat_cols(xs):
    R, widths = xs[0].R, [x.D for x in xs]
    out = _empty(R, sum(widths), xs[0].t.device)
    offs = [sum(widths[:i]) for i in range(len(xs))]
    for x, o, w in zip(xs, offs, widths):
        raw_copy2d(x.t, out, R, w, dst_off=o)
    return V(out, tuple(xs), lambda g, needs: tuple(slice_cols(g, o, w) if n else None for o, w, n in zip(offs, widths, needs)))


def split_cols(x, widths):
    """torch.split(x, widths, dim=1) as values of their own.  The pieces hang off one join node: their gradients are
    collected per piece and written into the columns of ONE matrix (a piece nobody differentiates leaves zeros) - not
    padded to full width one by one and added."""
    R, D = x.R, x.D
    assert sum(widths) == D
    if len(widths) == 1:
        return [x]
    offs = [sum(widths[:i]) for i in range(len(widths))]

    def join(pieces, needs):
        if len(pieces) == len(widths):
            return (cat_cols([pieces[k] for k in range(len(widths))]),)
        out = None
        for k, g in pieces.items():
            t = pad_cols(g, offs[k], D)
            out = t if out is None else add(out, t)
        return (out,)

    hub = V(None, (x,), join)
    outs = []
    for k, (o, w) in enumerate(zip(offs, widths)):
        t = _empty(R, w, x.t.device)
        raw_copy2d(x.t, t, R, w, src_off=o)
        outs.append(V(t, (hub,), (lambda k: lambda g, needs: ({k: g},))(k)))
    return outs


def reshape(x, R, D):
    R0, D0 = x.R, x.D
    if (R, D) == (R0, D0):
        return x
    return V(x.t.view(R, D), (x,), lambda g, needs: (reshape(g, R0, D0),))


def _tn_vjp(a, b):
    """vjp of a^T @ b (the rule `mm` gives its "tn" nodes)."""
    return lambda g, needs: (mm(b, g, "nt") if needs[0] else None, mm(a, g, "nn") if needs[1] else None)


def mm(a, b, mode):
    """The three GEMM forms; the derivative of each is made of the other two."""
    def vjp(g, needs):
        da = db = None
        if mode == "nt":
            da = mm(g, b, "nn") if needs[0] else None
            db = mm(g, a, "tn") if needs[1] else None
        elif mode == "nn":
            da = mm(g, b, "nt") if needs[0] else None
            db = mm(a, g, "tn") if needs[1] else None
        else:
            da = mm(b, g, "nt") if needs[0] else None
            db = mm(a, g, "nn") if needs[1] else None
        return da, db

    if mode == "tn" and _tn_deferrable(a, b):
        return _PENDING.tn(a, b, (a, b), vjp)
    return V(mm_raw(a.t, b.t, mode), (a, b), vjp)


def linear(x, w, b=None):
    """x @ w^T + b, the bias in the GEMM's epilogue; b is a [1, NO] value."""
    if b is None:
        return mm(x, w, "nt")

    def vjp(g, needs):
        dx = mm(g, w, "nn") if needs[0] else None
        if needs[1] and needs[2] and _tn_deferrable(g, x):   # dW and db of one Linear: one problem of the batch
            dw = _PENDING.tn(g, x, (g, x), _tn_vjp(g, x), bias=True)
            return dx, dw, dw.bias
        return dx, (mm(g, x, "tn") if needs[1] else None), (reduce(COL, g) if needs[2] else None)

    return V(mm_raw(x.t, w.t, "nt", bias=b.t.view(-1)), (x, w, b), vjp)


def agg(x, Wf, lay, pair_flag, swap):
    """propagate(aggr="add") of CFConv (schnet.py:190,194-195) in pair-slot form, or its transpose (swap)."""
    def vjp(g, needs):
        return (agg(g, Wf, lay, pair_flag, not swap) if needs[0] else None,
                pairprod(g, x, lay, pair_flag, swap) if needs[1] else None)

    return V(ops.aggregate(x.t, Wf.t, pair_flag, lay, swap=swap), (x, Wf), vjp)


def agg_wide(x, Wf, lay, pair_flag, swap):
    """`agg` at ANY feature width (the reference's num_filters is free, schnet.py:17-30): the aggregation kernels take 32,
    64 or 128 columns, so other widths go through them in zero-padded column slabs of at most 128 (the padding's columns
    aggregate zeros and are cut off again; their gradients are the slices' pads)."""
    D = x.D
    if D in (32, 64, 128):
        return agg(x, Wf, lay, pair_flag, swap)
    outs = []
    for c0 in range(0, D, 128):
        w = min(128, D - c0)
        wp = 32 if w <= 32 else (64 if w <= 64 else 128)
        xs, ws = pad_cols(slice_cols(x, c0, w), 0, wp), pad_cols(slice_cols(Wf, c0, w), 0, wp)
        outs.append(slice_cols(agg(xs, ws, lay, pair_flag, swap), 0, w))
    return outs[0] if len(outs) == 1 else cat_cols(outs)


def pairprod(a, b, lay, pair_flag, swap):
    """out[p] = f0 a[i] b[j] + f1 a[j] b[i] over the pair slots p = (i < j): d aggregate / d filter."""
    def vjp(g, needs):
        return (agg(b, g, lay, pair_flag, swap) if needs[0] else None,
                agg(a, g, lay, pair_flag, not swap) if needs[1] else None)

    return V(ops.pair_product(a.t, b.t, lay, pair_flag, swap), (a, b), vjp)


def ssp(x):
    return unary(SSP, x)


def silu(x):
    return unary(SILU, x)


def _padded_row(vec, width, pad_value):
    """[1, width] constant: vec followed by pad_value."""
    out = raw_full(1, width, vec.device, pad_value)
    v = vec.detach().reshape(1, -1)
    if not v.is_contiguous() or v.dtype != torch.float32:
        raise TypeError("expected a contiguous float32 vector")
    raw_copy2d(v, out, 1, v.size(1))
    return out


FAR = 1.0e4  # a Gaussian centre that no distance comes near: the padded columns of a smearing are exact zeros


# -------------------------------------------------------------------------------------------------------- the backbones
def schnet_atom_features(z, x, lay, cfg, ps):
    """schnet.py:89-101 (embedding .. head) on the tape; `ps` (values) in _core_params order, x = positions [N, 3]."""
    L, Fd, G, cutoff = cfg["L"], cfg["F"], cfg["G"], cfg["cutoff"]
    emb_w, head = ps[0], ps[1 + 9 * L:]
    layers = [ps[1 + 9 * l: 1 + 9 * (l + 1)] for l in range(L)]
    N = x.R
    h = gather(emb_w, Index(z, emb_w.R))                                          # :89
    _, _, pair_flag = ops.pair_geometry(x.t, lay, cutoff)                         # :91 (the graph carries no gradient)
    P = lay.P
    pix = lay.__dict__.get("_tape_pair_index")   # (the incidence lists of the pair slots: once per layout)
    if pix is None:
        pix = lay._tape_pair_index = (Index(lay.pair_i, N), Index(lay.pair_j, N))
    r = sub(gather(x, pix[0]), gather(x, pix[1]))                                 # [P, 3]
    d = unary(SQRT, reduce(ROW, mul(r, r)))                                       # :93 for every pair slot, [P, 1]
    Gp = _up(G, 8)                                                                # contraction width of the row GEMM
    off = const(_padded_row(cfg["offset"], Gp, FAR))
    rbf = unary(GAUSS, binary(SUB, d, ROW, off, COL, P, Gp), float(cfg["coeff"]))  # :205-207, [P, Gp]
    C = unary(AFFINE, unary(COS, d, math.pi / cutoff), 0.5, 0.5)                  # :186
    for lp in layers:
        w1, b1, w2, b2, lin1_w, lin2_w, lin2_b, lin_w, lin_b = lp
        W = linear(ssp(linear(rbf, pad_cols(w1, 0, Gp), b1)), w2, b2)
        W = binary(MUL, W, FULL, C, ROW, P, W.D)                                  # :187
        xl = mm(h, lin1_w, "nt")                                                  # :189
        xl = agg_wide(xl, W, lay, pair_flag, False)                               # :190
        xl = ssp(linear(xl, lin2_w, lin2_b))                                      # :191,165
        h = add(h, linear(xl, lin_w, lin_b))                                      # :166,97
    h = ssp(linear(h, head[0], head[1]))                                          # :99-100
    return linear(h, head[2], head[3])                                            # :101


def painn_atom_features(z, x, idx_i, idx_j, cfg, ps, inc=None):
    """painn.py:230-255 (edge geometry .. last mixing block) on the tape; `ps` in PaiNN._params() order; `inc`: the
    incidence lists of the batch's edge layout ({"i": (ptr, edges), "j": ...}, layout.EdgeLayout.inc) if it has them."""
    Fd, L, cutoff = cfg["F"], cfg["L"], cfg["cutoff"]
    emb_w, fw, fb = ps[0], ps[1], ps[2]
    inter = [ps[3 + 4 * l: 7 + 4 * l] for l in range(L)]
    mix = [ps[3 + 4 * L + 5 * l: 8 + 4 * L + 5 * l] for l in range(L)]
    N, dev = x.R, x.t.device
    inc = inc or {}
    ii, ij = Index(idx_i, N, inc.get("i")), Index(idx_j, N, inc.get("j"))
    E = ii.n
    r = sub(gather(x, ii), gather(x, ij))                                         # :232, [E, 3]
    d = unary(SQRT, reduce(ROW, mul(r, r)))                                       # :236, [E, 1]
    dirv = binary(MUL, r, FULL, unary(RECIP, d), ROW, E, 3)                       # :237
    R = cfg["offsets"].numel()
    Rp = _up(R, 8)
    off = const(_padded_row(cfg["offsets"], Rp, FAR))
    inv_w = const(_raw_unary(RECIP, _padded_row(cfg["widths"], Rp, 1.0)))
    u = binary(MUL, binary(SUB, d, ROW, off, COL, E, Rp), FULL, inv_w, COL, E, Rp, math.sqrt(0.5))
    phi = unary(GAUSS, u, -1.0)                                                   # painn_utils.py:99-102, [E, Rp]
    inside = const(_raw_unary(LT, d.t, cutoff))                                   # :154
    fcut = mul(unary(AFFINE, unary(COS, d, math.pi / cutoff), 0.5, 0.5), inside)  # :152-154
    filters = linear(phi, pad_cols(fw, 0, Rp), fb)
    filters = binary(MUL, filters, FULL, fcut, ROW, E, filters.D)                 # :241, [E, L * 3F]
    # padding_idx = 0 (:247): row 0 of the table receives no gradient - the rows gathered from it carry a zero factor
    # on the way back (a column of ones with a zero for the padding row, gathered like the table)
    keep = raw_full(emb_w.R, 1, dev, 1.0)
    raw_fill(keep[0:1], 0.0)
    zi = Index(z, emb_w.R)
    keep_rows = const(gather(const(keep), zi).t)
    q = _embedding_padded(emb_w, zi, keep_rows)                                   # :247
    mu = const(raw_full(3 * N, Fd, dev, 0.0))                                     # :249, [3N, F]
    dir_flat = reshape(dirv, 3 * E, 1)
    # (shared_filters: ONE filter of width 3F for every interaction, painn.py:178-181,242-243)
    filters = [filters] * L if cfg.get("share_filters") else split_cols(filters, [3 * Fd] * L)
    for l in range(L):
        c0w, c0b, c1w, c1b = inter[l]
        xx = linear(silu(linear(q, c0w, c0b)), c1w, c1b)                          # :53, [N, 3F]
        xx = mul(filters[l], gather(xx, ij))                                      # :54,56, [E, 3F]
        dq_e, dmuR, dmumu = split_cols(xx, [Fd] * 3)                              # :58
        dq = scatter(dq_e, ii)                                                    # :59
        mu_j = reshape(gather(reshape(mu, N, 3 * Fd), ij), 3 * E, Fd)
        dmu = add(binary(MUL, dmuR, THIRD, dir_flat, ROW, 3 * E, Fd),
                  binary(MUL, dmumu, THIRD, mu_j, FULL, 3 * E, Fd))               # :60
        dmu = reshape(scatter(reshape(dmu, E, 3 * Fd), ii), 3 * N, Fd)            # :61
        q, mu = add(q, dq), add(mu, dmu)                                          # :63-64
        i0w, i0b, i1w, i1b, mw = mix[l]
        mu_mix = mm(mu, mw, "nt")                                                 # :100, [3N, 2F]
        mu_V, mu_W = split_cols(mu_mix, [Fd] * 2)                                 # :101
        mu_Vn = unary(SQRT, reduce(THIRD, mul(mu_V, mu_V)), 1.0, cfg["eps"])      # :102
        xx = linear(silu(linear(cat_cols([q, mu_Vn]), i0w, i0b)), i1w, i1b)       # :104-105
        dq_intra, dmu_intra, dqmu_intra = split_cols(xx, [Fd] * 3)                # :107
        q = add(add(q, dq_intra), mul(dqmu_intra, reduce(THIRD, mul(mu_V, mu_W))))   # :110,112
        mu = add(mu, binary(MUL, dmu_intra, THIRD, mu_W, FULL, 3 * N, Fd))        # :108,113
    return q


def _embedding_padded(w, zi, keep_rows):
    out = gather(w, zi)
    R, D = out.R, out.D
    return V(out.t, (w,), lambda g, needs: (scatter(binary(MUL, g, FULL, keep_rows, ROW, R, D), zi),))


# ------------------------------------------------------------------------------------------------------ second order
def second_order(features, mask, need, cot, dhout, pos, params):
    """The derivative of the first-order gradients (d pos, d params given d h), contracted with their cotangents `cot`:
    `features(x, ps)` recorded on the tape, differentiated once with respect to the masked inputs and once more with
    respect to the needed ones.  Returns tensors (or None) for (dhout, pos, *params) where `need`."""
    with torch.no_grad():
        dh = leaf(dhout, need[0])
        x = leaf(pos, True)
        ps = [leaf(p, bool(m or n)) for p, m, n in zip(params, mask[1:], need[2:])]
        h = features(x, ps)
        # first-order gradients only where a cotangent waits for them: the node's mask names every input that requires
        # grad (all parameters), but a force loss differentiates d pos alone - the parameters' first-order gradients
        # were 32 column GEMMs per pass that nothing read (round 6)
        wrt_all = [t for t, m in zip([x] + ps, mask) if m]
        assert len(cot) == len(wrt_all), (len(cot), len(wrt_all))
        wrt = [t for t, c in zip(wrt_all, cot) if c is not None]
        first = grad([h], [dh], wrt) if wrt else []
        outs, gouts = [], []
        for f, c in zip(first, [c for c in cot if c is not None]):
            if f is not None:
                outs.append(f)
                gouts.append(const(c.reshape(f.t.shape) if c.dim() != 2 else c))
        ins = [t for t, n in zip([dh, x] + ps, need) if n]
        if outs and ins:
            # the second pass ends in the parameters: its ~100 weight-gradient products run as a handful of batches
            if _env("GEOSSL_TAPE_NO_DEFER"):   # (A/B: every column GEMM a launch of its own, the round-5 form)
                second = grad(outs, gouts, ins)
            else:
                with deferred_tn():
                    second = grad(outs, gouts, ins)
        else:
            second = [None] * len(ins)
    it = iter(second)
    shapes = [dhout.shape, pos.shape] + [p.shape for p in params]
    return [(lambda v, s: None if v is None else v.t.reshape(s))(next(it), s) if n else None for n, s in zip(need, shapes)]
