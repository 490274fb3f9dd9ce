"""Capacity buckets: the static inputs and index structures behind a step graph that is replayed on batches of ANY
size sequence.

The reference's loader is ``DataLoaderAtomTuple(dataset, batch_size, shuffle=True)`` over ragged molecules
(examples/pretrain_GeoSSL.py:301, Geom3D/dataloaders/dataloaders_AtomTuple.py:81-88; with BFS masking the sizes are
re-drawn every epoch, Geom3D/datasets/datasets_3D.py:24-67): no two batches share a size sequence, so a graph keyed by
the sequence (``pretrain_GeoSSL.structure_fingerprint``) is never replayed there.  A ``Bucket`` fixes what a captured
graph binds - addresses, grids, by-value counts - at a CAPACITY (atoms, pair slots, super-edges, work items of the
aggregation; the number of molecules is the loader's batch size) and turns everything else into device DATA:

* every index structure of the step (``mol_ptr`` / ``pair_ptr`` of the two-view batch, the pair-slot atoms, the
  aggregation's work list, ``se_ptr``, the divisor of NCSN.py:210-212, the incidence lists) lives in static buffers that
  ``fill`` rewrites before a replay: the pointer arrays and the work list are computed on the host from the molecule
  sizes the collation knows (a few cumulative sums over B integers) and go up in ONE pinned copy, the per-slot arrays are
  produced on the device by the layout kernels launched eagerly with the batch's exact counts;
* the real counts sit in ``dims`` (int32, device); the kernels of the captured step are the ``_dyn`` entry points of
  include/geossl_hip.h, which take their grid from the capacity and their row count from ``dims``: rows past the real
  count are never read or written.

Layout of the fused two-view batch in a bucket: ``[view 0 atoms | view 1 atoms | unused]`` - view 1 starts right behind
the REAL atoms of view 0 (``dims[N]``), so the molecule CSR stays contiguous.
"""
import ctypes as C

import numpy as np
import torch
from .switches import env as _env

from . import _lib
from ._lib import call, ptr, stream

MAX_N = 255         # largest molecule a bucket takes (the limit of the aggregation's work list and of the heads)
SMALL_N = 33        # ... and the largest one of the register-form aggregation's size classes: buckets whose molecules
                    # all fit it keep the flat pair-geometry kernel and the ragged layer loop
MAX_N_CLASSES = (SMALL_N, 64, 128, MAX_N)   # a bucket's bound on the molecule size (LDS of the radius-graph kernel)
D_N, D_N2, D_P2, D_S, D_W, D_B, D_N6, D_E2, D_BIG0, D_BIG1 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9   # words of `dims`
DIMS_WORDS = 16
PAINN_MAX_N_CLASSES = (22, 33, 44, 64, 96, 128, MAX_N)   # PaiNN: the bound sizes the LDS of its per-molecule kernels


class DynDims:
    """Device addresses of a bucket's real counts, as the ``dyn_*`` arguments of the ``_dyn`` entry points."""

    def __init__(self, dims):
        base = dims.data_ptr()
        self.tensor = dims
        self.n_atoms = base + 4 * D_N       # atoms of one view (= the row offset of view 1)
        self.n_atoms2 = base + 4 * D_N2     # atoms of the two-view batch
        self.n_pairs2 = base + 4 * D_P2     # pair slots of the two-view batch
        self.n_super = base + 4 * D_S       # super-edges of one view
        self.n_work = base + 4 * D_W        # work items of the aggregation
        self.n_atoms2x3 = base + 4 * D_N6   # PaiNN: rows of the vector features viewed as [3 N2, F]
        self.n_edges2 = base + 4 * D_E2     # PaiNN: edges of the two-view batch
        self.n_big = (base + 4 * D_BIG0, base + 4 * D_BIG1)   # PaiNN: atoms of the molecules above the two stage caps


class _Layout:
    """Duck type of layout.MolLayout for the two-view batch of a bucket (capacity shapes, static buffers)."""

    uniform = False
    order = None
    _sizes_host = None

    big = None
    _warned_cap = False

    def loop_plan(self, *a, **k):
        return (None, 0)   # (the layer loop is a plan for uniform batches: they keep their per-structure graph)

    def big_atoms(self, cap):
        """PaiNN: (static list of the atoms of molecules above `cap` atoms, its capacity, device address of the real
        count) - rewritten per step by Bucket.fill; None for a cap the bucket was not made for."""
        if self.big is None:
            return None
        got = self.big.get(cap)
        if got is None and 0 < cap < self.max_n and not _Layout._warned_cap:
            import warnings
            _Layout._warned_cap = True
            warnings.warn("PaiNN bucket has no atom list for stage cap %d (lists: %s): the molecule-staged kernels are "
                          "skipped for the whole batch (the bucket was made for another radial basis or "
                          "GEOSSL_PAINN_MMA_CAP changed since)" % (cap, sorted(self.big)))
        return got


class _SuperEdges:
    """Duck type of layout.SuperEdgeLayout for a bucket."""


class _Edges:
    """Duck type of layout.EdgeLayout for the two-view batch of a PaiNN bucket: every array is a static buffer at the
    bucket's capacity that geossl_painn_edge_layout rewrites per step; the groups of a molecule end at mol_grp_end."""

    def groups(self, side, mol_ptr=None):
        assert side == "i"
        return (self.row_edge, self.grp_atom, None, self.mol_grp)


def _parts_table():
    from .layout import parts_table
    return parts_table()


_PARTS = None


def max_n_class(hi, prev=None, model_3d="schnet"):
    """The bucket's bound on the molecule size for a batch whose largest molecule has `hi` atoms."""
    classes, exact = (PAINN_MAX_N_CLASSES, 44) if model_3d == "painn" else (MAX_N_CLASSES, SMALL_N)
    # above the sizes whose class selects a faster kernel form (the register aggregation / flat geometry / ragged loop of
    # SchNet, the matrix-pipe interaction of PaiNN) the class only sizes LDS arrays: a quarter of head room there, so that
    # the next batch's largest molecule does not cost another capture
    import os
    want = hi if (hi <= exact or _env("GEOSSL_BUCKET_NO_HEADROOM")) else int(np.ceil(1.25 * hi))
    for c in classes:
        if want <= c and (prev is None or c >= prev):
            return c
    return MAX_N


def sizes_array(batch):
    """The batch's molecule sizes as an int64 array, made once per batch object."""
    arr = batch.__dict__.get("_geossl_sizes_np")
    if arr is None:
        arr = batch.__dict__["_geossl_sizes_np"] = np.asarray(batch._sizes, dtype=np.int64)
    return arr


def batch_counts(sizes, option):
    """(atoms N, pair slots P, super-edges S, aggregation work items W of the two-view batch) of molecules `sizes`."""
    global _PARTS
    if _PARTS is None:
        _PARTS = _parts_table()
    from .layout import aggregate_by_targets, work_items_bound
    n = sizes if isinstance(sizes, np.ndarray) else np.asarray(sizes, dtype=np.int64)
    P = int((n * (n - 1) // 2).sum())
    S = P if option == "combination" else 2 * P
    W = work_items_bound(np.concatenate([n, n]), aggregate_by_targets(2 * len(n)))   # (a bound: the list has 8 padded queues)
    return int(n.sum()), P, S, W


def eligible(batch, model_3d, normalize=False):
    """Can this batch go through a bucket graph?  Molecule sizes known on the host (1 .. 255 atoms, at least one molecule
    with a pair), super_edge_index the extractor's full enumeration - every index tensor of the SchNet step is then a
    function of the sizes; PaiNN: also a collated radius_edge_index on the device (its structures are rebuilt on the
    device per step, geossl_painn_edge_layout)."""
    sizes, canon = getattr(batch, "_sizes", None), getattr(batch, "_canonical", None)
    if (model_3d not in ("schnet", "painn") or normalize or sizes is None or canon not in ("combination", "permutation")
            or not len(sizes)):
        return False
    lo, hi = size_range(batch)
    if lo < 1 or hi > MAX_N or hi < 2:
        return False
    if getattr(batch, "_dataset", None) is not None:   # a handle on a device-resident dataset: gathered by the fill itself
        return model_3d != "painn" or batch.n_edges is not None
    if model_3d == "painn":
        rei = getattr(batch, "radius_edge_index", None)
        if (rei is None or not rei.is_cuda or rei.dtype != torch.long or rei.dim() != 2 or rei.size(0) != 2
                or rei.stride(1) != 1):
            return False
    return tensors_ok(batch)


def tensors_ok(batch):
    """The batch's tensors are what `Bucket.fill` copies by byte count: int64 x [N, c] / batch [N] / super_edge_index
    [2, S], float32 positions [N, 3], all contiguous along what is copied, with N and S the counts the molecule sizes
    give (an int32 x, or a super_edge_index altered after the extractor marked it canonical, would make the copy read past
    the source).  Checked once per (batch object, tensor versions)."""
    x, pos, bv, sei = batch.x, batch.positions, batch.batch, batch.super_edge_index
    from .pretrain_GeoSSL import _tensor_uid   # (lifetime-unique stamps: id() of a freed tensor is handed out again)
    tag = tuple((_tensor_uid(t_), t_._version) for t_ in (x, pos, bv, sei))
    got = batch.__dict__.get("_geossl_tensors_ok")
    if got is not None and got[0] == tag:
        return got[1]
    n = sizes_array(batch)
    N = int(n.sum())
    P = int((n * (n - 1) // 2).sum())
    S = P if batch._canonical == "combination" else 2 * P
    ok = (pos.is_cuda and pos.dtype == torch.float32 and pos.dim() == 2 and pos.size(1) == 3 and pos.is_contiguous()
          and pos.size(0) == N
          and x.is_cuda and x.dtype == torch.long and x.dim() == 2 and x.is_contiguous() and x.size(0) == N
          and bv.is_cuda and bv.dtype == torch.long and bv.dim() == 1 and bv.is_contiguous() and bv.numel() == N
          and sei.is_cuda and sei.dtype == torch.long and sei.dim() == 2 and sei.size(0) == 2 and sei.size(1) == S
          and sei.stride(1) == 1)
    batch.__dict__["_geossl_tensors_ok"] = (tag, ok)
    return ok


MODULE_SWITCHES = ("GEOSSL_NO_CHAIN", "GEOSSL_NCSN_SPLIT_BWD", "GEOSSL_NCSN_SEPARATE_HEADS", "GEOSSL_ARITH_24BIT")  # read by modules_ok
PAINN_SWITCHES = ("GEOSSL_PAINN_NO_CHAIN", "GEOSSL_PAINN_SILU_KERNELS")   # ... for a PaiNN backbone as well


def modules_ok(model, n1, n2):
    """The step of these modules can run on a bucket: the F = 128 chain path of SchNet / PaiNN (the chained row kernel is
    the one that takes a device-side row count) and the paired NCSN heads (two different modules of width 128)."""
    import os
    from .Geom3D.models.painn import PaiNN
    from .Geom3D.models.schnet import SchNet
    from .NCSN import NCSN_version_03, _head_params
    if any(_env(k) for k in MODULE_SWITCHES):
        return False
    if isinstance(model, SchNet):
        if (model.hidden_channels != 128 or model.num_filters != 128 or model.num_interactions < 1 or model.dipole
                or model.atomref is not None or model.mean is not None):
            return False
    elif isinstance(model, PaiNN):
        import torch.nn.functional as F_
        if (model.n_atom_basis != 128 or model.radial_basis.n_rbf not in (8, 16, 20) or model.share_filters
                or model.n_interactions < 1 or model.activation is not F_.silu
                or (model.n_interactions > 1 and model.interactions[0] is model.interactions[1])
                or any(_env(k) for k in PAINN_SWITCHES)):
            return False
    else:
        return False
    if not (isinstance(n1, NCSN_version_03) and isinstance(n2, NCSN_version_03)) or n1 is n2 \
            or n1.emb_dim != 128 or n2.emb_dim != 128:
        return False
    return not ({id(p) for p in _head_params(n1)} & {id(p) for p in _head_params(n2)})


def size_range(batch):
    r = batch.__dict__.get("_geossl_size_range")
    if r is None:
        n = sizes_array(batch)
        r = batch.__dict__["_geossl_size_range"] = (int(n.min()), int(n.max()))
    return r


def is_uniform(batch):
    sizes = getattr(batch, "_sizes", None)
    if sizes is None or not len(sizes):
        return False
    lo, hi = size_range(batch)
    return lo == hi


def _round_up(v, g):
    return int(-(-int(v) // g) * g)


def _slacks(B, sizes=None):
    """Relative head room of a capacity over the first batch's count, for (atoms, pair-slot-like counts): it has to cover
    the spread of a shuffled loader's batch sums, ~ cv / sqrt(B) with cv the relative spread of the per-molecule count -
    three of those standard deviations, from the sizes of the batch at hand when they are given (molecules with hydrogens
    spread twice as much in their pair-slot counts as the 18 +- 4 atoms of set B), never below 1.5 / sqrt(B)."""
    base = min(0.25, max(0.03, 1.5 / np.sqrt(max(B, 1))))
    if sizes is None or len(sizes) < 2:
        return base, base
    n = np.asarray(sizes, dtype=np.float64)
    p = n * (n - 1) / 2
    cv = lambda v: float(v.std() / max(v.mean(), 1e-9))
    f = 3.0 / np.sqrt(max(B, 1))
    return max(base, min(0.4, f * cv(n))), max(base, min(0.4, f * cv(p)))


def capacities(N, P, S, W, B, prev=None, sizes=None):
    """Capacities for a batch with these counts: a slack that covers the spread of a shuffled loader's batch sums
    (`_slacks`), rounded to the kernels' tile sizes; never below a previous bucket's."""
    sn, sp = _slacks(B, sizes)
    if prev is not None:   # a bucket that was outgrown once: the first batch underestimated the spread
        sn, sp = 1.5 * sn, 1.5 * sp
    cap = lambda v, g, sl: _round_up(v * (1.0 + sl) + g, g)
    out = [cap(N, 32, sn), cap(P, 64, sp), cap(S, 64, sp), cap(W, 64, sp)]
    if prev is not None:
        out = [max(a, b) for a, b in zip(out, prev)]
    return tuple(out)


def edge_capacity(E, B, prev=None, sizes=None):
    """Capacity for the edges of a PaiNN batch (one view) with E edges, with the slack of `capacities` for pair-slot-like
    counts (the edges of a molecule lie between its atoms and its pair slots)."""
    slack = _slacks(B, sizes)[1] * (1.5 if prev is not None else 1.0)
    cap = _round_up(E * (1.0 + slack) + 64, 64)
    return cap if prev is None else max(cap, int(prev))


def host_plan(sizes, option):
    """Everything of a batch's index structures that is a function of the molecule sizes alone, as numpy arrays - the part
    of a bucket fill that runs on the host (and is tested without a GPU): counts (N, P, S, W); mol_ptr / pair_ptr of the
    TWO-VIEW batch ([2B + 1], view 1 behind view 0); se_ptr [B + 1]; the aggregation's work list over the 2B molecules
    (largest first, stable; 27 .. 33-atom molecules as 2 or 4 items, larger ones one item per atom: molecule | part << 24); the divisor of NCSN.py:210-212
    (last molecule with a super-edge, + 1); inc_ptr [N + 1] (an atom of an n-atom molecule lies on n - 1 tuples of the
    "combination" enumeration, 2 (n - 1) of "permutation")."""
    global _PARTS
    if _PARTS is None:
        _PARTS = _parts_table()
    n = sizes if isinstance(sizes, np.ndarray) else np.asarray(sizes, dtype=np.int64)
    B = n.shape[0]
    N, P, S, W = batch_counts(n, option)
    mult = 1 if option == "combination" else 2
    mp = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(n, out=mp[1:])
    npair = n * (n - 1) // 2
    pp = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(npair, out=pp[1:])
    n2 = np.concatenate([n, n])
    from .layout import aggregate_by_targets, aggregate_work_list
    work = aggregate_work_list(n2, aggregate_by_targets(2 * B))
    if work.size > W:
        raise ValueError("aggregation work list longer than its bound")   # (work_items_bound: cannot happen)
    has = np.nonzero(npair > 0)[0]
    ip = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.repeat((n - 1) * mult, n), out=ip[1:])
    return dict(counts=(N, P, S, W), mol_ptr2=np.concatenate([mp, mp[1:] + N]), pair_ptr2=np.concatenate([pp, pp[1:] + P]),
                se_ptr=pp * mult, work=work, divisor=int(has[-1]) + 1 if has.size else 0, inc_ptr=ip)


class Bucket:
    """kind "schnet": pair-slot structures of the two-view batch (pointer arrays and work list from the host, pair-slot
    atoms by geossl_pair_index_fill).  kind "painn": the structures of the batch's radius_edge_index for the two-view batch
    (geossl_painn_edge_layout: one launch on the batch's own edge tensor, outputs at the edge capacity E_cap)."""

    def __init__(self, device, B, caps, option, x_cols=2, max_n=SMALL_N, kind="schnet", E_cap=0, n_rbf=20):
        from .pretrain_GeoSSL import Batch
        self.device, self.B, self.option, self.max_n = device, int(B), option, int(max_n)
        self.kind, self.E_cap = kind, int(E_cap)
        self.N_cap, self.P_cap, self.S_cap, self.W_cap = (int(c) for c in caps[:4])
        B, Nc, Pc, Sc, Wc = self.B, self.N_cap, self.P_cap, self.S_cap, self.W_cap
        i32 = dict(dtype=torch.int32, device=device)
        i64 = dict(dtype=torch.int64, device=device)
        # ---- the blob: everything the host computes, uploaded in one copy (int32 words; int64 parts 8-byte aligned)
        o = {"dims": 0}
        o["mol_ptr"] = DIMS_WORDS
        o["pair_ptr"] = o["mol_ptr"] + 2 * B + 1
        o["se_ptr"] = o["pair_ptr"] + 2 * B + 1
        o["work"] = o["se_ptr"] + B + 1
        o["stats"] = _round_up(o["work"] + Wc, 2)
        o["inc_ptr"] = o["stats"] + 4
        o["big0"] = o["inc_ptr"] + 2 * (Nc + 1)
        # PaiNN, molecules above the stage caps of the molecule-staged interaction kernels: their atoms (two lists)
        self.big_caps = ()
        if kind == "painn" and self.max_n > 0:
            from .layout import painn_stage_caps
            # (the stage caps depend on the radial basis: 73 / 74 / 75 atoms for the backward at 20 / 16 / 8 functions - lists
            # made for another basis would never be found and the whole batch would fall back to the per-atom kernels)
            self.big_caps = tuple(c for c in painn_stage_caps(128, int(n_rbf)) if 0 < c < self.max_n)
        o["big1"] = o["big0"] + (2 * Nc if len(self.big_caps) > 0 else 0)
        # a batch drawn from a device-resident dataset (Geom3D.dataloaders.DeviceDataset): where its molecules' atoms and
        # (PaiNN) radius edges start in the dataset, and the edge offsets of the batch
        o["src_off"] = o["big1"] + (2 * Nc if len(self.big_caps) > 1 else 0)
        o["e_src_off"] = o["src_off"] + B
        o["e_ptr"] = o["e_src_off"] + (B if kind == "painn" else 0)
        self.words = o["e_ptr"] + (B + 1 if kind == "painn" else 0)
        self.off = o
        self.blob = torch.zeros(self.words, **i32)
        self.dims = self.blob[0:DIMS_WORDS]
        self.dyn = DynDims(self.dims)
        self._host = [[torch.zeros(self.words, dtype=torch.int32).pin_memory(), None] for _ in range(3)]
        self._slot = 0
        # ---- static inputs of the step
        self.x = torch.zeros(Nc, x_cols, **i64)
        self.positions = torch.zeros(Nc, 3, dtype=torch.float32, device=device)
        self.batch_vec = torch.zeros(Nc, **i64)
        self.sei = torch.zeros(2, Sc, **i64)
        self.b2 = torch.zeros(2 * Nc, **i64)   # placeholder for the backbone's `batch` argument (the layout is passed)
        # ---- two-view molecule layout
        lay = _Layout()
        lay.N, lay.B, lay.P, lay.max_n = 2 * Nc, 2 * B, 2 * Pc, self.max_n
        lay.mol_ptr = self.blob[o["mol_ptr"]:o["mol_ptr"] + 2 * B + 1]
        lay.pair_ptr = self.blob[o["pair_ptr"]:o["pair_ptr"] + 2 * B + 1]
        lay.device, lay.dyn = device, self.dyn
        lay._batch_version = self.b2._version
        lay.agg_work = None
        if self.big_caps:
            lay.big = {c: (self.blob[o["big%d" % k]:o["big%d" % k] + 2 * Nc], 2 * Nc, self.dyn.n_big[k])
                       for k, c in enumerate(self.big_caps)}
        if kind == "schnet":
            lay.pair_i = torch.zeros(2 * Pc, **i32)
            lay.pair_j = torch.zeros(2 * Pc, **i32)
            lay.agg_work = self.blob[o["work"]:o["work"] + Wc]
            from .layout import aggregate_by_targets
            lay.agg_targets = aggregate_by_targets(2 * B)
        self.lay2 = lay
        # ---- PaiNN: edge structures of the two-view batch
        self.el = None
        if kind == "painn":
            Ec = self.E_cap
            el = _Edges()
            el.E, el.N, el.B = 2 * Ec, 2 * Nc, 2 * B
            el.idx_i, el.idx_j = torch.zeros(2 * Ec, **i64), torch.zeros(2 * Ec, **i64)
            el.inc = {"i": (torch.zeros(2 * Nc + 1, **i64), torch.zeros(max(2 * Ec, 1), **i32)),
                      "j": (torch.zeros(2 * Nc + 1, **i64), torch.zeros(max(2 * Ec, 1), **i32))}
            G = int(_lib.load().geossl_painn_group_capacity(2 * Ec, 2 * Nc))
            el.row_edge = torch.full((4 * G,), -1, **i32)
            el.grp_atom = torch.full((G,), -1, **i32)
            el.mol_grp = torch.zeros(2 * B + 1, **i32)
            el.mol_grp_end = torch.zeros(2 * B, **i32)
            # set by geossl_painn_edge_layout on an edge that leaves its molecule (the reference's collated
            # radius_edge_index never has one); read without draining the stream, a few steps late (_lib.StatusWord)
            self.el_status = _lib.StatusWord(device, "radius_edge_index must be grouped by molecule in batch order with both "
                                             "ends in the same molecule (collated MoleculeDataset3DRadius output is)")
            el.status = self.el_status.word
            el.dyn = self.dyn
            self.el = el
            self.e2 = torch.zeros(2, 1, **i64)   # placeholder for PaiNN.forward's radius_edge_index argument
            self.rei = None                       # the collated edges of a batch drawn from a dataset (made on first use)
        # ---- super-edge bookkeeping of the heads
        sel = _SuperEdges()
        sel.sei0, sel.sei1, sel.batch = self.sei[0], self.sei[1], self.batch_vec
        sel.S, sel.N, sel.B = Sc, Nc, B
        sel.se_ptr = self.blob[o["se_ptr"]:o["se_ptr"] + B + 1]
        sel.stats = self.blob[o["stats"]:o["stats"] + 4].view(torch.int64)
        sel.inc_ptr = self.blob[o["inc_ptr"]:o["inc_ptr"] + 2 * (Nc + 1)].view(torch.int64)
        sel.inc_idx = torch.zeros(2 * Sc, **i32)
        sel.dyn = self.dyn
        sel._versions = (self.batch_vec._version, self.sei._version)
        self.sel = sel
        # ---- the batch object the captured step sees
        self.batch = Batch(self.x, self.positions, self.batch_vec, self.sei, None, B, None, option)
        self.batch._bucket = self
        self.real = None  # (N, P, S, W) of the batch last filled in

    def caps(self):
        return (self.N_cap, self.P_cap, self.S_cap, self.W_cap)

    def fits(self, counts, hi=None, E=None):
        """counts = (N, P, S, W) of a batch; hi: its largest molecule; E: its edges (PaiNN)."""
        return (all(c <= cap for c, cap in zip(counts, self.caps())) and (hi is None or hi <= self.max_n)
                and (E is None or self.kind != "painn" or E <= self.E_cap))

    def fill(self, batch, counts=None, zero=None):
        """The batch's atom types, positions, index tensors and derived structures into the static buffers; `zero`: a
        float32 buffer cleared by the same launch (the owner's flat gradient buffer).  `batch`: a collated batch on the
        device, or a handle on a device-resident dataset (Geom3D.dataloaders.DatasetBatch) whose molecules are gathered
        from there.  One pinned upload (everything that is a function of the molecule sizes) + one launch
        (geossl_gather_molecules: atom rows, batch vector, super-edges, pair-slot atoms, incidence lists, radius edges,
        the cleared buffer) [+ geossl_painn_edge_layout]."""
        global _PARTS
        if _PARTS is None:
            _PARTS = _parts_table()
        B, o = self.B, self.off
        n = sizes_array(batch)
        if n.shape[0] != B:
            raise ValueError("bucket of %d molecules got a batch of %d" % (B, n.shape[0]))
        N, P, S, W = counts if counts is not None else batch_counts(n, self.option)
        ds = getattr(batch, "_dataset", None)
        rei, E = None, 0
        if ds is not None:
            if ds.option != self.option or ds.x_cols != self.x.size(1) or ds.device != self.x.device:
                raise ValueError("dataset and bucket disagree (tuple option, x columns or device)")
            if self.kind == "painn":
                if batch.n_edges is None:
                    raise ValueError("PaiNN bucket fill expects a dataset built with radius=...")
                E = batch.n_edges
        elif self.kind == "painn":
            rei = batch.radius_edge_index
            if (rei is None or not rei.is_cuda or rei.dtype != torch.long or rei.dim() != 2 or rei.size(0) != 2
                    or rei.stride(1) != 1):
                raise ValueError("PaiNN bucket fill expects a collated int64 radius_edge_index [2, E] on the device")
            E = int(rei.size(1))
        if not self.fits((N, P, S, W), int(n.max()), E) or P < 1:
            raise ValueError("batch exceeds the bucket's capacity")
        if ds is None and (not tensors_ok(batch) or batch.x.size(1) != self.x.size(1)):
            raise ValueError("bucket fill expects contiguous collated int64 / float32 tensors of the sizes' shapes")
        slot = self._host[self._slot]
        self._slot = (self._slot + 1) % len(self._host)
        if slot[1] is not None:
            slot[1].synchronize()   # the upload that last read this staging buffer (three steps ago)
        h = slot[0].numpy()
        hp = host_plan(n, self.option)
        Wr = hp["work"].size          # (the list's real length: 8 queues; <= the bound W the capacity was checked with)
        h[0:8] = (N, 2 * N, 2 * P, S, Wr, B, 6 * N, 2 * E)
        if self.big_caps:
            from .layout import big_atom_list
            n2 = np.concatenate([n, n])
            for k, c in enumerate(self.big_caps):
                idx = big_atom_list(n2, c)
                h[8 + k] = idx.size
                h[o["big%d" % k]:o["big%d" % k] + idx.size] = idx
        h[o["mol_ptr"]:o["mol_ptr"] + 2 * B + 1] = hp["mol_ptr2"]
        h[o["pair_ptr"]:o["pair_ptr"] + 2 * B + 1] = hp["pair_ptr2"]
        h[o["se_ptr"]:o["se_ptr"] + B + 1] = hp["se_ptr"]
        if self.kind == "schnet":
            h[o["work"]:o["work"] + Wr] = hp["work"]
        st = h[o["stats"]:o["stats"] + 4].view(np.int64)
        st[0], st[1] = hp["divisor"], 0
        h[o["inc_ptr"]:o["inc_ptr"] + 2 * (N + 1)].view(np.int64)[:] = hp["inc_ptr"]
        if ds is not None:
            h[o["src_off"]:o["src_off"] + B] = ds.off[batch.ids]
            if self.kind == "painn":
                h[o["e_src_off"]:o["e_src_off"] + B] = ds.edge_off[batch.ids]
                ep = h[o["e_ptr"]:o["e_ptr"] + B + 1]
                ep[0] = 0
                np.cumsum(ds.edge_cnt[batch.ids], out=ep[1:])
        self.blob.copy_(slot[0], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        # ---- device side: one launch
        base = self.blob.data_ptr()
        lay, sel = self.lay2, self.sel
        g = _lib.Gather()
        g.option, g.x_cols = (0 if self.option == "combination" else 1), self.x.size(1)
        g.mol_ptr, g.se_ptr = base + 4 * o["mol_ptr"], base + 4 * o["se_ptr"]
        g.x_dst, g.pos_dst, g.batch_dst = ptr(self.x), ptr(self.positions), ptr(self.batch_vec)
        g.sei0, g.sei1 = ptr(self.sei[0]), ptr(self.sei[1])
        if ds is not None:
            g.x_src, g.pos_src, g.src_off = ptr(ds.x), ptr(ds.positions), base + 4 * o["src_off"]
        else:   # a collated batch: its molecules start where the bucket's do (the extractor's enumeration is generated)
            g.x_src, g.pos_src, g.src_off = ptr(batch.x), ptr(batch.positions), g.mol_ptr
        if self.kind == "schnet":
            g.pair_ptr2, g.pair_i, g.pair_j = base + 4 * o["pair_ptr"], ptr(lay.pair_i), ptr(lay.pair_j)
        g.inc_ptr, g.inc_idx = ptr(sel.inc_ptr), ptr(sel.inc_idx)
        if ds is not None and self.kind == "painn":
            if self.rei is None:
                self.rei = torch.zeros(2, max(self.E_cap, 1), dtype=torch.int64, device=self.device)
            rei = self.rei
            if E:
                g.e0_src, g.e1_src = ptr(ds.edges[0]), ptr(ds.edges[1])
                g.e_src_off, g.e_ptr = base + 4 * o["e_src_off"], base + 4 * o["e_ptr"]
                g.e0_dst, g.e1_dst = ptr(rei[0]), ptr(rei[1])
        if zero is not None:
            g.zero, g.zero_count = ptr(zero), zero.numel()
        st_ = stream()
        call("geossl_gather_molecules", C.byref(g), B, st_)
        if self.kind == "painn":
            el = self.el
            try:
                self.el_status.poll()
            except IndexError as e:
                raise ValueError(str(e)) from None
            call("geossl_painn_edge_layout", ptr(rei[0]), ptr(rei[1]), E, ptr(lay.mol_ptr), N, B, 2 * self.N_cap,
                 ptr(el.idx_i), ptr(el.idx_j), ptr(el.inc["i"][0]), ptr(el.inc["i"][1]), ptr(el.inc["j"][0]),
                 ptr(el.inc["j"][1]), ptr(el.row_edge), ptr(el.grp_atom), ptr(el.mol_grp), ptr(el.mol_grp_end),
                 ptr(el.status), st_)
            self.el_status.arm(every=8)
        self.real = (N, P, S, W)
        self.real_E = E
        return self.real
