"""Position-independent index structures of a collated batch, built on the GPU once per batch.

Host-side mirror of what the reference gets from ``BatchAtomTuple.from_data_list``
(Geom3D/dataloaders/dataloaders_AtomTuple.py:46-78): ``batch`` is sorted, index tensors carry
node offsets, ``num_graphs = batch[-1] + 1``.
"""
import os

import numpy as np
import torch
from .switches import env as _env

from . import _lib
from ._lib import call, ptr, stream


class MolLayout:
    """mol_ptr / pair-slot enumeration of a sorted ``batch`` vector.

    pair slots: per molecule the n(n-1)/2 pairs (i<j) in lexicographic order — the enumeration of
    AtomTupleExtractor 'combination' (dataloaders_AtomTuple.py:22-23); slot p joins atoms
    pair_i[p] < pair_j[p].
    """

    def __init__(self, batch, num_graphs=None, sizes=None):
        """``sizes``: atoms per molecule as HOST integers, when the caller collated the batch itself (a loader does):
        the two scalars the allocation below needs (largest molecule, number of pair slots) then come from the host
        and nothing waits for the device — without it they are read back from the layout kernel, which drains the
        stream once per new batch."""
        _lib.require_cuda(batch)
        if batch.dtype != torch.long or batch.dim() != 1:
            raise ValueError("batch must be a 1-D int64 tensor")
        batch = batch.contiguous()
        self.N = int(batch.numel())
        dev = batch.device
        if sizes is not None:
            sizes = [int(n) for n in sizes]
            if sum(sizes) != self.N or (num_graphs is not None and int(num_graphs) != len(sizes)):
                raise ValueError("sizes do not match the batch vector")
            self.B = len(sizes)
        elif self.N == 0:
            self.B = int(num_graphs or 0)
        else:
            self.B = int(num_graphs) if num_graphs is not None else int(batch[-1].item()) + 1  # :75-78
        B = self.B
        self.mol_ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        self.pair_ptr = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        stats = torch.zeros(4, dtype=torch.int64, device=dev)
        if B > 0:
            call("geossl_layout_build", ptr(batch), self.N, B, ptr(self.mol_ptr), ptr(self.pair_ptr), ptr(stats),
                 stream())
        if sizes is not None:
            max_n, P = max(sizes, default=0), sum(n * (n - 1) // 2 for n in sizes)
        else:
            max_n, P, bad, _ = stats.tolist()
            if bad:
                raise ValueError("batch vector must be sorted ascending with ids in [0, num_graphs) "
                                 "(collated batches are; dataloaders_AtomTuple.py:61,72)")
        self.max_n, self.P = int(max_n), int(P)
        self.pair_i = torch.empty(self.P, dtype=torch.int32, device=dev)
        self.pair_j = torch.empty(self.P, dtype=torch.int32, device=dev)
        if self.P > 0:
            call("geossl_pair_index_fill", ptr(self.mol_ptr), ptr(self.pair_ptr), B, ptr(self.pair_i),
                 ptr(self.pair_j), stream())
        # molecules by descending size: the sequence in which the per-molecule blocks of the aggregation are started
        # (ops.aggregate); identity when all sizes are equal
        self.order = None
        if B > 1:
            nat = self.mol_ptr[1:] - self.mol_ptr[:-1]
            self.order = torch.argsort(nat, descending=True, stable=True).to(torch.int32)
        # ragged batches built from host sizes: the work list of the aggregation (geossl_cfconv_aggregate_work), molecules
        # by descending size, the 21..33-atom ones as 2 or 4 work items (one group of target atoms each)
        # (molecules above 33 atoms - Molecule3D with hydrogens - as 16 items each: lists of target atoms, no size class)
        self.agg_work, self.agg_targets = None, False
        if sizes is not None and 20 < self.max_n <= 255 and B < (1 << 24) and not _env("GEOSSL_AGG_NO_SPLIT"):
            self.agg_targets = aggregate_by_targets(B)
            self.agg_work = torch.from_numpy(aggregate_work_list(np.asarray(sizes, dtype=np.int64), self.agg_targets)).to(dev)
        self.device = dev
        self._batch_version = batch._version
        self._sizes_host = sizes
        self._loop_plan = None
        self.uniform = False
        self.dyn = None  # bucket.DynDims when the layout belongs to a capacity bucket (device-side row counts)

    def big_atoms(self, cap):
        """The atoms of the molecules with more than `cap` atoms (PaiNN: the molecules a molecule-staged interaction
        launch skips, covered by the per-atom kernels) -> (int32 tensor, entries, None) - cached; None when the molecule
        sizes are not known on the host (the caller then keeps one kernel form for the whole batch) or a graph capture is
        open and the list was not made before it."""
        got = self._big.get(cap) if hasattr(self, "_big") else None
        if got is None:
            if self._sizes_host is None or torch.cuda.is_current_stream_capturing():
                return None
            if not hasattr(self, "_big"):
                self._big = {}
            idx = big_atom_list(np.asarray(self._sizes_host, dtype=np.int64), cap)
            t_ = torch.from_numpy(idx if idx.size else np.zeros(1, np.int32)).to(self.device)
            got = self._big[cap] = (t_, int(idx.size), None)
        return got

    def loop_plan(self, max_rows=96, max_mols=None):
        """Blocks of the layer loop (geossl_schnet_layer_loop): consecutive molecules of a uniform batch in blocks of at
        most `max_rows` atom rows -> (int32 tensor [nblocks, 4] = first row, end row, first molecule, end molecule;
        nblocks), or (None, 0) when the batch has no such plan or the layout was not built from host sizes (a loader that
        knows them passes them: prepare_batch, BatchAtomTuple.from_sizes / from_data_list)."""
        if self._loop_plan is None:
            sizes = self._sizes_host
            if sizes is not None and torch.cuda.is_current_stream_capturing():
                return (None, 0)  # (no upload inside a capture; not remembered)
            if sizes is None:
                # a layout built from the batch vector alone: the sizes are on the device only, and reading them back
                # would drain the stream once per new batch (0.3 ms measured on the forward-only line) - no plan, the
                # caller launches the operations one by one
                self._loop_plan = (None, 0)
                return self._loop_plan
            self.uniform = len(sizes) > 0 and len(set(sizes)) == 1
            plan = loop_block_plan(sizes, max_rows, max_mols)
            if plan is None:
                self._loop_plan = (None, 0)
                return self._loop_plan
            t_ = torch.from_numpy(np.ascontiguousarray(plan)).to(self.device)
            self._loop_plan = (t_, len(plan))
        return self._loop_plan


def painn_stage_caps(F=128, R=20):
    """(forward, backward) molecule sizes up to which PaiNN's molecule-staged interaction kernels take a molecule when a
    layout holds larger ones (the rest goes to the per-atom kernels, atom list by atom list): the LDS limits of the
    library (geossl_painn_stage_cap); 0 = no split (one kernel form for the whole batch).  The forward's is 0 unless
    GEOSSL_PAINN_MMA_CAP asks for one (A/B runs)."""
    if _env("GEOSSL_PAINN_NO_SPLIT"):   # (A/B runs: one kernel form for the whole batch, the round-4 behaviour)
        return 0, 0
    lib = _lib.load()
    cf, cb = int(lib.geossl_painn_stage_cap(0, F, R)), int(lib.geossl_painn_stage_cap(2, F, R))
    # The FORWARD split is off unless asked for: measured on set C (molecules with hydrogens), the matrix-pipe forward with
    # its LDS sized for 44-atom molecules (one block of four waves per CU) plus a per-atom pass for the rest is slower
    # than the per-atom kernel for the whole batch (bs = 128: 2.24 against 2.09 ms per step; bs = 1024: 10.6 against 10.6)
    # - tools/experiments/painn_forms.py.  The backward split pays (the molecule-staged backward up to 73 atoms).
    env = _env("GEOSSL_PAINN_MMA_CAP")
    cf = min(cf, int(env)) if env is not None else 0
    return cf, cb


def big_atom_list(n, cap):
    """Atom indices (int32, ascending) of the molecules with more than `cap` atoms, for molecules of sizes `n` laid out
    back to back."""
    start = np.concatenate([[0], np.cumsum(n)[:-1]]) if len(n) else np.zeros(0, np.int64)
    sel = n > cap
    k = n[sel]
    if not k.size:
        return np.zeros(0, dtype=np.int32)
    ends = np.cumsum(k)
    return (np.repeat(start[sel], k) + (np.arange(int(ends[-1]), dtype=np.int64) - np.repeat(ends - k, k))).astype(np.int32)


_PARTS = None


def parts_table():
    """geossl_aggregate_parts(n) for n = 0 .. 255 (work items of an n-atom molecule in geossl_cfconv_aggregate_work)."""
    global _PARTS
    if _PARTS is None:
        lib = _lib.load()
        _PARTS = np.array([lib.geossl_aggregate_parts(k) for k in range(256)], dtype=np.int64)
    return _PARTS


def aggregate_by_targets(num_mols):
    """Small launches (at most GEOSSL_AGG_TARGETS_MAX molecules, default 256: the two views of the reference's batch size)
    aggregate with one work item per TARGET atom (geossl_cfconv_aggregate_targets_dyn): such a launch is bound by its
    longest walk, not by bytes.  Measured per DDM step (tools/experiments/agg_targets.sh): set C at 10 A, 128 molecules
    per view 1.283 against 1.316 ms, set B without the ragged layer loop 0.806 against 0.847 ms; at 256 molecules per view
    the doubled filter-row traffic already loses (set C 2.106 against 2.065 ms, set B 1.244 against 1.106 ms)."""
    return num_mols <= int(_env("GEOSSL_AGG_TARGETS_MAX", 256))


XCD = 8   # L2 domains of the chip: workgroups are dealt to them round-robin


def work_items_bound(n, targets=False):
    """Upper bound on the entries of aggregate_work_list(n, targets) (8 queues of equal length), from two sums - what a
    capacity check needs per step without building the list."""
    if not len(n):
        return 0
    parts = n if targets else parts_table()[n]
    return XCD * (-(-int(parts.sum()) // XCD) + int(parts.max()))


_QUEUE_GRID = {}


def aggregate_work_list(n, targets=False):
    """Work list of geossl_cfconv_aggregate_work for molecules of `n` atoms (int64 array), as int32 words: every
    molecule as geossl_aggregate_parts(size) items molecule | part << 24 (targets: one item per atom, molecule | atom << 24,
    for geossl_cfconv_aggregate_targets_dyn), in EIGHT queues of equal length (padded with -1; queue k = entries
    [k Q, (k + 1) Q)): the molecules are dealt to the queues largest first in snake order (balanced sums), the items of a
    molecule are consecutive entries of its queue - workgroup b of the launch takes entry b / 8 of queue b mod 8, i.e.
    runs on XCD b mod 8: items that read each other's filter rows share an L2."""
    M = len(n)
    if M == 0:
        return np.empty(0, dtype=np.int32)
    idx = np.argsort(-n, kind="stable")
    parts_sorted = n[idx] if targets else parts_table()[n][idx]
    grid = _QUEUE_GRID.get(M)
    if grid is None:   # rank of queue k's t-th molecule (snake order), a function of the molecule count alone
        t = np.arange(-(-M // XCD), dtype=np.int64)[None, :]
        k = np.arange(XCD, dtype=np.int64)[:, None]
        rank = t * XCD + np.where(t % 2 == 0, k, XCD - 1 - k)
        valid = rank < M
        if len(_QUEUE_GRID) > 64:
            _QUEUE_GRID.clear()
        grid = _QUEUE_GRID[M] = (np.minimum(rank, M - 1), valid)
    rank, valid = grid
    pq = np.where(valid, parts_sorted[rank], 0)                 # [XCD, T]: items of each molecule, queue-major
    cnt = pq.sum(axis=1)
    flat_parts = pq[valid]                                      # queue by queue, rank order inside a queue
    mol = np.repeat(idx[rank[valid]], flat_parts)
    total = mol.size
    ends = np.cumsum(flat_parts)
    part = np.arange(total, dtype=np.int64) - np.repeat(ends - flat_parts, flat_parts)
    words = (mol | (part << 24)).astype(np.uint32).view(np.int32)
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    out = np.full((XCD, int(cnt.max())), -1, dtype=np.int32)
    out[np.repeat(np.arange(XCD), cnt), np.arange(total, dtype=np.int64) - np.repeat(starts, cnt)] = words
    return out.reshape(-1)


def loop_block_plan(sizes, max_rows=96, max_mols=None, slots=512):
    """Block plan of the layer loop (geossl_schnet_layer_loop) for molecules of `sizes` atoms: int32 array [nblocks, 4] =
    (first row, end row, first molecule, end molecule), or None when the batch has no plan.

    Uniform batches only (ops.layer_loop).  ONE round of blocks: the blocks of a second round would start when the first
    round ends, and a block runs for the whole pass - 515 blocks on the chip's 512 slots (two blocks of four waves per
    CU) would double the launch.  So: the fewest molecules per block (one per wave is the target, fewer for a small batch)
    with which the batch fits `slots` blocks of at most `max_rows` rows; None if there is none (the caller launches the
    operations one by one)."""
    B = len(sizes)
    if B == 0 or len(set(sizes)) != 1 or sizes[0] > max_rows or sizes[0] < 1:
        return None
    n = int(sizes[0])
    if max_mols is None:
        max_mols = max(1, -(-B // slots))
    if max_mols * n > max_rows:
        return None
    ptr_ = np.arange(B + 1, dtype=np.int64) * n
    m0s = np.arange(0, B, max_mols, dtype=np.int64)
    m1s = np.minimum(m0s + max_mols, B)
    return np.stack([ptr_[m0s], ptr_[m1s], m0s, m1s], axis=1).astype(np.int32)


def get_layout(batch):
    """Layout cached on the batch tensor object itself (the same object is passed for both views
    and both heads of a DDM step)."""
    lay = getattr(batch, "_geossl_layout", None)
    hs = getattr(batch, "_geossl_sizes", None)  # (host sizes, tensor version) left by prepare_batch
    sizes = hs[0] if hs is not None and hs[1] == batch._version and sum(hs[0]) == batch.numel() else None
    if lay is None or lay._batch_version != batch._version or lay.N != batch.numel():
        lay = MolLayout(batch, sizes=sizes)
        batch._geossl_layout = lay
    elif sizes is not None and lay._sizes_host is None and len(sizes) == lay.B:
        lay._sizes_host, lay._loop_plan = sizes, None  # a layout built before the collation's sizes were attached
    return lay


class SuperEdgeLayout:
    """Per-batch bookkeeping of ``super_edge_index`` for the DDM head: first super-edge of every
    molecule, the divisor ``max(edge2graph)+1`` of NCSN.py:210-212 (kept on the device), and the
    atom -> incident super-edge lists used to reduce d loss / d node_feature without atomics."""

    def __init__(self, batch, super_edge_index, num_graphs, validate=True):
        _lib.require_cuda(batch, super_edge_index)
        sei = super_edge_index
        if sei.dtype != torch.long or sei.dim() != 2 or sei.size(0) != 2:
            raise ValueError("super_edge_index must be int64 [2, S]")
        self.sei0 = sei[0].contiguous()
        self.sei1 = sei[1].contiguous()
        self.batch = batch.contiguous()
        self.S = int(sei.size(1))
        self.N = int(batch.numel())
        self.B = int(num_graphs)
        dev = batch.device
        self.se_ptr = torch.zeros(self.B + 1, dtype=torch.int32, device=dev)
        self.stats = torch.zeros(2, dtype=torch.int64, device=dev)
        call("geossl_super_edge_ptr", ptr(self.batch), ptr(self.sei0), ptr(self.sei1), self.S, self.B,
             ptr(self.se_ptr), ptr(self.stats), stream())
        inc_cnt = torch.zeros(self.N, dtype=torch.int32, device=dev)
        call("geossl_incidence_count", ptr(self.batch), ptr(self.sei0), ptr(self.sei1), ptr(self.se_ptr), self.N, 3,
             ptr(inc_cnt), stream())
        self.inc_ptr = torch.zeros(self.N + 1, dtype=torch.int64, device=dev)
        self.inc_ptr[1:] = torch.cumsum(inc_cnt, 0, dtype=torch.int64)
        self.inc_idx = torch.empty(2 * self.S, dtype=torch.int32, device=dev)
        call("geossl_incidence_fill", ptr(self.batch), ptr(self.sei0), ptr(self.sei1), ptr(self.se_ptr), self.N, 3,
             ptr(self.inc_ptr), ptr(self.inc_idx), stream())
        if validate and int(self.stats[1].item()):
            raise ValueError("super_edge_index must be grouped by molecule in batch order with both ends in "
                             "the same molecule (collated AtomTupleExtractor output is)")
        self._versions = (batch._version, sei._version)
        self.dyn = None


def get_super_edge_layout(batch, super_edge_index, num_graphs, validate=True):
    lay = getattr(super_edge_index, "_geossl_layout", None)
    # (collated AtomTupleExtractor output, marked by prepare_batch: grouped by molecule by construction - no check, no sync)
    if getattr(super_edge_index, "_geossl_grouped", None) == (batch._version, super_edge_index._version):
        validate = False
    if (lay is None or lay._versions != (batch._version, super_edge_index._version)
            or lay.S != super_edge_index.size(1) or lay.N != batch.numel()):
        lay = SuperEdgeLayout(batch, super_edge_index, num_graphs, validate=validate)
        super_edge_index._geossl_layout = lay
    return lay


def prepare_batch(batch_vec, super_edge_index, sizes, lazy=False):
    """Collation-time construction of every position-independent index structure the DDM step reads (two-view
    molecule layout, super-edge incidence lists) from HOST molecule sizes: no read-back from the device, so a
    loader that calls this for batch k+1 does not stall behind the kernels of batch k.  The structures are cached on
    the tensors and found by the step (get_layout / pretrain_GeoSSL._two_view_batch / get_super_edge_layout).

    lazy: only leave the sizes on the tensors; the structures are built from them by the first step that asks (still
    without a read-back).  A step that replays a capacity-bucket graph (geossl_amd/bucket.py) never asks: it writes the
    same structures into the bucket's static buffers, so a streaming loader does not pay for both."""
    sizes = [int(n) for n in sizes]
    B = len(sizes)
    batch_vec._geossl_sizes = (sizes, batch_vec._version)
    if super_edge_index is not None:
        super_edge_index._geossl_grouped = (batch_vec._version, super_edge_index._version)
    if lazy:
        return
    b2 = torch.cat([batch_vec, batch_vec + B])
    lay2 = MolLayout(b2, 2 * B, sizes=sizes + sizes)
    lay2.loop_plan()  # the block plan of the layer loop, now: its small upload must not fall into a graph capture
    batch_vec._geossl_two_view = (b2, lay2, batch_vec._version)
    batch_vec._geossl_sizes = (sizes, batch_vec._version)  # a one-view layout (get_layout), if one is asked for, is built from them too
    if super_edge_index is not None:
        # the caller built super_edge_index from the same sizes (AtomTupleExtractor): its grouping needs no check
        get_super_edge_layout(batch_vec, super_edge_index, B, validate=False)


class EdgeLayout:
    """Per-batch bookkeeping of PaiNN's precomputed ``radius_edge_index`` (datasets_3D_Radius.py:120, collated with
    node offsets by dataloaders_AtomTuple.py:64-65): the edges grouped by idx_i = row 0 (forward scatter,
    painn.py:59,61) and by idx_j = row 1 (backward), each in ascending edge order."""

    def __init__(self, batch, edge_index, num_graphs, validate=True):
        _lib.require_cuda(batch, edge_index)
        if edge_index.dtype != torch.long or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise ValueError("radius_edge_index must be int64 [2, E]")
        self.idx_i = edge_index[0].contiguous()
        self.idx_j = edge_index[1].contiguous()
        batch = batch.contiguous()
        self.E, self.N, self.B = int(edge_index.size(1)), int(batch.numel()), int(num_graphs)
        dev = batch.device
        e_ptr = torch.zeros(self.B + 1, dtype=torch.int32, device=dev)
        stats = torch.zeros(2, dtype=torch.int64, device=dev)
        st = stream()
        call("geossl_super_edge_ptr", ptr(batch), ptr(self.idx_i), ptr(self.idx_j), self.E, self.B, ptr(e_ptr),
             ptr(stats), st)
        self.inc = {}
        for name, sides in (("i", 1), ("j", 2)):
            cnt = torch.zeros(self.N, dtype=torch.int32, device=dev)
            call("geossl_incidence_count", ptr(batch), ptr(self.idx_i), ptr(self.idx_j), ptr(e_ptr), self.N, sides,
                 ptr(cnt), st)
            iptr = torch.zeros(self.N + 1, dtype=torch.int64, device=dev)
            iptr[1:] = torch.cumsum(cnt, 0, dtype=torch.int64)
            idx = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
            call("geossl_incidence_fill", ptr(batch), ptr(self.idx_i), ptr(self.idx_j), ptr(e_ptr), self.N, sides,
                 ptr(iptr), ptr(idx), st)
            self.inc[name] = (iptr, idx)
        if validate and self.E > 0 and int(stats[1].item()):
            raise ValueError("radius_edge_index must be grouped by molecule in batch order with both ends in the "
                             "same molecule (collated MoleculeDataset3DRadius output is)")
        self._versions = (batch._version, edge_index._version)
        self._groups = {}

    def groups(self, side, mol_ptr=None):
        """Row layout of the matrix-pipe interaction kernels (painn_mma.hip) for the incidence list `side` ("i":
        edges by target, forward; "j": edges by source, backward): the edges of an atom in incidence order, padded to a
        multiple of four rows ("groups" - the rows of a group share their atom; an atom without edges has one group of
        padding).  -> (row_edge int32 [4 G'], -1 =
        padding; grp_atom int32 [G'] = 2 * atom + (last group of its atom), -1 = unused; grp_ptr int64 [N + 1] first group of
        an atom).  G' = E/4 + N + 1 bounds the
        number of groups, so nothing is read back from the device."""
        got = self._groups.get(side)
        if got is None:
            iptr, idx = self.inc[side]
            dev, N, E = iptr.device, self.N, self.E
            deg = iptr[1:] - iptr[:-1]
            grp_ptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
            # (an atom without edges gets one group of padding rows: it takes the same path as every other atom)
            torch.cumsum(torch.clamp((deg + 3) // 4, min=1), 0, out=grp_ptr[1:])
            cap = E // 4 + N + 1
            row_edge = torch.full((4 * cap,), -1, dtype=torch.int32, device=dev)
            grp_atom = torch.full((cap,), -1, dtype=torch.int32, device=dev)
            if E > 0:
                owner = torch.repeat_interleave(torch.arange(N, dtype=torch.int64, device=dev), deg, output_size=E)
                pos = 4 * grp_ptr[owner] + (torch.arange(E, dtype=torch.int64, device=dev) - iptr[owner])
                row_edge[pos] = idx[:E]
                # group code: atom * 2 + (the group is the last one of its atom)
                grp = pos // 4
                grp_atom[grp] = (2 * owner + (grp + 1 == grp_ptr[owner + 1]).to(torch.int64)).to(torch.int32)
            # the first group of EVERY atom (so that atoms without edges have their code too, without a data-dependent
            # index list, which would be a read-back): last group of its atom iff the atom has one group
            ng = grp_ptr[1:] - grp_ptr[:-1]
            grp_atom[grp_ptr[:-1]] = (2 * torch.arange(N, dtype=torch.int64, device=dev) + (ng == 1).to(torch.int64)).to(torch.int32)
            got = self._groups[side] = (row_edge, grp_atom, grp_ptr)
        if mol_ptr is None:
            return got
        # first group of every molecule (int32 [B + 1]), for the kernels' walk over molecules
        mg = self._groups.get((side, "mol"))
        if mg is None or mg[0] is not mol_ptr:
            mg = self._groups[(side, "mol")] = (mol_ptr, got[2][mol_ptr.long()].to(torch.int32))
        return got + (mg[1],)


def get_edge_layout(batch, edge_index, num_graphs):
    lay = getattr(edge_index, "_geossl_layout", None)
    if (lay is None or lay._versions != (batch._version, edge_index._version) or lay.E != edge_index.size(1)
            or lay.N != batch.numel()):
        lay = EdgeLayout(batch, edge_index, num_graphs)
        edge_index._geossl_layout = lay
    return lay
