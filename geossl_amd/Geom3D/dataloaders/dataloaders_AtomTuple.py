"""Batch assembly behind the names of the reference's Geom3D/dataloaders/dataloaders_AtomTuple.py.

Two ways in, same results:

* the reference's own surface (:9-88) — ``AtomTupleExtractor()(data)`` as a per-molecule transform that sets
  ``data.super_edge_index``, ``BatchAtomTuple.from_data_list(data_list)`` and ``DataLoaderAtomTuple(dataset,
  batch_size, shuffle, **kw)`` — so the caller at pretrain_GeoSSL.py:289-301 runs unchanged.  Molecules live on the
  host there; ``batch.to(device)`` (pretrain_GeoSSL.py:248) moves the collated tensors and builds the index structures
  of the DDM step from the molecule sizes the collation already knows (no device read-back).
* the device path — ``AtomTupleExtractor()(batch_vector)`` and ``BatchAtomTuple.from_sizes`` —
  where everything index-shaped is produced on the GPU by ``geossl_atom_tuples`` from the collated ``batch`` vector:
  the same tuples in the same order (itertools.combinations / itertools.permutations, node offsets added), bit-exact,
  with no per-molecule Python work.

``torch_geometric`` is not a dependency: ``Data`` below is the small attribute container the reference's collate
needs (``keys``, item access, ``__cat_dim__``).
"""
import itertools
import re

import numpy as np
import torch
from torch.utils.data import DataLoader

from ... import _lib
from ..._lib import call, ptr, stream
from ...layout import get_layout

_INDEX_KEYS = ("edge_index", "radius_edge_index", "super_edge_index")  # offset by the node count when collated (:64-65)


class Data:
    """One molecule (or one collated batch): tensors as attributes, e.g. ``Data(x=..., positions=...)``.
    The part of ``torch_geometric.data.Data`` the reference's dataloader touches."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    def __cat_dim__(self, key, value):
        # PyG: index-like tensors ([2, E]) are concatenated along the last dimension, everything else along the first
        return -1 if re.search("(index|face)", key) else 0

    def _apply(self, fn):
        for k in self.keys:
            v = self[k]
            if torch.is_tensor(v):
                self[k] = fn(v)
        return self

    def contiguous(self):
        return self._apply(lambda t: t.contiguous())

    def to(self, device, **kw):
        return self._apply(lambda t: t.to(device, **kw))


def _tuples_np(n, option):
    """All tuples of one molecule exactly as the reference enumerates them (:19-24)."""
    if option == "permutation":
        t = list(itertools.permutations(np.arange(n), 2))
    else:
        t = list(itertools.combinations(np.arange(n), 2))
    return np.array(t).T


class AtomTupleExtractor:
    """``AtomTupleExtractor(ratio=1, option="permutation")`` (:9-37).

    ``extractor(data)``: the reference's per-molecule transform — sets ``data.super_edge_index`` ((2, S) int64, local
    atom indices) and returns ``data``.  ``ratio < 1`` keeps ``int(S * ratio)`` tuples drawn with
    ``np.random.choice(S, ., replace=False)`` (:25-29): same global numpy stream, same call, same tuples.

    ``extractor(batch_vector)`` (a sorted CUDA int64 ``batch`` tensor): ``super_edge_index`` of all molecules at once,
    node offsets included, built on the GPU (ratio = 1) — or on the host in molecule order with the same
    ``np.random.choice`` calls when ``ratio < 1``."""

    def __init__(self, ratio=1, option="permutation"):
        self.ratio = ratio
        self.option = option

    def _one(self, n):
        if n < 2:
            return torch.empty((2, 0), dtype=torch.long)
        sei = _tuples_np(n, self.option)
        if self.ratio < 1:
            M = sei.shape[1]
            sampled_M = int(M * self.ratio)
            sampled = np.random.choice(M, sampled_M, replace=False)
            sei = sei[:, sampled]
        return torch.tensor(sei, dtype=torch.long)

    def __call__(self, data):
        if torch.is_tensor(data):
            return self._from_batch_vector(data)
        data.super_edge_index = self._one(len(data.x))
        # the full enumeration is a function of the atom count: collated batches of such molecules can share step graphs
        # (pretrain_GeoSSL.structure_fingerprint); a sampled subset (ratio < 1) is not
        data._sei_canonical = self.option if self.ratio >= 1 else None
        return data

    def _from_batch_vector(self, batch):
        _lib.require_cuda(batch)
        lay = get_layout(batch)
        B = lay.B
        n = (lay.mol_ptr[1:] - lay.mol_ptr[:-1]).to(torch.int64)
        if self.ratio < 1:  # host RNG decides which tuples stay: per molecule, in batch order (:25-29)
            off = lay.mol_ptr[:-1].tolist()
            parts = [self._one(int(k)) + o for k, o in zip(n.tolist(), off)]
            sei = torch.cat(parts, dim=1) if parts else torch.empty((2, 0), dtype=torch.long)
            return sei.to(batch.device)
        perm = self.option == "permutation"
        cnt = n * (n - 1) if perm else n * (n - 1) // 2
        tuple_ptr = torch.zeros(B + 1, dtype=torch.int64, device=batch.device)
        torch.cumsum(cnt, 0, out=tuple_ptr[1:])
        S = int(tuple_ptr[-1].item()) if B > 0 else 0
        sei = torch.empty(2, S, dtype=torch.int64, device=batch.device)
        if S > 0:
            call("geossl_atom_tuples", ptr(lay.mol_ptr), ptr(tuple_ptr), B, 1 if perm else 0, ptr(sei[0]), ptr(sei[1]),
                 stream())
        return sei


class BatchAtomTuple(Data):
    """A collated batch with the attributes ``do_DDM`` / ``NCSN_version_03`` read (:40-78): ``x``, ``positions``,
    ``batch``, ``super_edge_index`` [, ``radius_edge_index``], ``num_graphs``."""

    def __init__(self, batch=None, **kwargs):
        super().__init__(**kwargs)
        self.batch = batch
        self._sizes = None       # atoms per molecule (host integers) when the collation knows them
        self._canonical = None   # the AtomTupleExtractor option when super_edge_index is its full enumeration
        self._num_graphs = None

    @staticmethod
    def from_data_list(data_list):
        """:46-73 — concatenate per-molecule ``Data`` objects; the three index keys get the cumulative node offset,
        ``batch = full((n_i,), i)``."""
        keys = [set(data.keys) for data in data_list]
        keys = list(set.union(*keys))
        assert "batch" not in keys
        items = {key: [] for key in keys}
        bvec, sizes = [], []
        cumsum_node = 0
        for i, data in enumerate(data_list):
            num_nodes = data.x.size()[0]
            bvec.append(torch.full((num_nodes,), i, dtype=torch.long))
            for key in data.keys:
                item = data[key]
                if key in _INDEX_KEYS:
                    item = item + cumsum_node
                items[key].append(item)
            cumsum_node += num_nodes
            sizes.append(int(num_nodes))
        out = BatchAtomTuple()
        for key in keys:
            out[key] = torch.cat(items[key], dim=data_list[0].__cat_dim__(key, items[key][0]))
        out.batch = torch.cat(bvec, dim=-1)
        out._sizes = sizes
        out._num_graphs = len(sizes)
        marks = {getattr(data, "_sei_canonical", None) for data in data_list}
        out._canonical = marks.pop() if len(marks) == 1 else None
        return out.contiguous()

    @classmethod
    def from_sizes(cls, x, positions, sizes, option="combination", radius=None):
        """Device path: x [N, C] int64, positions [N, 3] f32 (already concatenated, on the GPU), sizes [B] atoms per
        molecule.  ``radius``: also build ``radius_edge_index`` on the given geometry (datasets_3D_Radius.py:120)."""
        _lib.require_cuda(x, positions)
        host_sizes = None if torch.is_tensor(sizes) and sizes.is_cuda else [int(n) for n in sizes]
        sizes = torch.as_tensor(sizes, dtype=torch.int64, device=x.device)
        B = int(sizes.numel())
        batch = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=x.device), sizes)  # :61
        sei = AtomTupleExtractor(option=option)(batch)
        out = cls(batch=batch, x=x, positions=positions, super_edge_index=sei)
        if radius is not None:
            from ... import ops
            out.radius_edge_index = ops.radius_graph(positions, radius, batch)
        out._sizes, out._num_graphs = host_sizes, B
        out._canonical = option if host_sizes is not None else None
        out._prepare(option == "combination")
        return out

    def _prepare(self, grouped=True):
        """Index structures of the DDM step from the host-side molecule sizes (layout.prepare_batch)."""
        if self._sizes is not None and self.batch is not None and self.batch.is_cuda and grouped:
            from ...layout import prepare_batch
            # lazy: a step that replays a capacity-bucket graph builds these structures in the bucket's own buffers
            prepare_batch(self.batch, getattr(self, "super_edge_index", None), self._sizes, lazy=True)

    def to(self, device, **kw):
        super().to(device, **kw)
        # collated AtomTupleExtractor output is grouped by molecule in batch order whatever the option / ratio
        self._prepare()
        return self

    @property
    def num_graphs(self):
        """:75-78"""
        if self._num_graphs is None:
            self._num_graphs = self.batch[-1].item() + 1
        return self._num_graphs


def _collate(data_list):
    return BatchAtomTuple.from_data_list(data_list)


class DataLoaderAtomTuple(DataLoader):
    """:81-88 — a ``torch.utils.data.DataLoader`` whose collate function is ``BatchAtomTuple.from_data_list``."""

    def __init__(self, dataset, batch_size=1, shuffle=True, **kwargs):
        super().__init__(dataset, batch_size, shuffle, collate_fn=_collate, **kwargs)
