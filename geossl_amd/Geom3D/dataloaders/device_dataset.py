"""A dataset of molecules resident in HBM and the loader that draws shuffled batches from it without host collation.

The reference trains from ``DataLoaderAtomTuple(dataset, batch_size, shuffle=True)`` (examples/pretrain_GeoSSL.py:
295-301): every step the host fetches ``batch_size`` molecules (Geom3D/datasets/datasets_3D.py:69-80), concatenates them
(``BatchAtomTuple.from_data_list``, Geom3D/dataloaders/dataloaders_AtomTuple.py:46-73) and copies the result to the
device (:248).  At MI355X step rates that collation IS the step (0.5-0.7 ms of host work beside a 1.3-1.6 ms
step at the reference's batch size).  Here the molecules are uploaded once:

* ``DeviceDataset`` keeps the concatenated ``x [Ntot, C] int64`` / ``positions [Ntot, 3] float32`` of all molecules in
  device memory, the atom offsets and sizes on the HOST (a size table: every index structure of a batch is a function
  of the sizes), and - for PaiNN - the concatenated per-molecule ``radius_edge_index`` built once on the device by the
  radius-graph kernel on the unperturbed geometry (datasets_3D_Radius.py:105-131) with per-molecule
  edge offsets on the host (ONE read-back, at construction).
* ``DeviceLoader`` yields ``DatasetBatch`` handles in the order ``torch.utils.data.DataLoader(shuffle=True)`` would
  visit the molecules (the same ``torch.randperm`` under the same seed).  A handle carries molecule ids and host sizes
  only.  A step that replays a captured graph hands it to ``geossl_gather_molecules`` (csrc/gather.hip), which writes
  the chosen molecules - atom rows, batch vector, super-edges, pair slots, incidence lists, radius edges - straight into
  the graph's static inputs: one small pinned upload of offsets + one launch per step, no read-back, no host tensors.
  Anything else that touches ``handle.x`` / ``.positions`` / ``.batch`` / ``.super_edge_index`` /
  ``.radius_edge_index`` gets the collated tensors (built by the same kernel into fresh memory, then cached), so a
  handle is accepted wherever a collated batch is (``do_DDM``, the modules themselves, eager steps).
"""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ..._lib import call, ptr, stream

_OPTIONS = {"combination": 0, "permutation": 1}


class _Staging:
    """A ring of pinned int32 staging buffers (grow-only) for the small per-step uploads; a slot is reused only after the
    copy that last read it has completed."""

    def __init__(self, slots=3):
        self.slots = [[None, None] for _ in range(slots)]
        self.i = 0

    def take(self, words):
        slot = self.slots[self.i]
        self.i = (self.i + 1) % len(self.slots)
        if slot[1] is not None:
            slot[1].synchronize()
            slot[1] = None
        if slot[0] is None or slot[0].numel() < words:
            slot[0] = torch.empty(max(words + words // 2, 1024), dtype=torch.int32).pin_memory()
        return slot

    @staticmethod
    def sent(slot):
        slot[1] = torch.cuda.Event()
        slot[1].record()


class DeviceDataset:
    def __init__(self, x, positions, sizes, device, option="combination", radius=None, max_num_neighbors=32):
        """x [Ntot, C] int64, positions [Ntot, 3] float32 (numpy or tensors; molecule after molecule), sizes [M] atoms
        per molecule.  option: the AtomTupleExtractor enumeration of the batches drawn from it.  radius: also build the
        per-molecule radius_edge_index (PaiNN) on this geometry."""
        if option not in _OPTIONS:
            raise ValueError("option is 'combination' or 'permutation'")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.GeosslHipError("a DeviceDataset lives on an MI355X (no CPU fallback)")
        if dev.index is None:   # ("cuda": the current device, by number - buckets and graphs compare devices)
            dev = torch.device("cuda", torch.cuda.current_device())
        as_t = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt).contiguous()
        self.device, self.option = dev, option
        self.sizes = np.ascontiguousarray(np.asarray(sizes, dtype=np.int64))
        M = self.sizes.shape[0]
        self.off = np.zeros(M + 1, dtype=np.int64)
        np.cumsum(self.sizes, out=self.off[1:])
        self.x, self.positions = as_t(x, torch.int64), as_t(positions, torch.float32)
        if self.x.dim() != 2 or self.positions.dim() != 2 or self.positions.size(1) != 3:
            raise ValueError("x is [Ntot, C], positions [Ntot, 3]")
        Nt = int(self.off[-1])
        if self.x.size(0) != Nt or self.positions.size(0) != Nt:
            raise ValueError("sizes sum to %d atoms, x has %d rows, positions %d" % (Nt, self.x.size(0), self.positions.size(0)))
        if M and (self.sizes.min() < 1 or Nt >= 2 ** 31):
            raise ValueError("molecules have at least one atom; at most 2^31 - 1 atoms in all")
        self.x_cols = int(self.x.size(1))
        self.pairs = self.sizes * (self.sizes - 1) // 2
        self.radius, self.edges, self.edge_cnt, self.edge_off = None, None, None, None
        self._staging = _Staging()
        if radius is not None:
            self._build_edges(float(radius), int(max_num_neighbors))

    # ---- construction
    @classmethod
    def from_numpy(cls, d, device, **kw):
        """From a dict with x, positions, sizes (geossl_amd.synthetic.make_molecules / make_batch)."""
        return cls(d["x"], d["positions"], d["sizes"], device, **kw)

    @classmethod
    def from_data_list(cls, data_list, device, **kw):
        """From the reference's per-molecule records (``Data`` objects with ``x`` and ``positions``,
        datasets_3D.py:69-80): concatenated once, uploaded once."""
        x = torch.cat([torch.as_tensor(d.x) for d in data_list], dim=0)
        pos = torch.cat([torch.as_tensor(d.positions) for d in data_list], dim=0)
        return cls(x, pos, [int(d.x.size(0)) for d in data_list], device, **kw)

    def _build_edges(self, radius, max_num_neighbors):
        """radius_graph(positions, r) of every molecule (datasets_3D_Radius.py:120) in two launches over the whole
        dataset: [source ; target] with DATASET atom ids, target-major - a molecule's edges are one contiguous run."""
        from ...ops import radius_cap
        M, Nt, dev = len(self), int(self.off[-1]), self.device
        mol_ptr = torch.from_numpy(self.off.astype(np.int32)).to(dev)
        max_n = int(self.sizes.max()) if M else 0
        r2 = float(torch.tensor(radius * radius, dtype=torch.float32))
        cap = radius_cap(max_num_neighbors)
        deg = torch.zeros(Nt, dtype=torch.int32, device=dev)
        call("geossl_radius_graph_count", ptr(self.positions), ptr(mol_ptr), M, max_n, r2, cap, ptr(deg), stream())
        edge_ptr = torch.zeros(Nt + 1, dtype=torch.int64, device=dev)
        edge_ptr[1:] = torch.cumsum(deg, 0, dtype=torch.int64)
        self.edge_off = edge_ptr[torch.from_numpy(self.off).to(dev)].cpu().numpy().astype(np.int64)   # the one read-back
        E = int(self.edge_off[-1])
        if E >= 2 ** 31:
            raise ValueError("at most 2^31 - 1 radius edges in all")
        e = torch.empty(2, max(E, 1), dtype=torch.int64, device=dev)
        w = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
        if E:
            call("geossl_radius_graph_fill", ptr(self.positions), ptr(mol_ptr), M, max_n, r2, cap, ptr(edge_ptr), ptr(e[0]),
                 ptr(e[1]), ptr(w), stream())
        self.edges, self.radius = e[:, :E], radius
        self.edge_cnt = np.diff(self.edge_off)

    def __len__(self):
        return int(self.sizes.shape[0])

    # ---- batches
    def batch(self, ids):
        return DatasetBatch(self, ids)

    def __getitem__(self, ids):
        if isinstance(ids, (int, np.integer)):
            ids = [int(ids)]
        return DatasetBatch(self, ids)

    def upload_plan(self, hb, with_edges):
        """int32 words [src_off (B) | mol_ptr (B+1) | se_ptr (B+1) | e_src_off (B) | e_ptr (B+1)] of a handle on the
        device (fresh memory) -> (tensor, offsets)."""
        B = hb.num_graphs
        o = {"src_off": 0, "mol_ptr": B, "se_ptr": 2 * B + 1}
        words = 3 * B + 2
        if with_edges:
            o["e_src_off"], o["e_ptr"] = words, words + B
            words += 2 * B + 1
        slot = self._staging.take(words)
        h = slot[0].numpy()
        h[0:B] = self.off[hb.ids]
        mp = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(hb._sizes, out=mp[1:])
        h[B:2 * B + 1] = mp
        sp = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(self.pairs[hb.ids] * (1 if self.option == "combination" else 2), out=sp[1:])
        h[2 * B + 1:3 * B + 2] = sp
        if with_edges:
            h[o["e_src_off"]:o["e_src_off"] + B] = self.edge_off[hb.ids]
            ep = np.zeros(B + 1, dtype=np.int64)
            np.cumsum(self.edge_cnt[hb.ids], out=ep[1:])
            h[o["e_ptr"]:o["e_ptr"] + B + 1] = ep
        blob = torch.empty(words, dtype=torch.int32, device=self.device)
        blob.copy_(slot[0][:words], non_blocking=True)
        self._staging.sent(slot)
        return blob, o

    def collate(self, hb):
        """The collated batch of a handle as fresh device tensors - what ``BatchAtomTuple.from_data_list`` over these
        molecules followed by ``.to(device)`` holds (bit for bit) - with the host-side sizes attached."""
        from ...pretrain_GeoSSL import Batch
        from ...layout import prepare_batch
        B, dev = hb.num_graphs, self.device
        N, S = hb.n_atoms, hb.n_super
        with_edges = self.edges is not None
        blob, o = self.upload_plan(hb, with_edges)
        x = torch.empty(N, self.x_cols, dtype=torch.int64, device=dev)
        pos = torch.empty(N, 3, dtype=torch.float32, device=dev)
        bvec = torch.empty(N, dtype=torch.int64, device=dev)
        sei = torch.empty(2, S, dtype=torch.int64, device=dev)
        g = _lib.Gather()
        g.x_src, g.pos_src, g.x_cols, g.option = ptr(self.x), ptr(self.positions), self.x_cols, _OPTIONS[self.option]
        base = blob.data_ptr()
        g.src_off, g.mol_ptr, g.se_ptr = base + 4 * o["src_off"], base + 4 * o["mol_ptr"], base + 4 * o["se_ptr"]
        g.x_dst, g.pos_dst, g.batch_dst = ptr(x), ptr(pos), ptr(bvec)
        if S:
            g.sei0, g.sei1 = ptr(sei[0]), ptr(sei[1])
        rei = None
        if with_edges:
            E = hb.n_edges
            rei = torch.empty(2, E, dtype=torch.int64, device=dev)
            if E:
                g.e0_src, g.e1_src = ptr(self.edges[0]), ptr(self.edges[1])
                g.e_src_off, g.e_ptr = base + 4 * o["e_src_off"], base + 4 * o["e_ptr"]
                g.e0_dst, g.e1_dst = ptr(rei[0]), ptr(rei[1])
        call("geossl_gather_molecules", C.byref(g), B, stream())
        out = Batch(x, pos, bvec, sei, rei, B, hb._sizes, self.option)
        prepare_batch(bvec, sei, hb._sizes, lazy=True)
        return out

    def gather_into(self, hb, x_dst, pos_dst, mol_ptr, zero=None):
        """x / positions of the handle's molecules into the static inputs of a per-structure graph (whose index tensors
        are bound): `mol_ptr` is the int32 [B+1] device array of that structure."""
        B = hb.num_graphs
        if x_dst.size(1) != self.x_cols or x_dst.size(0) != hb.n_atoms or pos_dst.size(0) != hb.n_atoms:
            raise ValueError("static inputs do not have the batch's shape")
        slot = self._staging.take(B)
        slot[0].numpy()[0:B] = self.off[hb.ids]
        dst = self.__dict__.get("_src_off_dev")   # (one buffer: upload and launch are ordered on the stream)
        if dst is None or dst.numel() < B:
            dst = self.__dict__["_src_off_dev"] = torch.empty(max(B, 1024), dtype=torch.int32, device=self.device)
        dst[:B].copy_(slot[0][:B], non_blocking=True)
        self._staging.sent(slot)
        g = _lib.Gather()
        g.x_src, g.pos_src, g.x_cols, g.option = ptr(self.x), ptr(self.positions), self.x_cols, _OPTIONS[self.option]
        g.src_off, g.mol_ptr = ptr(dst), ptr(mol_ptr)
        g.x_dst, g.pos_dst = ptr(x_dst), ptr(pos_dst)
        if zero is not None:
            g.zero, g.zero_count = ptr(zero), zero.numel()
        call("geossl_gather_molecules", C.byref(g), B, stream())


class DatasetBatch:
    """``batch_size`` molecules of a ``DeviceDataset`` by id: what the loader hands to a step.  Host side only - ids,
    sizes, counts; the collated tensors exist once somebody asks for them."""

    def __init__(self, dataset, ids):
        self._dataset = dataset
        self.ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
        if self.ids.ndim != 1 or not self.ids.size:
            raise ValueError("a batch is a non-empty list of molecule ids")
        if self.ids.min() < 0 or self.ids.max() >= len(dataset):
            raise IndexError("molecule id out of range")
        self._sizes = dataset.sizes[self.ids]            # (numpy: bucket.sizes_array takes it as it is)
        self._canonical = dataset.option
        self.device = dataset.device
        self.x_cols = dataset.x_cols
        self.num_graphs = int(self.ids.size)
        self.n_atoms = int(self._sizes.sum())
        P = int(dataset.pairs[self.ids].sum())
        self.n_super = P if dataset.option == "combination" else 2 * P
        self.n_edges = int(dataset.edge_cnt[self.ids].sum()) if dataset.edges is not None else None
        self._batch = None

    def materialize(self):
        if self._batch is None:
            self._batch = self._dataset.collate(self)
        return self._batch

    x = property(lambda self: self.materialize().x)
    positions = property(lambda self: self.materialize().positions)
    batch = property(lambda self: self.materialize().batch)
    super_edge_index = property(lambda self: self.materialize().super_edge_index)
    radius_edge_index = property(lambda self: self.materialize().radius_edge_index)

    def to(self, device, **kw):
        """``batch.to(device)`` of the reference's loop (pretrain_GeoSSL.py:248): the molecules are there already."""
        dev = torch.device(device)
        if dev.type != self.device.type or (dev.index is not None and dev.index != self.device.index):
            raise _lib.GeosslHipError("a DatasetBatch lives on %s (its molecules are gathered there; there is no copy to "
                                      "another device)" % self.device)
        return self

    def fingerprint(self):
        """pretrain_GeoSSL.structure_fingerprint of the collated batch, from the host sizes alone."""
        fp = self.__dict__.get("_fp")
        if fp is None:
            fp = self.__dict__["_fp"] = ("sizes", self._canonical, self.n_atoms, self.n_super,
                                         np.asarray(self._sizes, dtype=np.int32).tobytes())
        return fp


class DeviceLoader:
    """``DataLoaderAtomTuple(dataset, batch_size, shuffle)`` (dataloaders_AtomTuple.py:81-88) over a ``DeviceDataset``:
    iterating yields ``DatasetBatch`` handles.  With ``shuffle=True`` the molecules are visited in the order
    ``torch.utils.data.RandomSampler`` produces under the same global torch seed (one ``torch.randperm`` per epoch from
    a generator seeded by one draw of the default generator), so a run is reproducible against the reference's loader."""

    def __init__(self, dataset, batch_size=1, shuffle=True, drop_last=False, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.generator = generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return np.arange(n, dtype=np.int64)
        gen = self.generator
        # torch.utils.data.DataLoader.__iter__ draws the workers' base seed first (_BaseDataLoaderIter.__init__), from the
        # loader's generator or the default one ...
        torch.empty((), dtype=torch.int64).random_(generator=gen)
        if gen is None:   # ... then RandomSampler.__iter__ seeds a generator of its own with one more draw
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
        return torch.randperm(n, generator=gen).numpy()

    def __iter__(self):
        order, bs = self.order(), self.batch_size
        for k in range(len(self)):
            yield DatasetBatch(self.dataset, order[k * bs:(k + 1) * bs])
