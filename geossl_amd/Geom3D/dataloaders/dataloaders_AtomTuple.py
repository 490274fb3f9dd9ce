"""Device-side batch assembly behind the names of the reference's
Geom3D/dataloaders/dataloaders_AtomTuple.py.

The reference builds ``super_edge_index`` per molecule on the host with itertools (AtomTupleExtractor,
:9-37) and concatenates Data objects in Python (BatchAtomTuple.from_data_list, :46-73).  Here the collated
``batch`` vector is the input and everything index-shaped is produced on the GPU by ``geossl_atom_tuples``:
the same tuples in the same order (itertools.combinations / itertools.permutations, node offsets added),
bit-exact, with no per-molecule Python work.  ``ratio < 1`` (a host ``np.random.choice`` sub-sample, :25-29)
is not built.
"""
import torch

from ... import _lib
from ..._lib import call, ptr, stream
from ...layout import get_layout


class AtomTupleExtractor:
    """``AtomTupleExtractor(ratio=1, option="permutation")`` (:10-13).  Called with the sorted ``batch`` vector of
    a collated batch it returns ``super_edge_index`` (2, S) int64 for all molecules at once."""

    def __init__(self, ratio=1, option="permutation"):
        if ratio != 1:
            raise NotImplementedError("ratio < 1 sub-samples tuples with the host RNG (dataloaders_AtomTuple.py:25-29); "
                                      "only ratio = 1 is built")
        if option not in ("permutation", "combination"):
            option = "combination"  # the reference treats every other string as combination (:20-23)
        self.ratio, self.option = ratio, option

    def __call__(self, batch):
        _lib.require_cuda(batch)
        lay = get_layout(batch)
        B = lay.B
        n = (lay.mol_ptr[1:] - lay.mol_ptr[:-1]).to(torch.int64)
        cnt = n * (n - 1) if self.option == "permutation" else n * (n - 1) // 2
        tuple_ptr = torch.zeros(B + 1, dtype=torch.int64, device=batch.device)
        torch.cumsum(cnt, 0, out=tuple_ptr[1:])
        S = int(tuple_ptr[-1].item()) if B > 0 else 0
        sei = torch.empty(2, S, dtype=torch.int64, device=batch.device)
        if S > 0:
            call("geossl_atom_tuples", ptr(lay.mol_ptr), ptr(tuple_ptr), B, 1 if self.option == "permutation" else 0,
                 ptr(sei[0]), ptr(sei[1]), stream())
        return sei


class BatchAtomTuple:
    """The attributes ``do_DDM`` / ``NCSN_version_03`` read from a collated batch (:40-78), assembled on the device
    from per-atom tensors and the molecule sizes."""

    def __init__(self, x, positions, batch, super_edge_index, radius_edge_index=None, num_graphs=None):
        self.x, self.positions, self.batch, self.super_edge_index = x, positions, batch, super_edge_index
        self.radius_edge_index = radius_edge_index
        self._num_graphs = num_graphs

    @classmethod
    def from_sizes(cls, x, positions, sizes, option="combination", radius=None):
        """x [N, C] int64, positions [N, 3] f32 (already concatenated, on the GPU), sizes [B] atoms per molecule.
        ``radius``: also build ``radius_edge_index`` on the given geometry (datasets_3D_Radius.py:120)."""
        _lib.require_cuda(x, positions)
        host_sizes = None if torch.is_tensor(sizes) and sizes.is_cuda else [int(n) for n in sizes]
        sizes = torch.as_tensor(sizes, dtype=torch.int64, device=x.device)
        B = int(sizes.numel())
        batch = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=x.device), sizes)  # :61
        sei = AtomTupleExtractor(option=option)(batch)
        rei = None
        if radius is not None:
            from ... import ops
            rei = ops.radius_graph(positions, radius, batch)
        if host_sizes is not None and option == "combination":
            from ...layout import prepare_batch
            prepare_batch(batch, sei, host_sizes)  # the step's index structures, no device read-back
        return cls(x, positions, batch, sei, rei, B)

    @property
    def num_graphs(self):
        if self._num_graphs is None:
            self._num_graphs = self.batch[-1].item() + 1  # :75-78
        return self._num_graphs
