"""Step API of ``pretrain_GeoSSL.py --GeoSSL_option=DDM`` (examples/pretrain_GeoSSL.py:68-74,179-212,
234-260) on the HIP path.

``perturb`` and ``do_DDM`` keep the reference signatures.  The reference reads two module globals
(``NCSN_model_01/02``, :207-208); here they are module attributes with the same names, or can be
passed with ``NCSN_models=``.  ``DDMTrainer`` is the training-loop body (:234-260) with all
parameters in one flat buffer: one fused Adam launch and one RCCL all-reduce per step.

Forward + backward of a step are captured into HIP graphs keyed on a FINGERPRINT of the batch's index structure
(``structure_fingerprint``) and replayed - by ``DDMTrainer`` and, for a caller that keeps the reference's own loop
(``loss, acc = do_DDM(...); optimizer.zero_grad(); loss.backward(); optimizer.step()``, :249-260), by ``do_DDM``
itself: the loss it returns is the output of an autograd node whose backward hands the replayed gradients to autograd.
"""
import ctypes as C
import os
import threading
import warnings
from collections import OrderedDict

import numpy as np
import torch
from .switches import env as _env

from . import _lib, ops
from ._lib import call, ptr, stream
from .layout import MolLayout

NCSN_model_01 = None
NCSN_model_02 = None


def perturb(x, positions, mu, sigma, noise=None, device_noise=False, generator=None):
    """pretrain_GeoSSL.py:68-74.  Default: the N(mu, sigma) draw is made on the CPU and copied to
    the device exactly like the reference (:72), so the same seed gives the same noise.
    ``device_noise=True`` draws on the GPU instead (no host RNG / PCIe copy); ``noise=`` injects it."""
    x_perturb = x
    device = positions.device
    if noise is None:
        if device_noise:
            noise = torch.empty_like(positions).normal_(mu, sigma, generator=generator)
        else:
            noise = torch.normal(mu, sigma, size=positions.size()).to(device)
    positions_perturb = ops.add_scaled(positions, noise, 1.0)
    return x_perturb, positions_perturb


class _SplitViews(torch.autograd.Function):
    """h of the fused (clean ‖ perturbed) batch -> the two per-view halves.  Plain slicing would make autograd pad each
    half's gradient to the full shape and add the two (two fills, two copies and an add over 2N x F); the backward
    here is one concatenation - or nothing at all when the two gradients already lie behind one another in one buffer,
    which is where the heads put them when the halves carry a `GradSlot` (`split_views`)."""

    STATS = {"adjacent": 0, "cat": 0}  # how the backward joined the halves (tests)

    @staticmethod
    def forward(ctx, h, n):
        ctx.n = n
        return h[:n], h[n:]

    @staticmethod
    def backward(ctx, g1, g2):
        if (g1.is_contiguous() and g2.is_contiguous() and g1.dtype == g2.dtype and g1.dim() == 2
                and g1.untyped_storage().data_ptr() == g2.untyped_storage().data_ptr()
                and g2.storage_offset() == g1.storage_offset() + g1.numel()):
            _SplitViews.STATS["adjacent"] += 1
            return g1.new_empty(0).set_(g1.untyped_storage(), g1.storage_offset(),
                                        (g1.size(0) + g2.size(0), g1.size(1)), (g1.size(1), 1)), None
        _SplitViews.STATS["cat"] += 1
        return torch.cat([g1, g2]), None


class GradSlot:
    """Where a consumer of one half of a split tensor may put that half's gradient: rows [lo, hi) of a buffer the two
    halves share (allocated by whoever asks first).  A head that finds the slot on its input (`_geossl_grad_slot`,
    NCSN._NcsnLoss) writes its gradient there and returns that view; anything else ignores it and `_SplitViews.backward`
    concatenates as before."""

    def __init__(self, pair, lo, hi, total):
        self.pair, self.lo, self.hi, self.total = pair, lo, hi, total

    def rows(self, n, width, dtype, device):
        if n != self.hi - self.lo:
            return None
        buf, given = self.pair.get("buf"), self.pair.setdefault("given", set())
        if (buf is None or self.lo in given  # a second backward through the same graph: never into a buffer handed out
                or buf.shape != (self.total, width) or buf.dtype != dtype or buf.device != device):
            buf = torch.empty(self.total, width, dtype=dtype, device=device)
            self.pair["buf"] = buf
            given.clear()
        given.add(self.lo)
        return buf[self.lo:self.hi]


def split_views(h, n):
    a, b = _SplitViews.apply(h, n)
    pair = {}
    a._geossl_grad_slot = GradSlot(pair, 0, n, h.size(0))
    b._geossl_grad_slot = GradSlot(pair, n, h.size(0), h.size(0))
    return a, b


def _two_view_batch(batch_vec, num_graphs):
    """batch ids of the concatenated (clean ‖ perturbed) 2B-molecule batch + its layout, cached on
    the batch tensor object."""
    cached = getattr(batch_vec, "_geossl_two_view", None)
    if cached is None or cached[2] != batch_vec._version:
        b2 = torch.cat([batch_vec, batch_vec + num_graphs])
        hs = getattr(batch_vec, "_geossl_sizes", None)  # host sizes left by layout.prepare_batch(lazy=True)
        sizes = hs[0] if hs is not None and hs[1] == batch_vec._version and len(hs[0]) == num_graphs else None
        lay2 = MolLayout(b2, 2 * num_graphs, sizes=None if sizes is None else sizes + sizes)
        if sizes is not None and not torch.cuda.is_current_stream_capturing():
            lay2.loop_plan()
        cached = (b2, lay2, batch_vec._version)
        batch_vec._geossl_two_view = cached
    return cached[0], cached[1]


def _two_view_edges(batch_vec, edge_index, num_graphs):
    """(batch ids, radius_edge_index) of the concatenated (clean ‖ perturbed) batch for PaiNN — the perturbed view
    keeps the clean view's graph (PaiNN does not re-derive it) — cached on the edge_index tensor so the
    incidence structures behind it are built once per batch."""
    cached = getattr(edge_index, "_geossl_two_view", None)
    key = (batch_vec._version, edge_index._version)
    if cached is None or cached[2] != key:
        b2 = torch.cat([batch_vec, batch_vec + num_graphs])
        e2 = torch.cat([edge_index, edge_index + batch_vec.numel()], dim=1)
        cached = (b2, e2, key)
        edge_index._geossl_two_view = cached
    return cached[0], cached[1]


def do_DDM(args, batch, model, criterion=None, mu=0.0, sigma=0.3, num_neg=1, NCSN_models=None, noise=None,
           fuse_views=True, device_noise=False, graph=None):
    """pretrain_GeoSSL.py:179-212 -> (loss, 0).

    noise: optional dict with pos_noise [N,3], noise_level_1/2 [B] int64, dist_noise_1/2 [S,1] —
    the five random draws of the step; anything missing is drawn like the reference does.
    fuse_views: run the clean and the perturbed view through the backbone as one 2B-molecule batch
    (molecules never interact, so every row sees the same arithmetic as in two separate calls).
    graph: replay a captured HIP graph of forward + backward (see ``_AutogradStep``) when gradients are wanted and the
    batch's index structure has a fingerprint; default: ``args.step_graph`` if present, else on unless
    ``GEOSSL_NO_STEP_GRAPH`` is set.  The draws, the loss and the gradients are those of the eager path, bit for bit.
    """
    n1, n2 = NCSN_models if NCSN_models is not None else (NCSN_model_01, NCSN_model_02)
    if n1 is None or n2 is None:
        raise RuntimeError("set geossl_amd.pretrain_GeoSSL.NCSN_model_01/02 or pass NCSN_models=(m1, m2)")
    if graph is None:
        graph = getattr(args, "step_graph", _env("GEOSSL_NO_STEP_GRAPH") is None)
    if graph and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing() and fuse_views:
        loss = _autograd_step(model, n1, n2).run(args, batch, mu, sigma, noise, device_noise)
        if loss is not None:
            return loss, 0
    return _do_ddm_eager(args, batch, model, mu, sigma, (n1, n2), noise, fuse_views, device_noise), 0


def _do_ddm_eager(args, batch, model, mu, sigma, heads, noise, fuse_views, device_noise):
    """The step as eager launches behind autograd's own nodes (one per backbone / head)."""
    noise = noise or {}
    n1, n2 = heads
    positions = batch.positions
    x_01 = batch.x[:, 0]
    positions_01 = positions
    super_edge_index = batch.super_edge_index
    if args.model_3d not in ("schnet", "painn"):
        raise Exception("3D model {} not included.".format(args.model_3d))
    bucket = getattr(batch, "_bucket", None)
    if bucket is not None:
        # the static batch of a capacity bucket (geossl_amd/bucket.py): tensors have the bucket's capacity, the real counts
        # are device data (bucket.dyn); the fused batch is [view 0 | view 1 | unused] and goes to the heads whole
        if args.model_3d != bucket.kind or not fuse_views or getattr(args, "normalize", False):
            raise _lib.GeosslHipError("capacity buckets serve the fused step of the backbone they were made for")
        pos_noise = noise.get("pos_noise")
        if pos_noise is None:
            pos_noise = torch.empty_like(positions).normal_(mu, sigma)
        pos2, distance_01, distance_02, x2 = ops.ddm_views(positions, pos_noise, super_edge_index[0], super_edge_index[1],
                                                           z=x_01, dyn=bucket.dyn)
        if bucket.kind == "schnet":
            _, h = model(x2, pos2, bucket.b2, return_latent=True, layout=bucket.lay2, latent_only=True)
        else:
            _, h = model(x2, pos2, bucket.e2, bucket.b2, return_latent=True, latent_only=True, layout=bucket.lay2,
                         edge_layout=bucket.el)
        from .NCSN import ddm_heads_loss
        return ddm_heads_loss(n1, n2, batch, h, distance_02, None, distance_01,
                              noise_level_1=noise.get("noise_level_1"), distance_noise_1=noise.get("dist_noise_1"),
                              noise_level_2=noise.get("noise_level_2"), distance_noise_2=noise.get("dist_noise_2"),
                              out_scale=0.5)
    if fuse_views:
        # perturb (:68-74), the concatenation of the two views and both sets of super-edge lengths (:199-205) in one
        # launch; the draw itself is perturb's (same generator, same place in the order of draws)
        pos_noise = noise.get("pos_noise")
        if pos_noise is None:
            if device_noise:
                pos_noise = torch.empty_like(positions).normal_(mu, sigma)
            else:
                pos_noise = torch.normal(mu, sigma, size=positions.size()).to(positions.device)
        N = positions.size(0)
        pos2, distance_01, distance_02, x2 = ops.ddm_views(positions, pos_noise, super_edge_index[0], super_edge_index[1],
                                                           z=x_01)
        if args.model_3d == "schnet":
            b2, lay2 = _two_view_batch(batch.batch, batch.num_graphs)
            # the readout is dead compute in this step (`_` at pretrain_GeoSSL.py:187): not evaluated
            _, h = model(x2, pos2, b2, return_latent=True, layout=lay2, latent_only=True)
        else:
            b2, e2 = _two_view_edges(batch.batch, batch.radius_edge_index, batch.num_graphs)
            _, h = model(x2, pos2, e2, b2, return_latent=True, latent_only=True)
        molecule_3D_repr_01, molecule_3D_repr_02 = split_views(h, N)
    else:
        x_02, positions_02 = perturb(x_01, positions, mu, sigma, noise=noise.get("pos_noise"), device_noise=device_noise)
        if args.model_3d == "schnet":
            _, molecule_3D_repr_01 = model(x_01, positions_01, batch.batch, return_latent=True)
            _, molecule_3D_repr_02 = model(x_02, positions_02, batch.batch, return_latent=True)
        else:
            _, molecule_3D_repr_01 = model(x_01, positions_01, batch.radius_edge_index, batch.batch, return_latent=True)
            _, molecule_3D_repr_02 = model(x_02, positions_02, batch.radius_edge_index, batch.batch, return_latent=True)
        distance_01 = ops.pair_distance(positions_01, super_edge_index[0], super_edge_index[1])  # :199-201
        distance_02 = ops.pair_distance(positions_02, super_edge_index[0], super_edge_index[1])  # :203-205

    if getattr(args, "normalize", False):  # :193-195
        molecule_3D_repr_01 = ops.row_normalize(molecule_3D_repr_01)
        molecule_3D_repr_02 = ops.row_normalize(molecule_3D_repr_02)

    # cross-view pairing (:207-208); each head returns 0.5 * its loss so the sum is (l1 + l2) / 2 (:210)
    # (both heads in the same launches where they allow it: NCSN.ddm_heads_loss)
    from .NCSN import ddm_heads_loss
    return ddm_heads_loss(n1, n2, batch, molecule_3D_repr_01, distance_02, molecule_3D_repr_02, distance_01,
                          noise_level_1=noise.get("noise_level_1"), distance_noise_1=noise.get("dist_noise_1"),
                          noise_level_2=noise.get("noise_level_2"), distance_noise_2=noise.get("dist_noise_2"),
                          out_scale=0.5)


class Batch:
    """Device-resident collated batch with the attributes do_DDM / NCSN_version_03 read
    (BatchAtomTuple, dataloaders_AtomTuple.py:40-78)."""

    def __init__(self, x, positions, batch, super_edge_index, radius_edge_index=None, num_graphs=None, sizes=None,
                 canonical=None):
        self.x, self.positions, self.batch, self.super_edge_index = x, positions, batch, super_edge_index
        self.radius_edge_index = radius_edge_index
        self._num_graphs = num_graphs
        # host-side knowledge of the collation (structure_fingerprint): atoms per molecule, and the AtomTupleExtractor
        # option when super_edge_index is its full enumeration in batch order
        self._sizes = None if sizes is None else [int(n) for n in sizes]
        self._canonical = canonical

    @property
    def num_graphs(self):
        if self._num_graphs is None:
            self._num_graphs = self.batch[-1].item() + 1  # dataloaders_AtomTuple.py:75-78
        return self._num_graphs

    @classmethod
    def from_numpy(cls, d, device, prepare=True):
        """prepare=False: the index structures of the step are built (from the host sizes) by the first step that needs
        them - a step that replays a capacity-bucket graph never does."""
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        rei = t(d["radius_edge_index"]) if "radius_edge_index" in d else None
        ng = int(len(d["sizes"])) if "sizes" in d else None
        canonical = canonical_option(d["sizes"], d["super_edge_index"]) if "sizes" in d else None
        out = cls(t(d["x"]), t(d["positions"]), t(d["batch"]), t(d["super_edge_index"]), rei, ng, d.get("sizes"),
                  canonical)
        if "sizes" in d:  # collation knows the molecule sizes on the host: index structures without a device read-back
            from .layout import prepare_batch
            prepare_batch(out.batch, out.super_edge_index, d["sizes"], lazy=not prepare)
        return out

    def to(self, device):
        dev = torch.device(device)
        if all(a is None or (a.device == dev or (dev.index is None and a.device.type == dev.type))
               for a in (self.x, self.positions, self.batch, self.super_edge_index, self.radius_edge_index)):
            return self  # (already there: like Tensor.to, no new object - what is cached on this one stays)
        mv = lambda a: None if a is None else a.to(device)
        return Batch(mv(self.x), mv(self.positions), mv(self.batch), mv(self.super_edge_index),
                     mv(self.radius_edge_index), self._num_graphs, self._sizes, self._canonical)


_PAIR_CACHE = {}


def canonical_option(sizes, super_edge_index):
    """"combination" / "permutation" when the HOST array super_edge_index [2, S] is exactly AtomTupleExtractor's full
    enumeration (dataloaders_AtomTuple.py:19-24, ratio = 1) of molecules with these sizes, collated in batch order with
    node offsets (:64-65); None otherwise.  Such index tensors are a function of the sizes alone."""
    from .synthetic import combination_pairs, permutation_pairs
    sizes = np.asarray(sizes, dtype=np.int64)
    sei = np.asarray(super_edge_index)
    S = sei.shape[1] if sei.ndim == 2 else -1
    off = np.concatenate([[0], np.cumsum(sizes)])
    for option, count, pairs in (("combination", sizes * (sizes - 1) // 2, combination_pairs),
                                 ("permutation", sizes * (sizes - 1), permutation_pairs)):
        if int(count.sum()) != S or S == 0:
            continue
        parts = []
        for m, n in enumerate(sizes.tolist()):
            if (option, n) not in _PAIR_CACHE:
                _PAIR_CACHE[(option, n)] = pairs(n)
            parts.append(_PAIR_CACHE[(option, n)] + off[m])
        if np.array_equal(np.concatenate(parts, axis=1), sei):
            return option
    return None


_UID = [0]


def _tensor_uid(t_):
    """A number that identifies the tensor OBJECT for as long as it lives and is never handed out again (id() is: CPython
    reuses the address of a freed tensor, and a streaming loader would then look like a batch that came back)."""
    uid = t_.__dict__.get("_geossl_uid") if hasattr(t_, "__dict__") else None
    if uid is None:
        _UID[0] += 1
        uid = _UID[0]
        try:
            t_._geossl_uid = uid
        except AttributeError:
            return ("id", id(t_))
    return uid


def structure_fingerprint(batch, model_3d="schnet"):
    """Hashable identity of everything a captured step binds besides x, positions and the noise: the batch vector,
    super_edge_index and (PaiNN) radius_edge_index with the index structures derived from them.

    * A batch that carries its molecule sizes on the host AND whose super_edge_index is known to be the extractor's
      full enumeration (``_sizes`` / ``_canonical``: set by ``Batch.from_numpy``, ``BatchAtomTuple.from_sizes`` and by
      ``BatchAtomTuple.from_data_list`` over molecules that went through ``AtomTupleExtractor(ratio=1)``) has index
      tensors that are a function of the sizes: the fingerprint is the sizes, so every batch of a shuffled loader with
      the same molecule sizes in the same order shares a graph - and no other batch does.
    * Anything else (sampled tuples, PaiNN's geometry-dependent edge list, batches built by hand) is identified by
      the tensor OBJECTS and their versions: a graph is replayed only for the very tensors it was captured on (a
      device-resident, pre-collated batch that comes back every epoch)."""
    if getattr(batch, "_dataset", None) is not None and model_3d != "painn":
        return batch.fingerprint()   # a handle on a device-resident dataset: its index tensors ARE a function of the sizes
    tag = lambda t_: None if t_ is None else (_tensor_uid(t_), t_._version, tuple(t_.shape))
    rei = getattr(batch, "radius_edge_index", None) if model_3d == "painn" else None
    tags = (tag(batch.batch), tag(batch.super_edge_index), tag(rei))
    cached = batch.__dict__.get("_geossl_fp")
    if cached is not None and cached[0] == tags:
        return cached[1]
    sizes, canon = getattr(batch, "_sizes", None), getattr(batch, "_canonical", None)
    if sizes is not None and canon is not None and rei is None and model_3d != "painn":
        fp = ("sizes", canon, int(batch.batch.numel()), int(batch.super_edge_index.size(1)),
              np.asarray(sizes, dtype=np.int32).tobytes())
    else:
        fp = ("tensors",) + tags
    batch.__dict__["_geossl_fp"] = (tags, fp)
    return fp


class Args:
    """The three fields do_DDM reads from the reference's argparse namespace."""

    def __init__(self, model_3d="schnet", normalize=False):
        self.model_3d = model_3d
        self.normalize = normalize


_NOISE_KEYS = ("pos_noise", "noise_level_1", "dist_noise_1", "noise_level_2", "dist_noise_2")
_OWN_CAPTURE = [0]  # > 0 while StepGraphs._capture has a capture open


def own_capture_open():
    """True while a StepGraphs capture of this module is recording on the current stream: the one situation in which the
    step's passes leave clearing the gradient buffer to StepGraphs.refresh.  A caller that captures the step in a CUDA
    graph of ITS OWN gets the fill recorded into that graph like any other launch."""
    return _OWN_CAPTURE[0] > 0 and torch.cuda.is_current_stream_capturing()


class StepGraphs:
    """HIP graphs of forward + backward of the DDM step; all in ONE memory pool (their activations are dead between
    steps, so N graphs cost the device memory of the largest plus the static inputs and the loss of each).
    ``fwd_bwd(batch, noise) -> loss`` is the owner's eager step; it must leave the gradients in buffers that are the same
    for every call (the graph binds their addresses).  Least-recently-used graphs are dropped beyond ``max_graphs``.

    Which graph serves a batch (``mode="auto"``):

    * a batch of EQUAL-SIZED molecules, and anything a bucket cannot take (PaiNN's geometry-dependent edge list, sampled
      tuples, batches built by hand): one graph per ``structure_fingerprint`` - the size sequence, or the tensor objects;
      a structure that only a fingerprint identifies is captured on its SECOND sighting (a loader that never repeats one
      never pays for a capture);
    * every other SchNet batch - ragged molecules in shuffled order, the reference's loader (pretrain_GeoSSL.py:301) - one
      graph per (molecules in the batch, tuple option) at a CAPACITY (``geossl_amd/bucket.py``): the graph's kernels read
      the batch's real atom / pair-slot / super-edge counts and every index structure from device memory, so ONE graph
      replays on any size sequence; a batch that outgrows the capacity makes a larger bucket (its graph is captured
      again: a handful of times at the start of a run).

    ``mode="structure"``: per-structure graphs only, captured at the first sighting (pre-collated, device-resident
    batches that come back every epoch)."""

    def __init__(self, fwd_bwd, model_3d, max_graphs=256, split=None, mode="auto", normalize=False, modules=None):
        # split = (fwd(batch, noise) -> loss with its autograd graph, bwd(loss)): forward and backward captured as TWO
        # graphs (same pool, same capture stream; replayed in this order) - the forward's loss is then on the device
        # before the backward runs, and the backward can run on a side stream while the host goes on (_AutogradStep)
        self.fwd_bwd, self.model_3d, self.max_graphs, self.split = fwd_bwd, model_3d, max_graphs, split
        self.mode, self.normalize = mode, normalize
        self.modules = modules  # (backbone, head, head): what bucket.modules_ok looks at
        self.graphs, self.pool = OrderedDict(), None
        self.enabled = True
        self.captures = 0
        self._warned = False
        self._seen = OrderedDict()
        self.bucket_ok = True   # cleared when a bucket capture failed: per-structure graphs from then on
        self._mod_ok = None
        # a float32 buffer the owner wants cleared before every replay (its flat gradient buffer): done by the launch that
        # refreshes the graph's inputs instead of a fill node inside the graph (one launch less per step)
        self.zero_with_refresh = None

    def __len__(self):
        return len(self.graphs)

    # ---- which graph
    def bucket_key(self, batch):
        """("bucket", molecules, option) when the batch goes through a capacity bucket, else None."""
        from . import bucket as bk
        if (self.mode != "auto" or not self.bucket_ok or _env("GEOSSL_NO_BUCKETS")
                or not bk.eligible(batch, self.model_3d, self.normalize)
                or (self.model_3d != "painn" and bk.is_uniform(batch))   # (PaiNN: the edge list differs batch by batch anyway)
                or self.modules is None or not self._modules_ok()):
            return None
        return ("bucket", len(batch._sizes), batch._canonical)

    def _modules_ok(self):
        """bucket.modules_ok of this engine's modules, remembered per state of the switches it reads (the modules of a
        StepGraphs are fixed: a replaced parameter rebuilds the engine, _AutogradStep.unchanged / DDMTrainer)."""
        from . import bucket as bk
        env = tuple(_env(k) for k in bk.MODULE_SWITCHES + bk.PAINN_SWITCHES)
        if self._mod_ok is None or self._mod_ok[0] != env:
            self._mod_ok = (env, bk.modules_ok(*self.modules))
        return self._mod_ok[1]

    def lookup(self, batch):
        """The graph that serves this batch (None: not captured yet, or its bucket is too small)."""
        key = self.bucket_key(batch)
        if key is not None:
            from . import bucket as bk
            g = self.graphs.get(key)
            if g is None:
                return None
            counts = bk.batch_counts(bk.sizes_array(batch), batch._canonical)
            if not g["bucket"].fits(counts, bk.size_range(batch)[1], self._edges(batch)):
                return None
            g["counts"] = counts
            self.graphs.move_to_end(key)
            return g
        fp = structure_fingerprint(batch, self.model_3d)
        g = self.graphs.get(fp)
        if g is not None:
            self.graphs.move_to_end(fp)
        return g

    def _edges(self, batch):
        """Edges of the batch's radius_edge_index (PaiNN; a tensor shape: no read-back), else None."""
        if self.model_3d != "painn":
            return None
        if getattr(batch, "_dataset", None) is not None:
            return batch.n_edges   # (the dataset's host-side edge counts)
        return int(batch.radius_edge_index.size(1))

    def capture_now(self, batch):
        """Should a batch without a graph be captured at this sighting?  Buckets and equal-sized molecules: yes (their
        graph serves every later batch of the kind); a structure known by fingerprint only: from its second sighting on
        (mode "structure": always)."""
        if self.mode != "auto" or self.bucket_key(batch) is not None:
            return True
        from . import bucket as bk
        if bk.is_uniform(batch) and getattr(batch, "_canonical", None) is not None and self.model_3d == "schnet":
            return True
        return self.seen_before(batch)

    def seen_before(self, batch):
        """True from the second call on for one fingerprint (a bounded memory of hashes: a collision or a forgotten
        entry only moves the moment of a capture)."""
        h = hash(structure_fingerprint(batch, self.model_3d))
        seen = h in self._seen
        self._seen[h] = True
        self._seen.move_to_end(h)
        while len(self._seen) > 8192:
            self._seen.popitem(last=False)
        return seen

    # ---- capture
    def _evict(self):
        while len(self.graphs) >= self.max_graphs:
            if not self._warned:
                warnings.warn("more than %d distinct batch structures: the least recently used step graphs are dropped "
                              "and captured again when they come back (raise max_graphs, or run without graphs)"
                              % self.max_graphs)
                self._warned = True
            self.graphs.popitem(last=False)

    def capture(self, batch, noise):
        """Capture fwd_bwd on static inputs: clones of x / positions / the five noise tensors with the batch's own index
        tensors bound as they are (per-structure graph), or the buffers of a capacity bucket filled from the batch.  None
        when the capture failed: the owner runs eagerly (a failed bucket: per-structure graphs) from then on."""
        key = self.bucket_key(batch)
        if key is not None:
            g = self._capture_bucket(key, batch, noise)
            if g is not None or not self.enabled:
                return g
            key = None  # the bucket could not be captured: this batch's own structure
        fp = structure_fingerprint(batch, self.model_3d)
        self._evict()
        sb = Batch(batch.x.clone(), batch.positions.clone(), batch.batch, batch.super_edge_index,
                   getattr(batch, "radius_edge_index", None), batch.num_graphs, getattr(batch, "_sizes", None),
                   getattr(batch, "_canonical", None))
        sn = {k: noise[k].clone() for k in _NOISE_KEYS}
        g = self._capture(sb, sn)
        if g is not None:
            self.graphs[fp] = g
        return g

    def _capture_bucket(self, key, batch, noise):
        from . import bucket as bk
        old = self.graphs.pop(key, None)
        counts = bk.batch_counts(bk.sizes_array(batch), batch._canonical)
        caps = bk.capacities(*counts, B=len(batch._sizes), prev=None if old is None else old["bucket"].caps(),
                             sizes=bk.sizes_array(batch))
        max_n = bk.max_n_class(bk.size_range(batch)[1], None if old is None else old["bucket"].max_n, self.model_3d)
        E_cap = 0
        if self.model_3d == "painn":
            E_cap = bk.edge_capacity(self._edges(batch), len(batch._sizes), None if old is None else old["bucket"].E_cap,
                                     sizes=bk.sizes_array(batch))
        del old  # (its graph and static buffers go before the larger ones are made)
        self._evict()
        from_ds = getattr(batch, "_dataset", None) is not None
        dev = batch.device if from_ds else batch.positions.device
        try:
            bkt = bk.Bucket(dev, len(batch._sizes), caps, batch._canonical,
                            x_cols=batch.x_cols if from_ds else batch.x.size(1), max_n=max_n,
                            n_rbf=getattr(getattr(self.modules[0], "radial_basis", None), "n_rbf", 20),
                            kind=self.model_3d, E_cap=E_cap)
            bkt.fill(batch, counts)
        except (ValueError, RuntimeError) as e:
            warnings.warn("capacity bucket not usable for this batch (%s); per-structure graphs from now on" % e)
            self.bucket_ok = False
            return None
        N, P, S, W = counts
        B = len(batch._sizes)
        f32 = dict(dtype=torch.float32, device=dev)
        sn = {"pos_noise": torch.zeros(bkt.N_cap, 3, **f32), "dist_noise_1": torch.zeros(bkt.S_cap, 1, **f32),
              "dist_noise_2": torch.zeros(bkt.S_cap, 1, **f32),
              "noise_level_1": torch.zeros(B, dtype=torch.long, device=dev),
              "noise_level_2": torch.zeros(B, dtype=torch.long, device=dev)}
        g0 = dict(bucket=bkt, noise=sn, counts=counts)
        self.copy_noise(g0, noise)
        g = self._capture(bkt.batch, sn)
        if g is None:
            self.enabled, self.bucket_ok = True, False  # (eager fallback is for a failed per-structure capture)
            return None
        g.update(bucket=bkt, counts=counts)
        self.graphs[key] = g
        return g

    def _capture(self, sb, sn):
        import time
        timing = [] if _env("GEOSSL_CAPTURE_TIMING") else None

        def lap(name):
            if timing is not None:
                torch.cuda.synchronize()
                timing.append((name, time.perf_counter()))
        lap("start")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the capture: builds the cached layouts, sets kernel attributes
            for _ in range(2 if not self.graphs else 1):
                self.fwd_bwd(sb, sn)
        torch.cuda.current_stream().wait_stream(side)
        # nothing may be pending on the device when the capture starts (in a multi-rank job the collective's
        # watchdog thread polls events of earlier all-reduces), and calls of other threads must not invalidate it
        torch.cuda.synchronize()
        lap("warm-up passes")
        graph, graph_bwd = torch.cuda.CUDAGraph(), None
        # No garbage collection while a capture is open: a collection that happens to run there finalises whatever old
        # graphs, events and streams sit in unreachable cycles (an engine of a model that is gone), and destroying a
        # graph or querying an event inside a capture aborts the process.  (Not collecting up front either: a full
        # collection of a process with torch loaded costs 45 ms, as much as the rest of the capture.)
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        _OWN_CAPTURE[0] += 1   # (the owner's passes leave the gradient buffer to `refresh` only inside THIS capture)
        try:
            if self.split is None:
                with torch.cuda.graph(graph, pool=self.pool, capture_error_mode="thread_local"):
                    loss = self.fwd_bwd(sb, sn)
                if self.pool is None:
                    self.pool = graph.pool()
            else:
                cap = torch.cuda.Stream()  # both captures on ONE stream: autograd runs a node's backward where its forward ran
                with torch.cuda.graph(graph, pool=self.pool, stream=cap, capture_error_mode="thread_local"):
                    live = self.split[0](sb, sn)
                if self.pool is None:
                    self.pool = graph.pool()
                graph_bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_bwd, pool=self.pool, stream=cap, capture_error_mode="thread_local"):
                    self.split[1](live)
                loss = live.detach()
                del live
        except Exception as e:  # capture is an optimisation: fall back to eager execution, loudly
            warnings.warn("HIP-graph capture of the DDM step failed (%s: %s); running eagerly" % (type(e).__name__, e))
            torch.cuda.synchronize()
            self.enabled = False
            return None
        finally:
            _OWN_CAPTURE[0] -= 1
            if gc_was_on:
                gc.enable()
        lap("capture + instantiate")
        if timing is not None:
            print("capture timing (ms):", ", ".join("%s %.1f" % (n, 1e3 * (t - timing[i][1])) for i, (n, t) in enumerate(timing[1:])))
        self.captures += 1
        return dict(graph=graph, graph_bwd=graph_bwd, batch=sb, noise=sn, loss=loss, zero=self.zero_with_refresh)

    # ---- per-step refresh of a graph's static inputs
    @staticmethod
    def noise_views(g, batch=None):
        """The graph's five static noise tensors at the size of THIS batch (a bucket's are capacity-sized: the real rows
        are the leading ones) - what a caller's draws are written into."""
        bkt = g.get("bucket")
        if bkt is None:
            return g["noise"]
        N, P, S, W = g["counts"]
        sn = g["noise"]
        return {"pos_noise": sn["pos_noise"][:N], "dist_noise_1": sn["dist_noise_1"][:S],
                "dist_noise_2": sn["dist_noise_2"][:S], "noise_level_1": sn["noise_level_1"],
                "noise_level_2": sn["noise_level_2"]}

    @staticmethod
    def copy_noise(g, noise):
        into = StepGraphs.noise_views(g)
        for k in _NOISE_KEYS:
            into[k].copy_(noise[k].view_as(into[k]))

    @staticmethod
    def refresh(g, batch, noise=None):
        """x, positions and (if given) the five noise tensors of this step into the graph's static inputs; a bucket also
        gets the batch's index structures and real counts."""
        bkt, zero = g.get("bucket"), g.get("zero")
        if bkt is not None:
            try:
                bkt.fill(batch, g["counts"], zero=zero)
            except ValueError as e:  # (tensors the bucket cannot copy: the caller runs this step eagerly)
                warnings.warn("capacity bucket refused a batch (%s); the step runs eagerly" % e)
                return False
        elif getattr(batch, "_dataset", None) is not None:
            # molecules of a device-resident dataset: gathered straight into the graph's static x / positions
            from .layout import get_layout
            sb = g["batch"]
            batch._dataset.gather_into(batch, sb.x, sb.positions, get_layout(sb.batch).mol_ptr, zero)
        else:
            dx, dp, sx, sp = g["batch"].x, g["batch"].positions, batch.x, batch.positions
            if (sx.is_cuda and sp.is_cuda and sx.dtype == dx.dtype and sp.dtype == dp.dtype and sx.shape == dx.shape
                    and sp.shape == dp.shape and all(t_.is_contiguous() for t_ in (dx, dp, sx, sp))):
                if zero is None:
                    call("geossl_copy2", ptr(dx), ptr(sx), dx.numel() * dx.element_size(), ptr(dp), ptr(sp),
                         dp.numel() * dp.element_size(), stream())  # one launch instead of two copies
                else:
                    cb = _lib.CopyBatch()
                    for k, (d_, s_, nb) in enumerate(((dx, sx, dx.numel() * dx.element_size()),
                                                      (dp, sp, dp.numel() * dp.element_size()),
                                                      (zero, None, zero.numel() * 4))):
                        cb.dst[k], cb.src[k], cb.bytes[k] = ptr(d_), (ptr(s_) if s_ is not None else None), nb
                    call("geossl_copy_n", C.byref(cb), 3, stream())
            else:
                dx.copy_(sx)
                dp.copy_(sp)
                if zero is not None:
                    zero.zero_()
        if noise is not None:
            StepGraphs.copy_noise(g, noise)
        return True


_PINNED = {"i": 0, "slots": [[None, None], [None, None]]}  # ring of two pinned staging buffers (grow-only) + copy events


def _host_normal_into(dst, mu, sigma):
    """torch.normal(mu, sigma, size) on the HOST generator (the reference's draw, pretrain_GeoSSL.py:72: same stream, same
    values) into the device tensor `dst` without stalling the host: the draw is made straight into pinned memory and
    copied asynchronously (a pageable source makes the copy wait for everything queued on the stream before it - in the
    reference loop that is the whole previous backward)."""
    key, n = tuple(dst.shape), dst.numel()
    ring = _PINNED
    slot = ring["slots"][ring["i"]]
    ring["i"] ^= 1
    if slot[1] is not None:
        slot[1].synchronize()  # the copy that last read this staging buffer (two steps ago)
    if slot[0] is None or slot[0].numel() < n:
        # one buffer for every size (a shuffled loader's batches all differ: pinning memory per shape cost a step's time)
        slot[0] = torch.empty(max(n + n // 4, 4096), dtype=torch.float32).pin_memory()
    stage = slot[0][:n].view(key)
    torch.normal(mu, sigma, size=key, out=stage)   # (the values of a fresh tensor of this size: the generator fills in order)
    dst.copy_(stage, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record()


def draw_step_noise(batch, n1, n2, mu, sigma, device_noise, given=None, into=None):
    """The five random draws of a step in the order the eager path makes them: perturb (pretrain_GeoSSL.py:72: a
    host draw copied to the device, or a device draw with device_noise), then per head the noise level (NCSN.py:190)
    and the distance noise (:194).  Entries of `given` are used instead of drawing; with `into` the results are written
    into those tensors (the static inputs of a graph) instead of new ones."""
    given = given or {}
    if getattr(batch, "_dataset", None) is not None:   # (a dataset handle knows its counts without collated tensors)
        dev, S, B = batch.device, batch.n_super, batch.num_graphs
    else:
        dev = batch.positions.device
        S, B = batch.super_edge_index.size(1), batch.num_graphs
    out = {}

    def put(key, make, fill):
        if given.get(key) is not None:
            if into is not None:
                into[key].copy_(given[key].view_as(into[key]))
            out[key] = into[key] if into is not None else given[key]
        elif into is not None:
            fill(into[key])
            out[key] = into[key]
        else:
            out[key] = make()

    if device_noise:
        put("pos_noise", lambda: torch.empty_like(batch.positions).normal_(mu, sigma), lambda t_: t_.normal_(mu, sigma))
    else:  # the reference's own draw (:72): CPU generator, then the copy
        put("pos_noise", lambda: torch.normal(mu, sigma, size=batch.positions.size()).to(dev),
            lambda t_: _host_normal_into(t_, mu, sigma))
    for k, head in (("1", n1), ("2", n2)):
        K = head.sigmas.size(0)
        put("noise_level_" + k, lambda: torch.randint(0, K, (B,), device=dev), lambda t_: t_.random_(0, K))
        put("dist_noise_" + k, lambda: torch.randn(S, 1, device=dev), lambda t_: t_.normal_())
    return out


def _next_noise_key(dev):
    """A 64-bit Philox key from the state of torch's CUDA generator of this device, taken on the HOST (no launch): the
    generator's seed mixed with its offset, which is then advanced like a draw of four values would - so
    torch.cuda.manual_seed governs the stream and other torch draws in between move it, exactly as when the key was
    drawn on the device."""
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    off = int(gen.get_offset())
    gen.set_offset(off + 4)
    m64 = (1 << 64) - 1
    x = (int(gen.initial_seed()) + 0x9E3779B97F4A7C15 * (off // 4 + 1)) & m64     # splitmix64 of (seed, draw number)
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m64
    return x ^ (x >> 31)


def draw_step_noise_fused(batch, n1, n2, mu, sigma, into=None):
    """The five draws of a step for a caller that owns its random stream (DDMTrainer with device noise): ONE 64-bit key
    derived on the host from torch's CUDA generator (so torch.cuda.manual_seed governs the stream), then one launch that
    fills all five tensors (geossl_ddm_noise_seeded: Philox under that key) - one launch where the torch calls are five.  New tensors, or
    in place into `into` (the static inputs of a graph)."""
    dev = batch.positions.device
    N, S, B = batch.positions.size(0), batch.super_edge_index.size(1), batch.num_graphs
    if into is None:
        into = {"pos_noise": torch.empty(N, 3, dtype=torch.float32, device=dev),
                "noise_level_1": torch.empty(B, dtype=torch.long, device=dev),
                "dist_noise_1": torch.empty(S, 1, dtype=torch.float32, device=dev),
                "noise_level_2": torch.empty(B, dtype=torch.long, device=dev),
                "dist_noise_2": torch.empty(S, 1, dtype=torch.float32, device=dev)}
    for k in _NOISE_KEYS:
        assert into[k].is_contiguous()
    call("geossl_ddm_noise_seeded", _next_noise_key(dev), float(mu), float(sigma), 3 * N, S, B, n1.sigmas.size(0), n2.sigmas.size(0),
         ptr(into["pos_noise"]), ptr(into["noise_level_1"]), ptr(into["dist_noise_1"]), ptr(into["noise_level_2"]),
         ptr(into["dist_noise_2"]), stream())
    return into


class _ReplayedLoss(torch.autograd.Function):
    """The loss of a replayed step as a differentiable function of the parameters: forward returns the loss the forward
    graph computed, backward waits for the backward graph (replayed on the engine's side stream right behind the forward)
    and hands autograd its gradients, scaled by the upstream gradient - one launch; every parameter's gradient is a view
    of the product."""

    @staticmethod
    def forward(ctx, loss, engine, ticket, *params):
        ctx.engine, ctx.ticket = engine, ticket
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, gout):
        eng, ticket = ctx.engine, ctx.ticket
        src = eng.collect(ticket)
        pre = ticket.pop("views", None)
        if pre is not None:
            # the step's own output buffer and its per-parameter views were made in do_DDM, while the host was going to
            # wait for the forward anyway; handing them out once keeps them stealable (AccumulateGrad adopts a gradient
            # nobody else references - the list dies with this frame)
            buf, outs = pre
            torch.mul(src, gout, out=buf)
            return (None, None, None) + tuple(outs)
        g = src * gout
        # one split for all parameters, then a reshape each (two tensor operations per parameter were a third of this
        # backward's host time at 59 parameters - and the host is what bounds the loop at small batches)
        outs = [v if len(shape) == 1 else v.view(shape) for v, (shape, _) in zip(g.split_with_sizes(eng.numels), eng.shapes)]
        return (None, None, None) + tuple(outs)


class _StepLoss(torch.Tensor):
    """The loss do_DDM's graph path hands to a caller that owns its optimizer - an ordinary tensor attached to the autograd
    graph (``_ReplayedLoss`` with every parameter as an input: ``torch.autograd.grad``, scaled losses, sums with other
    terms all differentiate through it as before) whose plain ``loss.backward()`` (pretrain_GeoSSL.py:259) does not go
    through the autograd engine: the step's gradients are already in the engine's buffer, so backward() copies them into
    the step's own buffer (one launch) and points every parameter's ``.grad`` at its view - what the 59 AccumulateGrad
    nodes of the engine path do one queue round trip each (0.34 ms per step on the host, which is what bounds the loop at
    the reference's batch size).  Anything else - a gradient argument, retain_graph, create_graph, inputs=, a parameter
    that already holds a gradient (accumulation over several backward() calls) or carries a tensor hook, an arithmetic
    result (ops on this class return plain tensors) - takes the engine.  GEOSSL_NO_DIRECT_BACKWARD: always the engine."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        st = self.__dict__.pop("_geossl_step", None)
        if (st is not None and gradient is None and not retain_graph and not create_graph and inputs is None
                and st[0].direct_backward(st[1])):
            return None
        return torch.Tensor.backward(self, gradient, retain_graph, create_graph, inputs)


class _Ticket(dict):
    """The claim of one do_DDM step on the gradients its backward replay leaves in the engine's buffer ("serial",
    "event", "g", "used").  The replay runs on a side stream and READS the parameters: when a loss is dropped without
    backward() (a skipped non-finite step followed by an EMA update, load_state_dict, another loss's optimizer step) the
    caller's stream is made to wait for the replay here, so that no later parameter write can overtake it."""

    def __del__(self):
        try:
            _restore_backward_threads(self)
        except Exception:
            pass
        try:
            if (not self.get("used") and self.get("event") is not None and not torch.cuda.is_current_stream_capturing()
                    and not self["event"].query()):
                torch.cuda.current_stream().wait_event(self["event"])
        except Exception:  # interpreter shutdown, a destroyed context: nothing left to order
            pass


class _AutogradStep:
    """do_DDM's graph path for a caller that owns its optimizer (the reference loop, pretrain_GeoSSL.py:249-260).

    Forward and backward of the step are captured as TWO graphs per index structure (one memory pool, one capture
    stream).  ``run`` draws the step's noise like the eager path (same generators, same order), refreshes the graph
    inputs, replays the forward on the caller's stream and the backward on a side stream right behind it, and returns
    the loss behind ``_ReplayedLoss``.  The reference's ``loss.detach().item()`` (:255) therefore waits for the forward
    only; ``optimizer.zero_grad()``, ``loss.backward()`` (one wait + one multiply; AccumulateGrad adopts the views) and
    the launches of ``optimizer.step()`` are issued by the host WHILE the backward graph runs.  The backward's gradients
    land in a flat static buffer (the parameters' .grad point into it only while a step is captured); a step whose
    backward() has not been called when the next step arrives keeps a snapshot of them."""

    def __init__(self, model, n1, n2):
        from .NCSN import _head_params
        self.model, self.n1, self.n2 = model, n1, n2
        # the parameters the step reaches (a parameter outside it - an atomref table, PaiNN's output layers - gets no
        # gradient at all, like in the eager path, not a zero one)
        backbone = model._params() if hasattr(model, "_params") else _schnet_step_params(model)
        seen, self.params = set(), []
        for p in list(backbone) + _head_params(n1) + _head_params(n2):
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                self.params.append(p)
        dev = self.params[0].device
        # the parameters move into ONE flat buffer (values unchanged; they stay ordinary leaves): the gradients this
        # engine hands to autograd are views of one flat buffer at the same offsets, which lets a stock torch.optim.Adam
        # over these parameters run as one launch (optim.ParamHome and its step hook)
        from .optim import ParamHome
        self.home = ParamHome.of(self.params)
        self.gflat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.gflat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.shapes = tuple((tuple(p.shape), p.numel()) for p in self.params)
        self.numels = [n for _, n in self.shapes]
        # where every parameter of the three modules hangs (submodule, name) with its object, address and grad flag:
        # `unchanged()` compares them per call without walking the module trees
        self._where = []
        for m in (self.model, self.n1, self.n2):
            seen_mod = set()
            for sub in m.modules():
                if id(sub) in seen_mod:
                    continue
                seen_mod.add(id(sub))
                for name, p in sub._parameters.items():
                    if p is not None:
                        self._where.append((sub._parameters, name, p, p.data_ptr(), p.requires_grad))
        self.graphs = {}   # (model_3d, normalize) -> StepGraphs
        self._cfg = None
        self._side = torch.cuda.Stream(device=dev)   # the backward graphs replay here
        self._bwd_done = None                        # event behind the last backward replay
        self._ticket = None                          # the step whose gradients are in gflat: {"serial", "event", "g"}
        self._serial = 0

    def unchanged(self):
        """No parameter of the three modules was replaced, moved or (un)frozen since the engine was built (its graphs bind
        the parameters' addresses)."""
        for d, name, p, addr, rg in self._where:
            if d.get(name) is not p or p.data_ptr() != addr or p.requires_grad != rg:
                return False
        return True

    # The engine hangs on the backbone module (model.__dict__) and holds graphs, streams and events: a copy or a pickle
    # of the model (copy.deepcopy for an EMA twin, torch.save(model)) carries None instead and builds its own on first use.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def _fwd(self, batch, noise):
        args, mu, sigma = self._cfg
        return _do_ddm_eager(args, batch, self.model, mu, sigma, (self.n1, self.n2), noise, True, True)

    def _bwd(self, loss):
        held = [p.grad for p in self.params]
        # tensor hooks of the caller (gradient clipping, logging, reducers) belong to ITS backward, which runs later on
        # the node `run` returns; the passes in here are internal - and in direct mode the nodes hand autograd no
        # gradients at all, so a hook would be called with None
        hooks = []
        for p in self.params:
            h = getattr(p, "_backward_hooks", None)
            if h:
                hooks.append((h, list(h.items())))
                h.clear()
        if not own_capture_open():
            self.gflat.zero_()  # (a replayed step: cleared by the launch that refreshes the graph's inputs, StepGraphs.refresh)
        for p, v in zip(self.params, self.views):
            p.grad = v
        try:
            with _lib.direct_grads():  # kernels accumulate straight into the static buffer
                loss.backward(self._one)  # (a standing 1.0: backward() would fill a new one, a launch per step)
            for p, v in zip(self.params, self.views):  # anything autograd replaced goes back into the buffer
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
        finally:
            for p, h in zip(self.params, held):
                p.grad = h
            for h, items in hooks:
                h.update(items)

    def _fwd_bwd(self, batch, noise):
        loss = self._fwd(batch, noise)
        self._bwd(loss)
        return loss.detach()

    def collect(self, ticket):
        """The gradients of the step `ticket` stands for (unscaled, read-only): the snapshot taken when a later step was
        about to overwrite them, else the static buffer itself once its backward replay is complete."""
        if ticket["g"] is not None:
            return ticket["g"]
        if ticket["serial"] != self._serial:
            raise RuntimeError("the gradients of this do_DDM step are gone (a later step replaced them after this step's "
                               "backward had already run once); call backward(retain_graph=...) before the next do_DDM")
        torch.cuda.current_stream().wait_event(ticket["event"])
        ticket["used"] = True
        return self.gflat

    def direct_backward(self, ticket):
        """loss.backward() of the step `ticket` stands for without the autograd engine (_StepLoss): True when done."""
        if _env("GEOSSL_NO_DIRECT_BACKWARD") or ticket.get("used") or "views" not in ticket:
            return False
        # hooks on the parameters' AccumulateGrad NODES (DistributedDataParallel's reducer) cannot be seen from here: with
        # more than one rank initialised the engine runs, whoever reduces the gradients
        dist = torch.distributed
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
        for p in self.params:
            # (an existing gradient: accumulate like AccumulateGrad would; a tensor hook: call it like the engine would)
            if p.grad is not None or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        src = self.collect(ticket)   # (the caller's stream waits for the backward replay)
        buf, outs = ticket.pop("views")
        buf.copy_(src)               # d loss / d loss = 1: the engine path multiplies by it, same values
        for p, v in zip(self.params, outs):
            p.grad = v
        return True

    def run(self, args, batch, mu, sigma, noise, device_noise):
        if getattr(batch, "_dataset", None) is None and (not batch.positions.is_cuda or batch.positions.requires_grad):
            return None
        key = (args.model_3d, bool(getattr(args, "normalize", False)))
        sg = self.graphs.get(key)
        if sg is None:
            sg = self.graphs[key] = StepGraphs(self._fwd_bwd, args.model_3d, split=(self._fwd, self._bwd),
                                               mode=getattr(args, "step_graph_mode", "auto"), normalize=key[1],
                                               modules=(self.model, self.n1, self.n2))
            sg.zero_with_refresh = self.gflat
        if not sg.enabled:
            return None
        self._cfg = (Args(args.model_3d, key[1]), mu, sigma)
        g = sg.lookup(batch)
        if g is None and not sg.capture_now(batch):
            # first sighting of an index structure that only its own graph can serve (sampled tuples, PaiNN edge lists,
            # batches built by hand): eager - a loader that never repeats such a structure never pays for a capture, one
            # that comes back is captured on its second step.  Ragged SchNet batches of a loader share ONE capacity-bucket
            # graph and equal-sized molecules one per-structure graph: those are captured right away.
            return None
        main = torch.cuda.current_stream()
        if self._bwd_done is not None:
            # the last backward replay reads the activations and writes the gradient buffer this step is about to reuse
            main.wait_event(self._bwd_done)
            t = self._ticket
            if t is not None and t["g"] is None and not t.get("used"):
                t["g"] = self.gflat.clone()  # a step still waiting for its backward() keeps its gradients
        if g is None:
            drawn = draw_step_noise(batch, self.n1, self.n2, mu, sigma, device_noise, noise)
            g = sg.capture(batch, drawn)
            if g is None:
                return None
            if not sg.refresh(g, batch, drawn):
                return None
        else:
            if not sg.refresh(g, batch):
                return None  # (the bucket refused the batch's tensors: this step as eager launches)
            draw_step_noise(batch, self.n1, self.n2, mu, sigma, device_noise, noise, into=sg.noise_views(g))
        g["graph"].replay()            # forward: the loss is on the device when this is done
        loss = g["loss"].clone()
        fwd_done = torch.cuda.Event()
        fwd_done.record(main)
        self._side.wait_event(fwd_done)
        with torch.cuda.stream(self._side):  # backward: behind the forward, beside whatever the host queues next
            g["graph_bwd"].replay()
            self._bwd_done = torch.cuda.Event()
            self._bwd_done.record(self._side)
        self._serial += 1
        prev = self._ticket
        if prev is not None:
            _restore_backward_threads(prev)  # (a loss that never saw its backward(): its switch goes back first)
        self._ticket = _Ticket(serial=self._serial, event=self._bwd_done, g=None)
        _single_thread_backward(self._ticket)
        buf = torch.empty_like(self.gflat)   # this step's (scaled) gradients as autograd will receive them
        # (a one-dimensional parameter's piece of the split already has its shape: no view for the biases - half the list)
        self._ticket["views"] = (buf, [v if len(shape) == 1 else v.view(shape)
                                       for v, (shape, _) in zip(buf.split_with_sizes(self.numels), self.shapes)])
        # deferred index check of the backbone: the embedding kernel flagged an out-of-range atom type in the status
        # word, but model.forward's own arm() does nothing while a graph is captured and a replay never reaches it -
        # read the word here like DDMTrainer.step does (IndexError up to eight steps late, like Embedding's, not never)
        st = self.model.__dict__.get("_geossl_status")
        if st is not None:
            st.poll()
            st.arm(every=8)
        out = _ReplayedLoss.apply(loss, self, self._ticket, *self.params).as_subclass(_StepLoss)
        out._geossl_step = (self, self._ticket)
        return out


_MT_PENDING = []   # tickets of this process whose autograd thread switch is still to be put back


def _single_thread_backward(ticket):
    """The loss of a replayed step has ONE node behind it (_ReplayedLoss) and its 59 AccumulateGrads; autograd would hand
    them to its per-device worker thread and wake the caller when it is done - a hand-over that costs 0.13 ms per
    backward() (measured, tools/ref_loop_profile.py), a fifth of a whole step at the reference's batch size.  With
    multithreading off the calling thread runs the backward itself.  The switch is autograd's own, thread-local
    (torch.autograd.set_multithreading_enabled); it is turned off when do_DDM hands out the loss and put back to what the
    caller had at the optimizer step that follows (a global optimizer post-step hook, optim.py) or, without one, at the
    next do_DDM - it cannot be put back from inside the backward: the engine restores the thread state it saw at
    backward()'s entry when the pass ends.  Other autograd work of the caller after the step - graphs that span several
    devices - sees its own setting.  GEOSSL_KEEP_AUTOGRAD_THREADS leaves the switch alone."""
    if _env("GEOSSL_KEEP_AUTOGRAD_THREADS"):
        return
    if torch.autograd.is_multithreading_enabled():
        torch.autograd.set_multithreading_enabled(False)
        # a small token, not the ticket: the pending list must not keep a step's gradient buffer alive, and the switch is
        # THREAD-LOCAL - it is put back only by the thread that turned it off
        token = ticket["mt_token"] = _MtToken(threading.get_ident())
        _MT_PENDING.append(token)


class _MtToken:
    __slots__ = ("thread", "live")

    def __init__(self, thread):
        self.thread, self.live = thread, True


def _restore_backward_threads(ticket=None):
    """Put autograd's multithreading switch back for `ticket` (None: for every step of THIS thread that is still
    pending).  A call from another thread (a ticket finalised by the garbage collector there) leaves the token pending:
    the owning thread's next step / optimizer hook restores it."""
    me = threading.get_ident()
    if ticket is not None:
        token = ticket.pop("mt_token", None) if isinstance(ticket, dict) else None
        todo = [] if token is None else [token]
    else:
        todo = [t for t in _MT_PENDING if t.thread == me]
    for t in todo:
        if t.live and t.thread == me:
            t.live = False
            torch.autograd.set_multithreading_enabled(True)
    _MT_PENDING[:] = [p for p in _MT_PENDING if p.live]


def _schnet_step_params(model):
    from .Geom3D.models.schnet import _core_params
    return _core_params(model)


def _autograd_step(model, n1, n2):
    """The _AutogradStep of a (backbone, head, head) triple, kept on the backbone module; rebuilt when a parameter was
    replaced, moved or frozen since (the graphs bind parameter addresses)."""
    eng = model.__dict__.get("_geossl_autograd_step")
    if eng is None or eng.n1 is not n1 or eng.n2 is not n2 or not eng.unchanged():
        eng = _AutogradStep(model, n1, n2)
        model.__dict__["_geossl_autograd_step"] = eng
    return eng


class DDMTrainer:
    """The body of ``train()`` (pretrain_GeoSSL.py:234-260) for the DDM option: forward of both
    views + both heads, backward, gradient all-reduce, Adam — flat parameter buffer, no host sync
    inside ``step`` (the reference's per-step ``loss.item()`` at :255 is logging, call
    ``float(loss)`` outside the timed region if wanted).

    ``use_graph=True``: forward + backward are captured once per index structure into a HIP graph and replayed;
    positions, atom types and the five noise tensors are copied into the graph's static buffers before each replay
    (with ``noise=None`` and ``device_noise=True`` the trainer makes the five draws itself, straight into those
    buffers).  Which graph a batch gets is decided by ``structure_fingerprint`` - the molecule sizes for batches whose
    index tensors are a function of them, the tensor objects otherwise - never by the caller: a shuffled loader with
    ragged molecules captures one graph per distinct size sequence (up to ``max_graphs``, least recently used dropped,
    all in one shared memory pool).  The all-reduce and the Adam launch stay outside the graph."""

    def __init__(self, model, ncsn_01, ncsn_02, lr=5e-4, weight_decay=0.0, mu=0.0, sigma=0.3, model_3d="schnet",
                 device_noise=True, use_graph=False, overlap_heads=True, max_graphs=256, graph_mode="auto"):
        from .optim import FlatParams, FusedAdam
        from .parallel import GradAllReduce
        self.model, self.n1, self.n2 = model, ncsn_01, ncsn_02
        self.args = Args(model_3d)
        self.mu, self.sigma = mu, sigma
        self.device_noise = device_noise
        self.flat = FlatParams([model, ncsn_01, ncsn_02])
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.reduce = GradAllReduce(self.flat.grad)
        self.use_graph = use_graph
        # two-pass NCSN backward only (GEOSSL_NCSN_SPLIT_BWD): its weight-gradient kernels on a side stream, concurrent
        # with the backbone's backward; the default one-pass backward has nothing to overlap
        self.overlap_heads = overlap_heads
        # graph_mode "auto": ragged SchNet batches share one capacity-bucket graph per batch size, equal-sized molecules
        # one graph per size, anything else one per structure from its second sighting on; "structure": one graph per
        # structure fingerprint, captured at first sight (StepGraphs)
        self.step_graphs = StepGraphs(self._fwd_bwd, model_3d, max_graphs, mode=graph_mode,
                                      modules=(model, ncsn_01, ncsn_02))
        self.step_graphs.zero_with_refresh = self.flat.grad
        self._zero_outside = True
        self._one = torch.ones((), dtype=torch.float32, device=self.flat.grad.device)
        self._side = None

    @property
    def _graphs(self):
        return self.step_graphs.graphs

    def _fwd_bwd(self, batch, noise):
        from . import NCSN as _ncsn
        if not (self._zero_outside and own_capture_open()):
            self.flat.zero_grad()  # (a replayed step: cleared with the refresh of the graph's inputs, StepGraphs.refresh)
        if noise is None and self.device_noise:
            noise = self._draw_noise(batch)  # the trainer's own stream, eager launches or replayed graph alike
        loss = _do_ddm_eager(self.args, batch, self.model, self.mu, self.sigma, (self.n1, self.n2), noise, True,
                             self.device_noise)
        if self._side is None and self.overlap_heads:
            self._side = torch.cuda.Stream()
        _ncsn.set_side_stream(self._side)  # head weight gradients overlap the backbone's backward
        try:
            with _lib.direct_grads():  # every p.grad is a view of self.flat.grad: kernels accumulate into it directly
                loss.backward(self._one)  # (a standing 1.0: backward() would fill a new one, a launch per step)
        finally:
            _ncsn.set_side_stream(None)
            _ncsn.join_side_stream()
        self.flat.rebind_grads()
        return loss.detach()

    _NOISE_KEYS = _NOISE_KEYS

    def _graph_fwd_bwd(self, batch, noise):
        sg = self.step_graphs
        g = sg.lookup(batch)
        own_noise = noise is None
        if g is None:
            if not sg.capture_now(batch):  # a structure only its own graph can serve, seen for the first time: eager
                return self._fwd_bwd(batch, noise)
            if own_noise:
                noise = self._draw_noise(batch)  # this step's draws (the capture needs tensors to clone)
            g = sg.capture(batch, noise)
            if g is None:  # capture failed: eager from now on
                self.use_graph = False
                return self._fwd_bwd(batch, noise)
            own_noise = False  # already drawn: copied below like a caller's
        if not sg.refresh(g, batch, None if own_noise else noise):
            return self._fwd_bwd(batch, noise)  # (the bucket refused the batch's tensors: this step as eager launches)
        if own_noise:  # the step's own draws go straight into the graph's static inputs (no staging copies)
            self._draw_noise(g["batch"], into=g["noise"])  # (g["batch"]: a bucket's draws cover its capacity)
        g["graph"].replay()
        # (a clone: the static scalar is overwritten by the next replay, and freed with its graph when that is dropped)
        return g["loss"].clone()

    def _draw_noise(self, batch, into=None):
        """The five random draws of a step on the device (perturb: pretrain_GeoSSL.py:72; the heads: NCSN.py:190,194),
        as tensors - new ones, or in place into `into`."""
        if _env("GEOSSL_TORCH_DRAWS"):  # the five torch calls (the form before geossl_ddm_noise)
            return draw_step_noise(batch, self.n1, self.n2, self.mu, self.sigma, True, None, into)
        return draw_step_noise_fused(batch, self.n1, self.n2, self.mu, self.sigma, into)

    def step(self, batch, noise=None, structure_key=None):
        """One training step.  ``structure_key`` is accepted for compatibility and ignored: graphs are found by the
        batch's own fingerprint."""
        if self.use_graph and ((noise is None and self.device_noise)
                               or (noise is not None and all(k in noise for k in _NOISE_KEYS))):
            loss = self._graph_fwd_bwd(batch, noise)
        else:
            loss = self._fwd_bwd(batch, noise)
        st = self.model.__dict__.get("_geossl_status")
        if st is not None:  # deferred index check of the backbone (a replayed graph cannot queue the host copy itself)
            st.poll()
            st.arm(every=8)  # an out-of-range atom type surfaces up to eight steps late
        scale = self.reduce()
        self.opt.step(grad_scale=scale)
        return loss
