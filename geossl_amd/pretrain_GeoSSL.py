"""Step API of ``pretrain_GeoSSL.py --GeoSSL_option=DDM`` (examples/pretrain_GeoSSL.py:68-74,179-212,
234-260) on the HIP path.

``perturb`` and ``do_DDM`` keep the reference signatures.  The reference reads two module globals
(``NCSN_model_01/02``, :207-208); here they are module attributes with the same names, or can be
passed with ``NCSN_models=``.  ``DDMTrainer`` is the training-loop body (:234-260) with all
parameters in one flat buffer: one fused Adam launch and one RCCL all-reduce per step.
"""
import torch

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
    here is one concatenation."""

    @staticmethod
    def forward(ctx, h, n):
        ctx.n = n
        return h[:n], h[n:]

    @staticmethod
    def backward(ctx, g1, g2):
        return torch.cat([g1, g2]), None


def _two_view_batch(batch_vec, num_graphs):
    """batch ids of the concatenated (clean ‖ perturbed) 2B-molecule batch + its layout, cached on
    the batch tensor object."""
    cached = getattr(batch_vec, "_geossl_two_view", None)
    if cached is None or cached[2] != batch_vec._version:
        b2 = torch.cat([batch_vec, batch_vec + num_graphs])
        lay2 = MolLayout(b2, 2 * num_graphs)
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
           fuse_views=True, device_noise=False):
    """pretrain_GeoSSL.py:179-212 -> (loss, 0).

    noise: optional dict with pos_noise [N,3], noise_level_1/2 [B] int64, dist_noise_1/2 [S,1] —
    the five random draws of the step; anything missing is drawn like the reference does.
    fuse_views: run the clean and the perturbed view through the backbone as one 2B-molecule batch
    (molecules never interact, so every row sees the same arithmetic as in two separate calls).
    """
    noise = noise or {}
    n1, n2 = NCSN_models if NCSN_models is not None else (NCSN_model_01, NCSN_model_02)
    if n1 is None or n2 is None:
        raise RuntimeError("set geossl_amd.pretrain_GeoSSL.NCSN_model_01/02 or pass NCSN_models=(m1, m2)")
    positions = batch.positions
    x_01 = batch.x[:, 0]
    positions_01 = positions
    x_02, positions_02 = perturb(x_01, positions, mu, sigma, noise=noise.get("pos_noise"), device_noise=device_noise)

    if args.model_3d == "schnet":
        if fuse_views:
            num_graphs = batch.num_graphs
            b2, lay2 = _two_view_batch(batch.batch, num_graphs)
            N = positions.size(0)
            # the readout is dead compute in this step (`_` at pretrain_GeoSSL.py:187): not evaluated
            _, h = model(torch.cat([x_01, x_02]), torch.cat([positions_01, positions_02]), b2, return_latent=True,
                         layout=lay2, latent_only=True)
            molecule_3D_repr_01, molecule_3D_repr_02 = _SplitViews.apply(h, N)
        else:
            _, molecule_3D_repr_01 = model(x_01, positions_01, batch.batch, return_latent=True)
            _, molecule_3D_repr_02 = model(x_02, positions_02, batch.batch, return_latent=True)
    elif args.model_3d == "painn":
        if fuse_views:
            b2, e2 = _two_view_edges(batch.batch, batch.radius_edge_index, batch.num_graphs)
            N = positions.size(0)
            _, h = model(torch.cat([x_01, x_02]), torch.cat([positions_01, positions_02]), e2, b2, return_latent=True)
            molecule_3D_repr_01, molecule_3D_repr_02 = _SplitViews.apply(h, N)
        else:
            _, molecule_3D_repr_01 = model(x_01, positions_01, batch.radius_edge_index, batch.batch, return_latent=True)
            _, molecule_3D_repr_02 = model(x_02, positions_02, batch.radius_edge_index, batch.batch, return_latent=True)
    else:
        raise Exception("3D model {} not included.".format(args.model_3d))

    if getattr(args, "normalize", False):  # :193-195
        molecule_3D_repr_01 = ops.row_normalize(molecule_3D_repr_01)
        molecule_3D_repr_02 = ops.row_normalize(molecule_3D_repr_02)

    super_edge_index = batch.super_edge_index
    distance_01 = ops.pair_distance(positions_01, super_edge_index[0], super_edge_index[1])  # :199-201
    distance_02 = ops.pair_distance(positions_02, super_edge_index[0], super_edge_index[1])  # :203-205

    # cross-view pairing (:207-208); each head returns 0.5 * its loss so the sum is (l1 + l2) / 2 (:210)
    loss_01 = n1(batch, molecule_3D_repr_01, distance_02, noise_level=noise.get("noise_level_1"),
                 distance_noise=noise.get("dist_noise_1"), out_scale=0.5)
    loss_02 = n2(batch, molecule_3D_repr_02, distance_01, noise_level=noise.get("noise_level_2"),
                 distance_noise=noise.get("dist_noise_2"), out_scale=0.5)
    loss = loss_01 + loss_02
    return loss, 0


class Batch:
    """Device-resident collated batch with the attributes do_DDM / NCSN_version_03 read
    (BatchAtomTuple, dataloaders_AtomTuple.py:40-78)."""

    def __init__(self, x, positions, batch, super_edge_index, radius_edge_index=None, num_graphs=None):
        self.x, self.positions, self.batch, self.super_edge_index = x, positions, batch, super_edge_index
        self.radius_edge_index = radius_edge_index
        self._num_graphs = num_graphs

    @property
    def num_graphs(self):
        if self._num_graphs is None:
            self._num_graphs = self.batch[-1].item() + 1  # dataloaders_AtomTuple.py:75-78
        return self._num_graphs

    @classmethod
    def from_numpy(cls, d, device):
        import numpy as np
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        rei = t(d["radius_edge_index"]) if "radius_edge_index" in d else None
        ng = int(len(d["sizes"])) if "sizes" in d else None
        out = cls(t(d["x"]), t(d["positions"]), t(d["batch"]), t(d["super_edge_index"]), rei, ng)
        if "sizes" in d:  # collation knows the molecule sizes on the host: index structures without a device read-back
            from .layout import prepare_batch
            prepare_batch(out.batch, out.super_edge_index, d["sizes"])
        return out

    def to(self, device):
        mv = lambda a: None if a is None else a.to(device)
        return Batch(mv(self.x), mv(self.positions), mv(self.batch), mv(self.super_edge_index),
                     mv(self.radius_edge_index), self._num_graphs)


class Args:
    """The three fields do_DDM reads from the reference's argparse namespace."""

    def __init__(self, model_3d="schnet", normalize=False):
        self.model_3d = model_3d
        self.normalize = normalize


class DDMTrainer:
    """The body of ``train()`` (pretrain_GeoSSL.py:234-260) for the DDM option: forward of both
    views + both heads, backward, gradient all-reduce, Adam — flat parameter buffer, no host sync
    inside ``step`` (the reference's per-step ``loss.item()`` at :255 is logging, call
    ``float(loss)`` outside the timed region if wanted).

    ``use_graph=True``: forward + backward of batches that share one index structure (same
    ``batch`` / ``super_edge_index`` contents, identified by the caller's ``structure_key``) are
    captured once into a HIP graph and replayed; positions, atom types and the five noise tensors
    are copied into the graph's static buffers before each replay (with ``noise=None`` and ``device_noise=True`` the
    trainer makes the five draws itself, straight into those buffers).  A loader with ragged molecules
    passes one key per batch (e.g. its index in the epoch): up to ``max_graphs`` graphs are kept,
    all in one shared memory pool.  The all-reduce and the Adam launch stay outside the graph."""

    def __init__(self, model, ncsn_01, ncsn_02, lr=5e-4, weight_decay=0.0, mu=0.0, sigma=0.3, model_3d="schnet",
                 device_noise=True, use_graph=False, overlap_heads=True, max_graphs=256):
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
        self._graphs, self._pool, self.max_graphs = {}, None, max_graphs
        self._side = None

    def _fwd_bwd(self, batch, noise):
        from . import NCSN as _ncsn
        self.flat.zero_grad()
        loss, _ = do_DDM(self.args, batch, self.model, None, self.mu, self.sigma, NCSN_models=(self.n1, self.n2),
                         noise=noise, device_noise=self.device_noise)
        if self._side is None and self.overlap_heads:
            self._side = torch.cuda.Stream()
        _ncsn.set_side_stream(self._side)  # head weight gradients overlap the backbone's backward
        try:
            with _lib.direct_grads():  # every p.grad is a view of self.flat.grad: kernels accumulate into it directly
                loss.backward()
        finally:
            _ncsn.set_side_stream(None)
            _ncsn.join_side_stream()
        self.flat.rebind_grads()
        return loss.detach()

    _NOISE_KEYS = ("pos_noise", "noise_level_1", "dist_noise_1", "noise_level_2", "dist_noise_2")

    def _graph_fwd_bwd(self, batch, noise, key):
        # The captured graph binds every index structure of the capture batch (batch vector, super_edge_index, and for
        # PaiNN the precomputed radius_edge_index with its incidence lists, which differ from batch to batch even when
        # the molecule sizes agree): only x, positions and the noise tensors are refreshed before a replay.  The
        # caller's structure_key vouches for batch / super_edge_index; radius_edge_index is identified here by tensor
        # object and version, so a different edge list gets its own capture instead of silently replaying the old one.
        rei = batch.radius_edge_index if self.args.model_3d == "painn" else None
        if rei is not None:
            key = (key, id(rei), rei._version, int(rei.size(1)))
        g = self._graphs.get(key)
        own_noise = noise is None
        if g is None:
            if own_noise:
                noise = self._draw_noise(batch)  # this step's draws (the capture needs tensors to clone)
            g = self._capture(batch, noise, key)
            if g is None:  # capture failed: eager from now on
                return self._fwd_bwd(batch, noise)
            own_noise = False  # already drawn: copied below like a caller's
        g["batch"].x.copy_(batch.x)
        g["batch"].positions.copy_(batch.positions)
        if own_noise:  # the step's own draws go straight into the graph's static inputs (no staging copies)
            self._draw_noise(batch, into=g["noise"])
        else:
            for k in self._NOISE_KEYS:
                g["noise"][k].copy_(noise[k])
        g["graph"].replay()
        return g["loss"]

    def _draw_noise(self, batch, into=None):
        """The five random draws of a step (perturb: pretrain_GeoSSL.py:72 with the draw made on the device; the heads:
        NCSN.py:190,194), as tensors - new ones, or in place into `into`."""
        dev = batch.positions.device
        if into is None:
            S, B = batch.super_edge_index.size(1), batch.num_graphs
            into = {"pos_noise": torch.empty_like(batch.positions),
                    "noise_level_1": torch.empty(B, dtype=torch.long, device=dev),
                    "noise_level_2": torch.empty(B, dtype=torch.long, device=dev),
                    "dist_noise_1": torch.empty(S, 1, dtype=torch.float32, device=dev),
                    "dist_noise_2": torch.empty(S, 1, dtype=torch.float32, device=dev)}
        into["pos_noise"].normal_(self.mu, self.sigma)
        into["noise_level_1"].random_(0, self.n1.sigmas.size(0))
        into["dist_noise_1"].normal_()
        into["noise_level_2"].random_(0, self.n2.sigmas.size(0))
        into["dist_noise_2"].normal_()
        return into

    def _capture(self, batch, noise, key):
        """One HIP graph per structure key.  Ragged batches (every batch its own index structure) get one graph each -
        captured once, replayed every epoch; all graphs share ONE memory pool (their activations are dead between
        steps), so the device memory of N graphs is that of the largest, plus the static inputs and the loss of each."""
        while len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        sb = Batch(batch.x.clone(), batch.positions.clone(), batch.batch, batch.super_edge_index,
                   batch.radius_edge_index, batch.num_graphs)
        sn = {k: noise[k].clone() for k in self._NOISE_KEYS}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up off the capture: builds the cached layouts, sets kernel attributes
            for _ in range(2 if not self._graphs else 1):
                self._fwd_bwd(sb, sn)
        torch.cuda.current_stream().wait_stream(side)
        # nothing may be pending on the device when the capture starts (in a multi-rank job the collective's
        # watchdog thread polls events of earlier all-reduces), and calls of other threads must not invalidate it
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                loss = self._fwd_bwd(sb, sn)
        except Exception as e:  # capture is an optimisation: fall back to eager execution, loudly
            import warnings
            warnings.warn("HIP-graph capture of the DDM step failed (%s: %s); running eagerly" % (type(e).__name__, e))
            torch.cuda.synchronize()
            self.use_graph = False
            return None
        if self._pool is None:
            self._pool = graph.pool()
        g = self._graphs[key] = dict(graph=graph, batch=sb, noise=sn, loss=loss)
        return g

    def step(self, batch, noise=None, structure_key=None):
        if self.use_graph and structure_key is not None and (
                (noise is None and self.device_noise) or (noise is not None and all(k in noise for k in self._NOISE_KEYS))):
            loss = self._graph_fwd_bwd(batch, noise, structure_key)
        else:
            loss = self._fwd_bwd(batch, noise)
        st = self.model.__dict__.get("_geossl_status")
        if st is not None:  # deferred index check of the backbone (a replayed graph cannot queue the host copy itself)
            st.poll()
            st.arm()
        scale = self.reduce()
        self.opt.step(grad_scale=scale)
        return loss
