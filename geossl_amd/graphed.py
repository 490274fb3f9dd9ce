"""Inference through replayed HIP graphs: ``SchNet.forward(z, pos, batch)`` under ``no_grad`` (the evaluation loops of the
reference: examples/finetune_qm9.py:278-384 ``eval()``; BASELINE config 2, forward only) as ONE graph launch per call.

Launched eagerly a forward pass is ~40 C-ABI calls, and at 1024 molecules per batch the host needs longer to issue them
than the GPU to run them (0.31 of 0.53 ms per pass were launch gaps).  A captured pass also takes the layer loop (the
chains and aggregations of the backbone as one launch, ``ops.layer_loop``), which only pays under capture.

A graph binds the batch's index structure: batches whose index tensors are a function of the molecule sizes (host
sizes known, ``pretrain_GeoSSL.structure_fingerprint``) share a graph per size sequence, captured at the first sighting
for equal-sized molecules and at the second for ragged ones (a loader that never repeats a size sequence never pays for
a capture); everything else runs eagerly.  The result is the graph's static output: valid until the next call (clone it
to keep it).
"""
import ctypes as C
from collections import OrderedDict

import torch

from ._lib import call, ptr, stream
from .layout import get_layout


class GraphedForward:
    def __init__(self, model, return_latent=False, max_graphs=64):
        self.model, self.return_latent, self.max_graphs = model, return_latent, max_graphs
        self.graphs, self.pool, self.captures = OrderedDict(), None, 0
        self._seen = OrderedDict()   # fingerprints seen once (bounded: a loader that never repeats a size sequence)
        self.enabled = True

    def _eager(self, x, positions, batch_vec):
        with torch.no_grad():
            return self.model(x[:, 0], positions, batch_vec, return_latent=self.return_latent)

    def __call__(self, batch):
        from . import bucket as bk
        from .pretrain_GeoSSL import structure_fingerprint
        fp = structure_fingerprint(batch, "schnet") if self.enabled and batch.positions.is_cuda else None
        if fp is None or fp[0] != "sizes":
            return self._eager(batch.x, batch.positions, batch.batch)
        g = self.graphs.get(fp)
        if g is None:
            if not bk.is_uniform(batch) and fp not in self._seen:   # ragged: from the second sighting on
                self._seen[fp] = True
                while len(self._seen) > 4096:
                    self._seen.popitem(last=False)
                return self._eager(batch.x, batch.positions, batch.batch)
            g = self._capture(batch)
            if g is None:
                return self._eager(batch.x, batch.positions, batch.batch)
            while len(self.graphs) >= self.max_graphs:
                self.graphs.popitem(last=False)
            self.graphs[fp] = g
        else:
            self.graphs.move_to_end(fp)
            sx, sp, dx, dp = batch.x, batch.positions, g["x"], g["pos"]
            if sx.is_contiguous() and sp.is_contiguous() and sx.dtype == dx.dtype and sp.dtype == dp.dtype:
                call("geossl_copy2", ptr(dx), ptr(sx), dx.numel() * dx.element_size(), ptr(dp), ptr(sp),
                     dp.numel() * dp.element_size(), stream())
            else:
                dx.copy_(sx)
                dp.copy_(sp)
        g["graph"].replay()
        st = self.model.__dict__.get("_geossl_status")
        if st is not None:  # deferred index check of the backbone (a replay never reaches model.forward's own)
            st.poll()
            st.arm(every=8)
        return g["out"]

    def _capture(self, batch):
        import gc
        import warnings
        # The graph binds the ADDRESSES of the index structures (mol_ptr, pair_ptr, pair atoms, aggregation work list, the
        # loop plan), which are cached on the batch vector (`_geossl_layout`).  The entry therefore owns a private clone of
        # that vector - with the host-side sizes re-attached, so the clone builds the same layout - and keeps vector and
        # layout alive for as long as the graph: the caller's batch may be freed (an eval loop over a loader) without the
        # caching allocator handing those addresses to someone else.
        x, pos = batch.x.clone(), batch.positions.clone()
        bvec = batch.batch.clone()
        hs = getattr(batch.batch, "_geossl_sizes", None)   # (host sizes, tensor version), layout.prepare_batch
        if hs is not None and hs[1] == batch.batch._version:
            bvec._geossl_sizes = (hs[0], bvec._version)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up off the capture: cached layouts, kernel attributes, the loop's block plan
            self._eager(x, pos, bvec)
            get_layout(bvec).loop_plan()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        gc_on = gc.isenabled()
        gc.disable()   # (no collection while a capture is open: see pretrain_GeoSSL.StepGraphs._capture)
        try:
            with torch.cuda.graph(graph, pool=self.pool, capture_error_mode="thread_local"):
                out = self._eager(x, pos, bvec)
            if self.pool is None:
                self.pool = graph.pool()
        except Exception as e:
            warnings.warn("HIP-graph capture of the forward pass failed (%s: %s); running eagerly" % (type(e).__name__, e))
            torch.cuda.synchronize()
            self.enabled = False
            return None
        finally:
            if gc_on:
                gc.enable()
        self.captures += 1
        return dict(graph=graph, x=x, pos=pos, out=out, batch_vec=bvec, layout=get_layout(bvec))


class ForceTrainer:
    """The training step of examples/finetune_md17.py:30-54 - energy head on the backbone's representation, predicted
    force = -dE/dpos taken with ``create_graph=True`` (:46), loss on energy and force (:46-51), ``loss.backward()``
    through the force (:53), ``optimizer.step()`` (:54) - on the library's second-order tape, with the whole of forward,
    position gradient, loss and second-order backward replayed as ONE captured HIP graph per batch structure.

    Launched eagerly such a step is ~1000 C-ABI calls of 10-15 us of host time each: host-bound by a factor of three or
    more.  The tape's launch sequence is a pure function of the batch's index structure, so it is
    captured once and replayed with new atom types, positions and targets copied into the graph's static inputs.

    Which batches share a graph: SchNet derives its graph from the positions inside the kernels (pair-slot form: no edge
    count reaches the host), so every batch with the same molecule sizes in the same order replays one graph - MD17
    (one molecule, a fixed batch size) never captures twice.  PaiNN takes the precomputed ``radius_edge_index`` as an
    input whose length the graph binds: a graph is replayed for batches whose edge tensor is the very same object
    (device-resident pre-collated batches that come back); anything else runs eagerly, like before.

    ``step(batch, actual_energy, actual_force) -> loss`` (a clone of the graph's static scalar)."""

    def __init__(self, model, head, model_3d="schnet", lr=5e-4, weight_decay=0.0, energy_coeff=0.05, force_coeff=0.95,
                 loss="l1", use_graph=True, max_graphs=16):
        from .optim import FlatParams, FusedAdam
        self.model, self.head, self.model_3d = model, head, model_3d
        self.coeff, self.loss_kind = (float(energy_coeff), float(force_coeff)), loss
        self.flat = FlatParams([model, head])
        self.opt = FusedAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.use_graph, self.max_graphs = use_graph, max_graphs
        self.graphs, self.pool, self.captures = OrderedDict(), None, 0
        self._seen = OrderedDict()   # (bounded: a loader that never repeats an edge tensor must not grow it for ever)

    # ---- the step as eager launches (what a capture records)
    def _body(self, x, positions, batch_vec, rei, y_e, y_f, ones):
        from . import _lib, ops
        pos = positions.detach().requires_grad_(True)                                                   # :33
        if self.model_3d == "painn":
            rep = self.model(x, pos, rei, batch_vec)
        else:
            rep = self.model(x[:, 0], pos, batch_vec)
        energy = self.head(rep).squeeze(1)                                                              # :36-44
        dE = torch.autograd.grad(energy, pos, grad_outputs=ones, create_graph=True, retain_graph=True)[0]   # :46
        loss = ops.energy_force_loss(energy, y_e, dE, y_f, self.coeff[0], self.coeff[1], self.loss_kind)    # :46-51
        self.flat.zero_grad()
        with _lib.direct_grads():
            loss.backward(inputs=self.flat.trainable)                                                   # :53
        return loss.detach()

    def _key(self, batch):
        from .pretrain_GeoSSL import _tensor_uid
        tag = lambda t_: None if t_ is None else (_tensor_uid(t_), t_._version, tuple(t_.shape))
        sizes = getattr(batch, "_sizes", None)
        if self.model_3d == "painn":
            return ("tensors", tag(batch.batch), tag(getattr(batch, "radius_edge_index", None)))
        if sizes is not None:
            import numpy as np
            return ("sizes", int(batch.batch.numel()), np.asarray(sizes, dtype=np.int32).tobytes())
        return ("tensors", tag(batch.batch))

    def _capture(self, batch, y_e, y_f):
        import gc
        import warnings
        B = int(y_e.numel())
        dev = batch.positions.device
        # the graph binds the addresses of the index structures cached on the batch vector: the entry owns them (a private
        # clone for graphs shared by size sequence, the caller's own tensors - kept alive here - for graphs that serve
        # exactly those tensors)
        bvec = batch.batch
        if self.model_3d != "painn":
            bvec = batch.batch.clone()
            hs = getattr(batch.batch, "_geossl_sizes", None)
            if hs is not None and hs[1] == batch.batch._version:
                bvec._geossl_sizes = (hs[0], bvec._version)
        st = dict(x=batch.x.clone(), pos=batch.positions.detach().clone(), y_e=y_e.detach().clone(),
                  y_f=y_f.detach().clone(), ones=torch.ones(B, dtype=torch.float32, device=dev),
                  bvec=bvec, rei=getattr(batch, "radius_edge_index", None) if self.model_3d == "painn" else None)
        run = lambda: self._body(st["x"], st["pos"], st["bvec"], st["rei"], st["y_e"], st["y_f"], st["ones"])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up off the capture: layouts, incidence lists, kernel attributes
            run()
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        gc_on = gc.isenabled()
        gc.disable()   # (no collection while a capture is open: see pretrain_GeoSSL.StepGraphs._capture)
        try:
            with torch.cuda.graph(graph, pool=self.pool, capture_error_mode="thread_local"):
                st["loss"] = run()
            if self.pool is None:
                self.pool = graph.pool()
        except Exception as e:
            warnings.warn("HIP-graph capture of the force-training step failed (%s: %s); running eagerly"
                          % (type(e).__name__, e))
            torch.cuda.synchronize()
            self.use_graph = False
            return None
        finally:
            if gc_on:
                gc.enable()
        self.flat.rebind_grads()
        self.captures += 1
        st["graph"] = graph
        return st

    def step(self, batch, actual_energy, actual_force):
        from . import _lib
        _lib.require_cuda(batch.positions, actual_energy, actual_force)
        if actual_energy.numel() != batch.num_graphs or tuple(actual_force.shape) != tuple(batch.positions.shape):
            raise ValueError("targets: one energy per molecule, one force row per atom")
        g = None
        if self.use_graph and not torch.cuda.is_current_stream_capturing():
            key = self._key(batch)
            g = self.graphs.get(key)
            if g is None and (key[0] == "sizes" or key in self._seen):   # (a tensor-identified structure: second sighting)
                g = self._capture(batch, actual_energy, actual_force)
                if g is not None:
                    while len(self.graphs) >= self.max_graphs:
                        self.graphs.popitem(last=False)
                    self.graphs[key] = g
            elif g is None:
                self._seen[key] = True
                while len(self._seen) > 4096:
                    self._seen.popitem(last=False)
            else:
                self.graphs.move_to_end(key)
        if g is None:
            ones = torch.ones(batch.num_graphs, dtype=torch.float32, device=batch.positions.device)
            loss = self._body(batch.x, batch.positions, batch.batch, getattr(batch, "radius_edge_index", None),
                              actual_energy, actual_force, ones)
            self.flat.rebind_grads()
        else:
            cb = _lib.CopyBatch()
            jobs = [(g["x"], batch.x), (g["pos"], batch.positions), (g["y_e"], actual_energy.reshape(g["y_e"].shape)),
                    (g["y_f"], actual_force)]
            if all(s.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape for d, s in jobs):
                for k, (d, s) in enumerate(jobs):
                    cb.dst[k], cb.src[k], cb.bytes[k] = ptr(d), ptr(s), d.numel() * d.element_size()
                call("geossl_copy_n", C.byref(cb), len(jobs), stream())
            else:
                for d, s in jobs:
                    d.copy_(s)
            g["graph"].replay()
            loss = g["loss"].clone()
        self.opt.step()                                                                                 # :54
        return loss
