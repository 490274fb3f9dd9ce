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
from collections import OrderedDict

import torch

from ._lib import call, ptr, stream
from .layout import get_layout


class GraphedForward:
    def __init__(self, model, return_latent=False, max_graphs=64):
        self.model, self.return_latent, self.max_graphs = model, return_latent, max_graphs
        self.graphs, self.pool, self.captures = OrderedDict(), None, 0
        self._seen = set()
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
                self._seen.add(fp)
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
