"""GPU tests added in round 6: the advisor's findings of round 5 (a replayed forward graph must own its index
structures; the energy / force loss checks its targets), the device-resident dataset and its molecule gather, and the
kernels this round touched."""
import gc
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import product_ncsn, product_schnet, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
            readout="mean")
SMALL = dict(hidden_channels=128, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=5.0, node_class=9,
             readout="mean")


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from geossl_amd import _lib
    _lib.load()


def _ragged_sizes(B, seed, lo=2, hi=33, mean=18.0, sd=4.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(mean, sd, size=B)), lo, hi).astype(np.int64)


# ------------------------------------------------------------------------------------ advisor, round 5
def test_graphed_forward_outlives_the_batch_it_was_captured_on():
    """the graph of GraphedForward binds the addresses of the first batch's index structures; an eval
    loop over a loader frees that batch.  The entry now owns a private batch vector + layout: capture, drop the batch,
    let the allocator hand its blocks to junk tensors, replay on a fresh same-sized batch -> the eager forward, bit for
    bit."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.graphed import GraphedForward
    from geossl_amd.synthetic import make_batch
    model = product_schnet(SMALL, DEV)
    gf = GraphedForward(model)
    sizes = _ragged_sizes(48, 3)
    for uniform in (True, False):
        mk = (lambda s: make_batch(48, seed=s)) if uniform else (lambda s: make_batch(0, seed=s, sizes=sizes))
        first = pg.Batch.from_numpy(mk(100), DEV)
        gf(first)
        gf(first)                                  # ragged: captured at the second sighting
        n_cap = gf.captures
        entry = next(reversed(gf.graphs.values()))
        assert entry["batch_vec"].data_ptr() != first.batch.data_ptr() and entry["layout"] is not None
        del first
        gc.collect()
        torch.cuda.synchronize()
        junk = [torch.full((n,), -7, dtype=torch.int32, device=DEV) for n in (49, 97, 864, 7344, 7344, 20000, 100000)]
        junk += [torch.full((n,), 2 ** 40, dtype=torch.int64, device=DEV) for n in (864, 49, 7344)]
        for s in (101, 102):
            bt = pg.Batch.from_numpy(mk(s), DEV)
            with torch.no_grad():
                want = model(bt.x[:, 0], bt.positions, bt.batch)
            got = gf(bt)
            assert gf.captures == n_cap and torch.equal(got, want)
        del junk


def test_energy_force_loss_rejects_mismatched_targets():
    """the loss kernels walk flat buffers of pred_energy.numel() / dE_dpos.numel() elements; a target
    of another size raises instead of being read past its end."""
    from geossl_amd import ops
    e, f = torch.randn(8, device=DEV), torch.randn(40, 3, device=DEV)
    ops.energy_force_loss(e, torch.randn(8, device=DEV), f, torch.randn(40, 3, device=DEV))
    with pytest.raises(ValueError):
        ops.energy_force_loss(e, torch.randn(4, device=DEV), f, torch.randn(40, 3, device=DEV))
    with pytest.raises(ValueError):
        ops.energy_force_loss(e, torch.randn(8, device=DEV), f, torch.randn(39, 3, device=DEV))
    with pytest.raises(ValueError):
        ops.energy_force_loss(e, torch.randn(8, device=DEV), f, torch.randn(3, 40, device=DEV))
