"""GPU tests added in round 4: capacity-bucket step graphs (ONE graph replayed on batches of any size sequence), the
`_dyn` entry points behind them through the C ABI, the headline shape against the oracle through the replayed trainer
path, the deferred atom-type check on do_DDM's graph path."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import ncsn_oracle_params, product_ncsn, product_schnet, schnet_oracle_params, t, unique_named_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = dict(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=5.0, node_class=9,
            readout="mean")
SMALL = dict(hidden_channels=128, num_filters=128, num_interactions=2, num_gaussians=51, cutoff=5.0, node_class=9,
             readout="mean")
TOL_OUT, TOL_GRAD = 1e-5, 1e-4


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from geossl_amd import _lib
    _lib.load()


def _ragged_sizes(B, seed, lo=2, hi=33):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(18.0, 4.0, size=B)), lo, hi).astype(np.int64)


def _trainer(cfg=SMALL, **kw):
    from geossl_amd import pretrain_GeoSSL as pg
    return pg.DDMTrainer(product_schnet(cfg, DEV), product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9),
                         lr=5e-4, **kw)


def _bucket_of(tr):
    keys = [k for k in tr._graphs if isinstance(k, tuple) and k and k[0] == "bucket"]
    assert len(keys) == 1, list(tr._graphs)
    return tr._graphs[keys[0]]["bucket"]


# ------------------------------------------------------------------------------------------------ bucket graphs
def test_one_graph_replays_on_every_size_sequence_bit_for_bit():
    """The reference's loader hands over ragged molecules in shuffled order (pretrain_GeoSSL.py:301): no two batches share
    a size sequence.  DDMTrainer(use_graph=True) serves them all from ONE captured graph (a capacity bucket: the kernels
    read the real counts and every index structure from device memory).  Five batches with five size sequences - different
    atom, pair-slot and super-edge counts, single-atom molecules among them: one capture, losses and parameters
    bit-identical to the SAME launches made eagerly on the same bucket, and equal to the plain eager step on the batch
    itself within fp32 summation order."""
    from geossl_amd import bucket as bk
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    B = 24
    specs = [_ragged_sizes(B, 10 + i) for i in range(5)]
    specs[2][3] = 1                       # a molecule without pairs in the middle of a batch
    specs[3][-1] = 1                      # ... and at the end (NCSN.py:210-212: the mean's divisor drops to B - 1)
    # (the batch with the most pair slots first: at 24 molecules per batch the counts spread by a third, more than the
    # slack of a first capture - growth is the next test's subject)
    specs.sort(key=lambda s: -int((s * (s - 1) // 2).sum()))
    raw = [make_batch(B, seed=200 + i, sizes=s) for i, s in enumerate(specs)]
    assert len({tuple(s) for s in map(tuple, specs)}) == 5
    assert len({b["x"].shape[0] for b in raw}) > 1 and len({b["super_edge_index"].shape[1] for b in raw}) > 1
    noise = [{k: t(v, DEV) for k, v in draw_noise(b, seed=300 + i).items()} for i, b in enumerate(raw)]

    tr = _trainer(use_graph=True)
    got = [float(tr.step(pg.Batch.from_numpy(b, DEV), nz)) for b, nz in zip(raw, noise)]
    assert tr.use_graph and tr.step_graphs.captures == 1 and len(tr._graphs) == 1
    bkt = _bucket_of(tr)
    params_graph = tr.flat.flat.clone()

    # the same launches, eagerly: a bucket of the same capacity, filled batch by batch
    te = _trainer(use_graph=False)
    eb = bk.Bucket(torch.device(DEV), B, bkt.caps(), "combination")
    f32 = dict(dtype=torch.float32, device=DEV)
    sn = {"pos_noise": torch.zeros(eb.N_cap, 3, **f32), "dist_noise_1": torch.zeros(eb.S_cap, 1, **f32),
          "dist_noise_2": torch.zeros(eb.S_cap, 1, **f32), "noise_level_1": torch.zeros(B, dtype=torch.long, device=DEV),
          "noise_level_2": torch.zeros(B, dtype=torch.long, device=DEV)}
    eager_bucket = []
    for b, nz in zip(raw, noise):
        N, P, S, W = eb.fill(pg.Batch.from_numpy(b, DEV))
        sn["pos_noise"][:N].copy_(nz["pos_noise"])
        sn["dist_noise_1"][:S].copy_(nz["dist_noise_1"])
        sn["dist_noise_2"][:S].copy_(nz["dist_noise_2"])
        sn["noise_level_1"].copy_(nz["noise_level_1"])
        sn["noise_level_2"].copy_(nz["noise_level_2"])
        loss = te._fwd_bwd(eb.batch, sn)
        te.opt.step(grad_scale=te.reduce())
        eager_bucket.append(float(loss))
    assert got == eager_bucket, (got, eager_bucket)
    assert torch.equal(params_graph, te.flat.flat)

    # the plain eager step on the batch itself (exact shapes, per-batch index structures)
    tp = _trainer(use_graph=False)
    plain = [float(tp.step(pg.Batch.from_numpy(b, DEV), nz)) for b, nz in zip(raw, noise)]
    for a, c in zip(got, plain):
        assert abs(a - c) <= 2e-6 * abs(c), (got, plain)
    assert rel_err(params_graph, tp.flat.flat) < 1e-5


def test_bucket_grows_when_a_batch_outgrows_it_and_keeps_one_graph():
    """A batch larger than the bucket's capacity (small molecules first, then large ones) makes a larger bucket and a new
    capture; the smaller batches that follow replay the larger graph.  Losses as the plain eager trainer computes them."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    B = 16
    specs = [np.full(B, 6), _ragged_sizes(B, 3), np.full(B, 30), _ragged_sizes(B, 4), np.full(B, 6)]
    specs[0][0], specs[2][0], specs[4][1] = 7, 29, 5      # (not uniform: uniform batches keep their per-structure graph)
    raw = [make_batch(B, seed=400 + i, sizes=s) for i, s in enumerate(specs)]
    noise = [{k: t(v, DEV) for k, v in draw_noise(b, seed=500 + i).items()} for i, b in enumerate(raw)]
    tr, tp = _trainer(use_graph=True), _trainer(use_graph=False)
    caps = []
    for b, nz in zip(raw, noise):
        a = float(tr.step(pg.Batch.from_numpy(b, DEV), nz))
        c = float(tp.step(pg.Batch.from_numpy(b, DEV), nz))
        assert abs(a - c) <= 2e-6 * abs(c), (a, c)
        caps.append(_bucket_of(tr).caps())
    assert len(tr._graphs) == 1 and tr.step_graphs.captures == 3          # 6-atom, ragged, 30-atom molecules
    assert caps[0] != caps[1] != caps[2] and caps[2] == caps[3] == caps[4]
    assert rel_err(tr.flat.flat, tp.flat.flat) < 1e-5


def test_trainer_draws_its_own_noise_into_a_bucket():
    """noise=None: the trainer's one-launch draws (geossl_ddm_noise) fill the bucket's capacity-sized noise buffers - the
    value of an element depends on the seed and its index only, so the step sees the draws of the exact-sized launch."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    B = 12
    raw = [make_batch(B, seed=600 + i, sizes=_ragged_sizes(B, 60 + i)) for i in range(3)]
    out = {}
    for use_graph in (False, True):
        torch.cuda.manual_seed(99)
        tr = _trainer(use_graph=use_graph)
        out[use_graph] = [float(tr.step(pg.Batch.from_numpy(b, DEV), None)) for b in raw]
        if use_graph:
            assert tr.step_graphs.captures == 1
    for a, c in zip(out[True], out[False]):
        assert np.isfinite(a) and abs(a - c) <= 2e-6 * abs(c), out


def test_reference_loop_on_a_shuffled_ragged_loader_captures_once():
    """examples/pretrain_GeoSSL.py:248-260 on the product modules over the reference's own loader surface
    (AtomTupleExtractor transform, DataLoaderAtomTuple(shuffle=True), batch.to(device)) with ragged molecules: every batch
    has its own size sequence, do_DDM captures ONE pair of graphs (forward, backward) at the first step and replays it
    from then on; the draws are the eager loop's (same generators, same order, same sizes), so losses and parameters
    follow the eager loop within fp32 summation order."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, Data, DataLoaderAtomTuple
    from geossl_amd.synthetic import make_batch
    sizes = _ragged_sizes(96, 7)
    b = make_batch(96, seed=77, sizes=sizes)
    ext = AtomTupleExtractor(ratio=1, option="combination")
    off, dataset = 0, []
    for n in sizes.tolist():
        dataset.append(ext(Data(x=torch.from_numpy(b["x"][off:off + n]), positions=torch.from_numpy(b["positions"][off:off + n]))))
        off += n
    out = {}
    for graph in (False, True):
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        model = product_schnet(SMALL, DEV)
        n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
        pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
        args = types.SimpleNamespace(model_3d="schnet", GeoSSL_mu=0.0, GeoSSL_sigma=0.3, lr=5e-4, decay=0.0, step_graph=graph)
        group = [{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()}, {"params": n2.parameters()}]
        optimizer = torch.optim.Adam(group, lr=args.lr, weight_decay=args.decay)
        gen = torch.Generator()
        gen.manual_seed(1)
        loader = DataLoaderAtomTuple(dataset, batch_size=16, shuffle=True, generator=gen)
        losses, seqs = [], set()
        try:
            for epoch in range(2):
                for batch in loader:
                    batch = batch.to(DEV)
                    seqs.add(tuple(batch._sizes))
                    loss, acc = pg.do_DDM(args, batch, model, criterion=None, mu=args.GeoSSL_mu, sigma=args.GeoSSL_sigma)
                    losses.append(loss.detach().item())
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
        finally:
            pg.NCSN_model_01 = pg.NCSN_model_02 = None
        assert len(seqs) == len(losses) == 12
        eng = model.__dict__.get("_geossl_autograd_step")
        caps = sum(sg.captures for sg in eng.graphs.values()) if eng is not None else 0
        out[graph] = (losses, torch.cat([p.detach().reshape(-1) for m in (model, n1, n2) for p in m.parameters()]).cpu(), caps)
    assert out[False][2] == 0 and 1 <= out[True][2] <= 2, out[True][2]       # (a second capture only if the bucket grew)
    for a, c in zip(out[True][0], out[False][0]):
        assert abs(a - c) <= 5e-6 * abs(c), (out[True][0], out[False][0])
    assert rel_err(out[True][1], out[False][1]) < 1e-4


def test_out_of_range_atom_type_is_reported_on_do_ddms_graph_path():
    """the reference raises IndexError from Embedding for an atom type outside the table; a replayed
    step never reaches model.forward's own check.  do_DDM polls the status word behind its replays like DDMTrainer.step:
    the error surfaces a few steps late instead of never."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
    args = pg.Args("schnet")
    b = make_batch(8, seed=3)
    good = pg.Batch.from_numpy(b, DEV)
    for _ in range(3):
        loss, _ = pg.do_DDM(args, good, model, NCSN_models=heads, graph=True)
        loss.backward()
    eng = model.__dict__["_geossl_autograd_step"]
    assert sum(sg.captures for sg in eng.graphs.values()) == 1
    b["x"][5, 0] = 17
    bad = pg.Batch.from_numpy(b, DEV)
    with pytest.raises(IndexError):
        for _ in range(40):
            loss, _ = pg.do_DDM(args, bad, model, NCSN_models=heads, graph=True)
            loss.backward()
            torch.cuda.synchronize()


def test_engine_on_the_module_survives_deepcopy_and_pickle():
    """the graph engine and the status word hang on the backbone's __dict__; copy.deepcopy (an EMA twin)
    and pickling the module must not trip over their streams / events / graphs."""
    import copy
    import io
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
    batch = pg.Batch.from_numpy(make_batch(8, seed=3), DEV)
    loss, _ = pg.do_DDM(pg.Args("schnet"), batch, model, NCSN_models=heads, graph=True)
    loss.backward()
    assert model.__dict__.get("_geossl_autograd_step") is not None
    twin = copy.deepcopy(model)
    assert twin.__dict__.get("_geossl_autograd_step") is None
    buf = io.BytesIO()
    torch.save(model, buf)
    l2, _ = pg.do_DDM(pg.Args("schnet"), batch, twin, NCSN_models=heads, graph=True)
    assert torch.isfinite(l2)


def test_paired_heads_refuse_shared_parameters():
    """the same head passed twice must not take the paired launches (two blocks of one launch would add
    into the same gradient addresses): it takes the two single-head calls and gives their sum."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(6, seed=9)
    batch = pg.Batch.from_numpy(b, DEV)
    nz = {k: t(v, DEV) for k, v in draw_noise(b, seed=10).items()}
    model, n1 = product_schnet(SMALL, DEV), product_ncsn(128, 50, 2, DEV)
    n1b = product_ncsn(128, 50, 2, DEV)
    la, _ = pg.do_DDM(pg.Args("schnet"), batch, model, NCSN_models=(n1, n1), noise=nz, graph=False)
    la.backward()
    ga = n1.output_mlp.layers[0].weight.grad.clone()
    model.zero_grad()
    lb, _ = pg.do_DDM(pg.Args("schnet"), batch, model, NCSN_models=(n1b, product_ncsn(128, 50, 2, DEV)), noise=nz, graph=False)
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb))
    assert torch.isfinite(ga).all() and float(ga.abs().max()) > 0


# ------------------------------------------------------------------------------------- the headline shape, pinned
def test_headline_shape_through_the_replayed_trainer_vs_oracle():
    """1024 molecules of set A with injected noise through DDMTrainer(use_graph=True) - the 512-block
    layer loop, the paired heads and the graph replay are all on - against oracle.nets.do_ddm_schnet: loss <= 1e-5,
    backbone gradients <= 1e-4; and the replay is the eager launch bit for bit at this size."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    b = make_batch(1024, seed=0, mode="A")
    nz = draw_noise(b, seed=1)
    batch = pg.Batch.from_numpy(b, DEV)
    noise = {k: t(v, DEV) for k, v in nz.items()}
    res = {}
    for use_graph in (True, False):
        tr = _trainer(FULL, use_graph=use_graph)
        loss = tr._graph_fwd_bwd(batch, noise) if use_graph else tr._fwd_bwd(batch, noise)   # (no Adam: gradients stay)
        if use_graph:
            assert tr.use_graph and tr.step_graphs.captures == 1
        res[use_graph] = (loss.clone(), tr.flat.grad.clone(), tr)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Pm, P1, P2 = schnet_oracle_params(FULL), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 5.0, 6, 2, "mean")
    ref.backward()
    assert rel_err(res[True][0].cpu(), ref.detach()) < TOL_OUT
    g = unique_named_grads(res[True][2].model)
    for k in ("lin2.weight", "lin1.weight", "interactions.0.mlp.0.weight", "interactions.3.mlp.2.weight",
              "interactions.5.conv.lin1.weight", "interactions.2.conv.lin2.weight", "interactions.4.lin.weight",
              "embedding.weight"):
        assert rel_err(g[k].cpu(), Pm[k].grad) < TOL_GRAD, k


def test_ragged_full_size_through_the_bucket_graph_vs_oracle():
    """Set B (ragged, 2..33 atoms) at the bench size through the bucket graph of DDMTrainer against the oracle:
    loss <= 1e-5, backbone gradients <= 1e-4 (512 molecules: the oracle's step stays within seconds)."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    b = make_batch(512, seed=5, mode="B")
    nz = draw_noise(b, seed=6)
    batch = pg.Batch.from_numpy(b, DEV)
    noise = {k: t(v, DEV) for k, v in nz.items()}
    tr = _trainer(FULL, use_graph=True)
    loss = tr._graph_fwd_bwd(batch, noise)
    assert tr.use_graph and tr.step_graphs.captures == 1 and _bucket_of(tr) is not None
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Pm, P1, P2 = schnet_oracle_params(FULL), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 5.0, 6, 2, "mean")
    ref.backward()
    assert rel_err(loss.cpu(), ref.detach()) < TOL_OUT
    g = unique_named_grads(tr.model)
    for k in ("lin2.weight", "interactions.0.mlp.0.weight", "interactions.5.conv.lin1.weight",
              "interactions.2.conv.lin2.weight", "embedding.weight"):
        assert rel_err(g[k].cpu(), Pm[k].grad) < TOL_GRAD, k


# ------------------------------------------------------------------------------- `_dyn` entry points, C ABI level
def _dims(*vals):
    return torch.tensor(list(vals), dtype=torch.int32, device=DEV)


def test_filter_forward_with_a_device_side_row_count():
    """geossl_cfconv_filter_fwd_dyn: grid and layer stride from the capacity, rows from device memory - the real rows are
    the plain launch's bit for bit, rows past the count are never written."""
    from geossl_amd import _lib
    L, F, G, P, Pc = 3, 128, 51, 1000, 1408
    gen = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=gen) * sc).contiguous()
    d, c = torch.rand(Pc, device=DEV, generator=gen) * 5.0, torch.rand(Pc, device=DEV, generator=gen)
    w = [(rnd(F, G, sc=0.2), rnd(F, sc=0.1), rnd(F, F, sc=0.1), rnd(F, sc=0.1)) for _ in range(L)]
    fw = _lib.FilterWeights()
    for l, (a, b1, a2, b2) in enumerate(w):
        fw.w1[l], fw.b1[l], fw.w2[l], fw.b2[l] = a.data_ptr(), b1.data_ptr(), a2.data_ptr(), b2.data_ptr()
    offset = torch.linspace(0.0, 5.0, G, device=DEV)
    coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    T0, W0 = torch.empty(L, P, F, device=DEV), torch.empty(L, P, F, device=DEV)
    _lib.call("geossl_cfconv_filter_fwd", d.data_ptr(), c.data_ptr(), P, C.byref(fw), L, F, G, offset.data_ptr(), coeff,
              T0.data_ptr(), W0.data_ptr(), _lib.stream())
    T1, W1 = torch.full((L, Pc, F), 7.0, device=DEV), torch.full((L, Pc, F), 7.0, device=DEV)
    dims = _dims(P)
    _lib.call("geossl_cfconv_filter_fwd_dyn", d.data_ptr(), c.data_ptr(), Pc, C.byref(fw), L, F, G, offset.data_ptr(), coeff,
              T1.data_ptr(), W1.data_ptr(), dims.data_ptr(), _lib.stream())
    assert torch.equal(T1[:, :P], T0) and torch.equal(W1[:, :P], W0)
    assert bool((T1[:, P:] == 7.0).all()) and bool((W1[:, P:] == 7.0).all())


def test_row_kernels_with_a_device_side_row_count():
    """geossl_linear_chain_dyn / geossl_linear_wgrad_dyn / geossl_embedding_{fwd,bwd}_dyn: the rows below the device-side
    count are computed as by the plain launch over exactly those rows (chain, embedding: bit for bit; the reductions over
    rows: the same sums cut at other block boundaries), the rows past it are neither read (NaN there) nor written."""
    from geossl_amd import _lib, ops
    F, R, Rc = 128, 700, 1024
    gen = torch.Generator(device=DEV).manual_seed(2)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=gen) * sc).contiguous()
    X, W, bias = rnd(Rc, F), rnd(F, F, sc=0.1), rnd(F, sc=0.1)
    X[R:] = float("nan")
    dims = _dims(R)
    img = ops.prepare_chain([W], transB=True)
    y0 = ops.linear_chain(X[:R], [dict(image=img[0], bias=bias, flags=_lib.EPI_SSP)])[0]
    out = torch.full((Rc, F), 7.0, device=DEV)
    ops.linear_chain(X, [dict(image=img[0], bias=bias, flags=_lib.EPI_SSP, out=out)], dyn_rows=dims.data_ptr())
    assert torch.equal(out[:R], y0) and bool((out[R:] == 7.0).all())
    # weight gradient: dW = A^T B over the real rows
    A, Bm = rnd(Rc, F), rnd(Rc, F)
    A[R:], Bm[R:] = float("nan"), float("nan")
    dW0, db0, dW1, db1 = (torch.empty(F, F, device=DEV), torch.empty(F, device=DEV), torch.empty(F, F, device=DEV),
                          torch.empty(F, device=DEV))
    ops.linear_wgrad([(A[:R], Bm[:R], dW0, db0)], R, F, F)
    ops.linear_wgrad([(A, Bm, dW1, db1)], Rc, F, F, dyn_rows=dims.data_ptr())
    ref = A[:R].double().t() @ Bm[:R].double()
    assert rel_err(dW1, ref) < 2e-6 and rel_err(dW1, dW0) < 2e-6 and rel_err(db1, A[:R].double().sum(0)) < 2e-6
    # embedding forward / backward
    z = torch.randint(0, 9, (Rc, 2), device=DEV, generator=gen)
    z[R:, 0] = 99                                       # outside the table: must not be looked at
    table, status = rnd(9, F), torch.zeros(1, dtype=torch.int32, device=DEV)
    h1 = torch.full((Rc, F), 7.0, device=DEV)
    _lib.call("geossl_embedding_fwd_dyn", z.data_ptr(), 2, table.data_ptr(), 9, Rc, F, h1.data_ptr(), status.data_ptr(),
              dims.data_ptr(), _lib.stream())
    assert torch.equal(h1[:R], table[z[:R, 0]]) and bool((h1[R:] == 7.0).all()) and int(status) == 0
    dh = rnd(Rc, F)
    dh[R:] = float("nan")
    nfl = _lib.load().geossl_embedding_bwd_workspace_floats(9, F)
    ws, g1 = torch.empty(nfl, device=DEV), torch.empty(9, F, device=DEV)
    _lib.call("geossl_embedding_bwd_dyn", z.data_ptr(), 2, dh.data_ptr(), 9, Rc, F, g1.data_ptr(), ws.data_ptr(), 0,
              dims.data_ptr(), _lib.stream())
    gref = torch.zeros(9, F, dtype=torch.float64, device=DEV).index_add_(0, z[:R, 0], dh[:R].double())
    assert rel_err(g1, gref) < 2e-6


def test_copy_n_and_ddm_views_with_device_side_counts():
    from geossl_amd import _lib, ops
    gen = torch.Generator(device=DEV).manual_seed(3)
    srcs = [torch.randint(0, 1000, (n,), device=DEV, generator=gen, dtype=torch.int32) for n in (5, 1000, 33, 0, 70000)]
    dsts = [torch.full((n + 3,), -1, device=DEV, dtype=torch.int32) for n in (5, 1000, 33, 0, 70000)]
    cb = _lib.CopyBatch()
    for k, (d_, s_) in enumerate(zip(dsts, srcs)):
        cb.dst[k], cb.src[k], cb.bytes[k] = d_.data_ptr(), s_.data_ptr(), 4 * s_.numel()
    _lib.call("geossl_copy_n", C.byref(cb), 5, _lib.stream())
    for d_, s_ in zip(dsts, srcs):
        assert torch.equal(d_[:s_.numel()], s_) and bool((d_[s_.numel():] == -1).all())
    # both views behind one another at the REAL atom count
    N, Nc, S, Sc = 50, 64, 120, 160
    pos, noise = torch.randn(Nc, 3, device=DEV, generator=gen), torch.randn(Nc, 3, device=DEV, generator=gen) * 0.3
    x = torch.randint(0, 9, (Nc, 2), device=DEV, generator=gen)
    sei = torch.randint(0, N, (2, Sc), device=DEV, generator=gen)
    p0, a0, b0, z0 = ops.ddm_views(pos[:N].contiguous(), noise[:N].contiguous(), sei[0, :S].contiguous(),
                                   sei[1, :S].contiguous(), z=x[:N, 0])

    class D:
        pass
    dims = _dims(N, S)
    dyn = D()
    dyn.n_atoms, dyn.n_super = dims.data_ptr(), dims.data_ptr() + 4
    p1, a1, b1, z1 = ops.ddm_views(pos, noise, sei[0].contiguous(), sei[1].contiguous(), z=x[:, 0], dyn=dyn)
    assert torch.equal(p1[:2 * N], p0) and torch.equal(z1[:2 * N], z0)
    assert torch.equal(a1[:S], a0) and torch.equal(b1[:S], b0)


# --------------------------------------------------------------------------- stock torch.optim.Adam on the flat path
def _ref_loop(graph, steps, batches, adam_kw=None, between=None, cfg=SMALL, seed=11):
    """examples/pretrain_GeoSSL.py:248-260 + :332-343 on the product modules -> (losses, parameters, optimizer)."""
    from geossl_amd import pretrain_GeoSSL as pg
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
    pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
    args = types.SimpleNamespace(model_3d="schnet", GeoSSL_mu=0.0, GeoSSL_sigma=0.3, lr=5e-4, decay=0.0, step_graph=graph)
    group = [{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()}, {"params": n2.parameters()}]
    optimizer = torch.optim.Adam(group, lr=args.lr, weight_decay=args.decay, **(adam_kw or {}))
    losses = []
    try:
        for step in range(steps):
            loss, acc = pg.do_DDM(args, batches[step % len(batches)], model, criterion=None, mu=0.0, sigma=0.3)
            losses.append(loss.detach().item())
            optimizer.zero_grad()
            loss.backward()
            if between is not None:
                between(step, model, n1, n2, optimizer)
            optimizer.step()
    finally:
        pg.NCSN_model_01 = pg.NCSN_model_02 = None
    params = torch.cat([p.detach().reshape(-1) for m in (model, n1, n2) for p in m.parameters()]).cpu()
    return losses, params, optimizer, (model, n1, n2)


def test_stock_adam_runs_as_one_launch_on_the_graph_path_bit_for_bit():
    """The reference's optimizer, unmodified (torch.optim.Adam over three groups, :332-343): on do_DDM's graph path its
    step is ONE geossl_adam_step launch (the parameters and the gradients autograd received are views of flat buffers; a
    global step hook does the update and leaves the stock step nothing to do) - with torch's foreach arithmetic bit for
    bit: losses and parameters equal the eager loop's, the optimizer's state has torch's layout and values, the gradients
    are back on the parameters after the step."""
    from geossl_amd import _lib
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    batches = [pg.Batch.from_numpy(make_batch(32, seed=40 + i), DEV) for i in range(4)]
    calls = []
    real = _lib.call

    def counting(name, *a):
        calls.append(name)
        return real(name, *a)

    eager = _ref_loop(False, 6, batches)
    pg.call, saved = counting, pg.call
    import geossl_amd.optim as go
    go.call = counting
    try:
        graph = _ref_loop(True, 6, batches)
    finally:
        pg.call, go.call = saved, real
    assert calls.count("geossl_adam_step") == 6
    assert graph[0] == eager[0] and torch.equal(graph[1], eager[1])
    so, se = graph[2].state_dict(), eager[2].state_dict()
    assert so["param_groups"] == se["param_groups"] and set(so["state"]) == set(se["state"])
    for k in se["state"]:
        assert float(so["state"][k]["step"]) == float(se["state"][k]["step"]) == 6.0
        assert torch.equal(so["state"][k]["exp_avg"], se["state"][k]["exp_avg"])
        assert torch.equal(so["state"][k]["exp_avg_sq"], se["state"][k]["exp_avg_sq"])
    model = graph[3][0]
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)
    # the optimizer's state survives a round trip, and the hook takes the loaded copies in again
    graph[2].load_state_dict(graph[2].state_dict())


def test_stock_adam_hook_steps_aside_for_what_it_does_not_know():
    """amsgrad (a different update), a gradient replaced by a copy between backward() and step() (not a view of the flat
    buffer any more) and in-place clipping (still the views: fused) - results as the eager loop computes them."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    batches = [pg.Batch.from_numpy(make_batch(16, seed=60 + i), DEV) for i in range(2)]
    a, b = _ref_loop(False, 4, batches, adam_kw=dict(amsgrad=True)), _ref_loop(True, 4, batches, adam_kw=dict(amsgrad=True))
    assert a[0] == b[0] and torch.equal(a[1], b[1])

    def replace_one(step, model, n1, n2, opt):
        if step == 2:
            model.lin2.weight.grad = model.lin2.weight.grad.clone() * 0.5

    a, b = _ref_loop(False, 4, batches, between=replace_one), _ref_loop(True, 4, batches, between=replace_one)
    assert a[0] == b[0] and torch.equal(a[1], b[1])

    def clip(step, model, n1, n2, opt):
        torch.nn.utils.clip_grad_norm_(list(model.parameters()) + list(n1.parameters()) + list(n2.parameters()), 0.5)

    a, b = _ref_loop(False, 4, batches, between=clip), _ref_loop(True, 4, batches, between=clip)
    for x, y in zip(a[0], b[0]):
        assert abs(x - y) <= 1e-6 * abs(y)
    assert rel_err(b[1], a[1]) < 1e-6


def test_filter_backward_rebuilds_the_hidden_rows_when_they_were_not_saved(monkeypatch):
    """GEOSSL_FILTER_RECOMPUTE_T: the forward stores Wf only and geossl_cfconv_filter_bwd (T = NULL) rebuilds
    t = ssp(W1 rbf(d) + b1) per tile - the same gradients as from the saved rows (fp32 rounding apart), on a ragged batch
    whose tiles straddle molecules and end in a partial tile."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(40, seed=21, sizes=_ragged_sizes(40, 21))
    batch = pg.Batch.from_numpy(b, DEV)
    noise = {k: t(v, DEV) for k, v in draw_noise(b, seed=22).items()}
    grads = {}
    for mode in ("saved", "rebuilt"):
        if mode == "rebuilt":
            monkeypatch.setenv("GEOSSL_FILTER_RECOMPUTE_T", "1")
        model = product_schnet(FULL, DEV)
        heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
        loss, _ = pg.do_DDM(pg.Args("schnet"), batch, model, NCSN_models=heads, noise=noise, graph=False)
        loss.backward()
        grads[mode] = (float(loss), {k: v.clone() for k, v in unique_named_grads(model).items()})
    assert grads["saved"][0] == grads["rebuilt"][0]                      # (the forward does not change)
    for k, v in grads["saved"][1].items():
        assert rel_err(grads["rebuilt"][1][k], v) < 2e-6, k


def test_bucket_graph_with_the_permutation_enumeration_and_the_switch_that_turns_buckets_off(monkeypatch):
    """AtomTupleExtractor(option="permutation") (dataloaders_AtomTuple.py:19-20: both orders of every pair, twice the
    super-edges and incidence entries) through a bucket graph; and GEOSSL_NO_BUCKETS, under which ragged batches go back to
    per-structure graphs from their second sighting on."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    B = 10
    raw = [make_batch(B, seed=700 + i, sizes=_ragged_sizes(B, 70 + i, hi=20), option="permutation") for i in range(3)]
    raw.sort(key=lambda b: -b["super_edge_index"].shape[1])
    noise = [{k: t(v, DEV) for k, v in draw_noise(b, seed=800 + i).items()} for i, b in enumerate(raw)]
    tr, tp = _trainer(use_graph=True), _trainer(use_graph=False)
    for b, nz in zip(raw, noise):
        bt = pg.Batch.from_numpy(b, DEV)
        assert bt._canonical == "permutation"
        a, c = float(tr.step(bt, nz)), float(tp.step(pg.Batch.from_numpy(b, DEV), nz))
        assert abs(a - c) <= 2e-6 * abs(c), (a, c)
    assert tr.step_graphs.captures == 1 and _bucket_of(tr).option == "permutation"
    assert rel_err(tr.flat.flat, tp.flat.flat) < 1e-5
    monkeypatch.setenv("GEOSSL_NO_BUCKETS", "1")
    tr2 = _trainer(use_graph=True)
    for rep in range(2):
        for b, nz in zip(raw, noise):
            tr2.step(pg.Batch.from_numpy(b, DEV), nz)
    assert tr2.step_graphs.captures == 3 and all(not (isinstance(k, tuple) and k[0] == "bucket") for k in tr2._graphs)


# ---------------------------------------------------------------- PaiNN forces on the fused kernels

PAINN = dict(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add")


def _painn_model_and_batch(sizes, seed, cfg=PAINN):
    from filler import fill_module_
    from geossl_amd import ops
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.Geom3D.models import PaiNN
    from geossl_amd.synthetic import make_batch
    b = make_batch(0, seed=seed, sizes=sizes)
    bt = pg.Batch.from_numpy(b, DEV)
    bt.x[:, 0].clamp_(max=8)
    bt.radius_edge_index = ops.radius_graph(bt.positions, cfg["cutoff"], bt.batch)
    return fill_module_(PaiNN(**cfg)).to(DEV), bt


@pytest.mark.parametrize("F_,R", [(128, 20), (64, 16)])
def test_painn_forces_on_fused_kernels_vs_fp64_oracle(F_, R):
    """finetune_md17.py:46, first order: -dE/dpos of the PaiNN backbone by the fused node (painn_force.hip: per-edge
    gradients of the radial basis, the cutoff and the direction, then the edge-geometry derivative) against
    oracle.nets.painn_forward differentiated in fp64; the parameter gradients of the same backward are bit-identical
    to a backward that does not ask for the positions; the net force on every molecule vanishes."""
    from oracle import nets
    from test_oracle_golden import painn_params
    cfg = dict(PAINN, n_atom_basis=F_, n_rbf=R)
    model, bt = _painn_model_and_batch([1, 2, 3, 18, 18, 40, 9, 33, 5], seed=31, cfg=cfg)
    wgt = torch.cos(torch.arange(F_, dtype=torch.float32, device=DEV))
    pos = bt.positions.clone().requires_grad_(True)
    rep = model(bt.x, pos, bt.radius_edge_index, bt.batch)
    energy = (rep * wgt).sum(dim=1)
    force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), retain_graph=True)[0]
    P64 = {k: v.double() for k, v in painn_params(cfg).items()}
    p64 = bt.positions.cpu().double().requires_grad_(True)
    rep64 = nets.painn_forward(P64, bt.x.cpu(), p64, bt.radius_edge_index.cpu(), bt.batch.cpu(), F_, 3, 5.0, "add")
    e64 = (rep64 * wgt.cpu().double()).sum(dim=1)
    f64 = -torch.autograd.grad(e64.sum(), p64)[0]
    assert rel_err(energy.detach().cpu().double(), e64.detach()) < TOL_OUT
    assert rel_err(force.cpu().double(), f64) < TOL_GRAD
    net = torch.zeros(9, 3, dtype=torch.float64).index_add_(0, bt.batch.cpu(), force.cpu().double())
    assert float(net.abs().max()) < 1e-4 * float(f64.abs().max())
    model.zero_grad()
    energy.sum().backward(inputs=[pos] + list(model.parameters()))
    with_pos = {k: v.clone() for k, v in unique_named_grads(model).items()}
    assert rel_err(pos.grad.cpu().double(), -f64) < TOL_GRAD
    model.zero_grad()
    (model(bt.x, bt.positions, bt.radius_edge_index, bt.batch) * wgt).sum().backward()
    for k, v in unique_named_grads(model).items():
        assert torch.equal(v, with_pos[k]), k


def test_painn_force_evaluation_launches_the_fused_node_only(monkeypatch):
    """The reference's evaluation loop asks for create_graph=True and detaches (finetune_md17.py:99): the PaiNN force is
    the fused kernels' either way (bit-identical), and the primitive restatement is never built."""
    from geossl_amd import higher_order

    def never(*a, **k):
        raise AssertionError("the primitive route ran during a first-order force evaluation")
    monkeypatch.setattr(higher_order, "painn_atom_features", never)
    model, bt = _painn_model_and_batch([12, 7, 21, 3], seed=8)
    forces = []
    for create_graph in (False, True):
        pos = bt.positions.clone().requires_grad_(True)
        energy = model(bt.x, pos, bt.radius_edge_index, bt.batch).sum(dim=1)
        f = torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=create_graph, retain_graph=True)[0]
        assert f.requires_grad == create_graph
        forces.append(f.detach())
    assert torch.equal(forces[0], forces[1])
    # frozen weights: positions only
    for p in model.parameters():
        p.requires_grad_(False)
    pos = bt.positions.clone().requires_grad_(True)
    energy = model(bt.x, pos, bt.radius_edge_index, bt.batch).sum(dim=1)
    f = torch.autograd.grad(energy.sum(), pos)[0]
    assert torch.equal(f, forces[0])


def test_painn_edge_gradient_kernels_through_the_c_abi():
    """geossl_painn_edge_grads / edge_geom_bwd / position_grad on their own (include/geossl_hip.h) against autograd of
    the same per-edge expressions in fp64: one interaction block, accumulate = 0 then 1 (twice the gradient)."""
    from geossl_amd._lib import call, ptr, stream
    from geossl_amd.layout import EdgeLayout
    torch.manual_seed(5)
    N, F_, R, cutoff = 23, 128, 20, 5.0
    pos = torch.randn(N, 3, device=DEV) * 1.5
    batch = torch.zeros(N, dtype=torch.long, device=DEV)
    d = torch.cdist(pos, pos)
    ii, jj = torch.nonzero((d < cutoff) & ~torch.eye(N, dtype=torch.bool, device=DEV), as_tuple=True)
    ei = torch.stack([ii, jj])
    el = EdgeLayout(batch, ei, 1)
    E = el.E
    offsets = torch.linspace(0.0, cutoff, R, device=DEV)
    widths = torch.full((R,), float(offsets[1] - offsets[0]), device=DEV)
    Wf, bf = torch.randn(3 * F_, R, device=DEV) * 0.3, torch.randn(3 * F_, device=DEV) * 0.1
    gq, gmu = torch.randn(N, F_, device=DEV), torch.randn(N, 3, F_, device=DEV)
    mu, xc = torch.randn(N, 3, F_, device=DEV), torch.randn(N, 3 * F_, device=DEV)
    f32 = dict(dtype=torch.float32, device=DEV)
    dirv, fcut, phi = torch.empty(E, 3, **f32), torch.empty(E, **f32), torch.empty(E, R, **f32)
    st = stream()
    call("geossl_painn_edge_geom", ptr(pos), ptr(el.idx_i), ptr(el.idx_j), E, cutoff, ptr(offsets), ptr(widths), R,
         ptr(dirv), ptr(fcut), ptr(phi), st)
    dphi, dfc, ddir = torch.empty(E, R, **f32), torch.empty(E, **f32), torch.empty(E, 3, **f32)
    for acc in (0, 1):
        call("geossl_painn_edge_grads", ptr(gq), ptr(gmu), ptr(mu), ptr(xc), ptr(el.idx_i), ptr(el.idx_j), ptr(phi),
             ptr(fcut), ptr(dirv), ptr(Wf), ptr(bf), E, F_, R, ptr(dphi), ptr(dfc), ptr(ddir), acc, st)
    dr, dpos = torch.empty(E, 3, **f32), torch.empty(N, 3, **f32)
    call("geossl_painn_edge_geom_bwd", ptr(pos), ptr(el.idx_i), ptr(el.idx_j), E, cutoff, ptr(offsets), ptr(widths), R,
         ptr(dphi), ptr(dfc), ptr(ddir), ptr(dr), st)
    (pi, ix), (pj, jx) = el.inc["i"], el.inc["j"]
    call("geossl_painn_position_grad", ptr(dr), ptr(pi), ptr(ix), ptr(pj), ptr(jx), N, ptr(dpos), st)
    # the same in fp64 by autograd (painn.py:232-241,56-61)
    D = lambda v: v.double().cpu()
    p64 = D(pos).requires_grad_(True)
    i_, j_ = ii.cpu(), jj.cpu()
    r = p64[i_] - p64[j_]
    dist = r.norm(dim=1, keepdim=True)
    u = r / dist
    ph = torch.exp((-0.5 / D(widths) ** 2) * (dist - D(offsets)) ** 2)
    fc = 0.5 * (torch.cos(dist * np.pi / cutoff) + 1.0) * (dist < cutoff)
    W = (ph @ D(Wf).T + D(bf)) * fc
    x = W * D(xc)[j_]
    m0, m1, m2 = torch.split(x, F_, dim=1)
    dmu_e = m1[:, None, :] * u[:, :, None] + m2[:, None, :] * D(mu)[j_]
    val = (m0 * D(gq)[i_]).sum() + (dmu_e * D(gmu)[i_]).sum()
    ref = torch.autograd.grad(val, p64)[0]
    assert rel_err(dpos.cpu().double(), 2.0 * ref) < 1e-5


def test_standalone_dense_and_silu_nodes_to_second_order():
    """Dense as a module of its own (the energy head of create_output_layers(), painn_utils.py:9-35): bias in the GEMM's
    epilogue, F.silu on geossl_silu_fwd / geossl_silu_bwd - values, first derivatives and the derivative of a
    gradient (what training on forces differentiates, finetune_md17.py:46-54) against torch in fp64."""
    import torch.nn.functional as Fn
    from geossl_amd.Geom3D.models.painn import Dense
    torch.manual_seed(9)
    for n_in, n_out, act in ((128, 64, Fn.silu), (64, 1, None), (20, 384, None)):
        lay = Dense(n_in, n_out, activation=act).to(DEV)
        torch.nn.init.normal_(lay.bias, std=0.3)
        x = torch.randn(37, n_in, device=DEV, requires_grad=True)
        c = torch.randn(37, n_out, device=DEV)
        y = lay(x)
        (gx,) = torch.autograd.grad(y, x, c, create_graph=True)
        loss = (gx ** 2).sum() + (y ** 2).sum()
        loss.backward()
        x64 = x.detach().double().cpu().requires_grad_(True)
        w64, b64 = lay.weight.detach().double().cpu().requires_grad_(True), lay.bias.detach().double().cpu().requires_grad_(True)
        y64 = Fn.linear(x64, w64, b64)
        y64 = Fn.silu(y64) if act is not None else y64
        (gx64,) = torch.autograd.grad(y64, x64, c.double().cpu(), create_graph=True)
        ((gx64 ** 2).sum() + (y64 ** 2).sum()).backward()
        assert rel_err(y.detach().cpu().double(), y64.detach()) < TOL_OUT
        assert rel_err(gx.detach().cpu().double(), gx64.detach()) < TOL_GRAD
        assert rel_err(x.grad.cpu().double(), x64.grad) < TOL_GRAD
        assert rel_err(lay.weight.grad.cpu().double(), w64.grad) < TOL_GRAD
        assert rel_err(lay.bias.grad.cpu().double(), b64.grad) < TOL_GRAD


# ---------------------------------------------------------------- small launches of a step (round 4, second half)

def test_flat_pair_geometry_is_the_general_kernel_bit_for_bit():
    """geossl_pair_geometry on molecules of at most cap atoms takes the flat form (pair slots dealt to the lanes, no
    adjacency pass): distances, envelope and flags bit-identical to the general kernel (forced by declaring max_n above
    the cap), ragged sizes 1 .. 33 and a radius that cuts pairs."""
    from geossl_amd._lib import call, ptr, stream
    from geossl_amd.layout import MolLayout
    from geossl_amd.synthetic import make_batch
    b = make_batch(0, seed=17, sizes=[1, 2, 33, 18, 3, 27, 33, 5, 18, 1, 9, 31, 2])
    pos = t(b["positions"], DEV) * 1.7   # (spread: some pairs beyond the radius)
    lay = MolLayout(t(b["batch"], DEV), len(b["sizes"]))
    assert lay.max_n == 33
    outs = []
    for max_n in (lay.max_n, 65):
        d = torch.full((lay.P,), -1.0, device=DEV)
        c = torch.full((lay.P,), -1.0, device=DEV)
        fl = torch.full((lay.P,), 9, dtype=torch.uint8, device=DEV)
        call("geossl_pair_geometry", ptr(pos), ptr(lay.mol_ptr), ptr(lay.pair_ptr), lay.B, max_n, 25.0, 33, 5.0, ptr(d),
             ptr(c), ptr(fl), stream())
        outs.append((d, c, fl))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    assert 0 < int((outs[0][2] == 0).sum()) < lay.P and set(outs[0][2].unique().tolist()) <= {0, 3}


def test_copy_n_fills_with_zeros_where_the_source_is_null():
    from geossl_amd import _lib
    from geossl_amd._lib import call, ptr, stream
    src = torch.arange(1000, dtype=torch.int32, device=DEV)
    dst = torch.full((1008,), -1, dtype=torch.int32, device=DEV)
    z = torch.full((5000,), 3.0, device=DEV)
    cb = _lib.CopyBatch()
    cb.dst[0], cb.src[0], cb.bytes[0] = ptr(dst), ptr(src), 4000
    cb.dst[1], cb.src[1], cb.bytes[1] = ptr(z), None, 4 * 4992
    call("geossl_copy_n", C.byref(cb), 2, stream())
    assert torch.equal(dst[:1000], src) and bool((dst[1000:] == -1).all())
    assert bool((z[:4992] == 0).all()) and bool((z[4992:] == 3.0).all())


def test_noise_key_by_value_is_the_device_seed_path_and_follows_the_generator():
    """geossl_ddm_noise_seeded(key) == geossl_ddm_noise(&key in device memory); the trainer's key follows torch's CUDA
    generator: the same manual seed gives the same draws, a torch draw in between moves them."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd._lib import call, ptr, stream
    from geossl_amd.synthetic import make_batch
    key = 0x1234ABCD5678EF01
    outs = []
    for by_value in (False, True):
        t_ = {"p": torch.empty(300, device=DEV), "l1": torch.empty(7, dtype=torch.long, device=DEV),
              "d1": torch.empty(50, device=DEV), "l2": torch.empty(7, dtype=torch.long, device=DEV), "d2": torch.empty(50, device=DEV)}
        seed = torch.tensor([key], dtype=torch.long, device=DEV)
        call("geossl_ddm_noise_seeded" if by_value else "geossl_ddm_noise", key if by_value else ptr(seed), 0.0, 0.3, 300, 50, 7,
             50, 30, ptr(t_["p"]), ptr(t_["l1"]), ptr(t_["d1"]), ptr(t_["l2"]), ptr(t_["d2"]), stream())
        outs.append(t_)
    assert all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0])
    batch = pg.Batch.from_numpy(make_batch(40, seed=2), DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 30, 2, DEV)

    def draws(extra):
        torch.cuda.manual_seed(11)
        if extra:
            torch.rand(4, device=DEV)
        a = {k: v.clone() for k, v in pg.draw_step_noise_fused(batch, n1, n2, 0.0, 0.3).items()}
        b = {k: v.clone() for k, v in pg.draw_step_noise_fused(batch, n1, n2, 0.0, 0.3).items()}
        return a, b
    (a0, b0), (a1, b1), (a2, _) = draws(False), draws(False), draws(True)
    assert all(torch.equal(a0[k], a1[k]) and torch.equal(b0[k], b1[k]) for k in a0)
    assert not torch.equal(a0["pos_noise"], b0["pos_noise"]) and not torch.equal(a0["pos_noise"], a2["pos_noise"])


@pytest.mark.parametrize("F_", [32, 64, 128, 48])
def test_embedding_forward_vector_form_matches_the_table(F_):
    """geossl_embedding_fwd: four columns per thread for F in (32, 64, 128), the generic form otherwise - rows of the
    table bit for bit, an out-of-range class flags the status word and gives a zero row."""
    from geossl_amd._lib import call, ptr, stream
    torch.manual_seed(1)
    table = torch.randn(9, F_, device=DEV)
    x = torch.randint(0, 9, (1237, 2), device=DEV)
    x[5, 0], x[77, 0] = 9, -1
    out = torch.full((1237, F_), 7.0, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    call("geossl_embedding_fwd", ptr(x), 2, ptr(table), 9, 1237, F_, ptr(out), ptr(status), stream())
    want = table[x[:, 0].clamp(0, 8)]
    want[5], want[77] = 0.0, 0.0
    assert torch.equal(out, want) and int(status) == 1


def test_painn_forces_without_edges_and_under_no_grad():
    """Single-atom molecules (no edges): the force is zero and finite; a forward under no_grad with positions that
    require a gradient saves nothing and matches the forward that does."""
    model, bt = _painn_model_and_batch([1, 1, 1], seed=3)
    assert bt.radius_edge_index.size(1) == 0
    pos = bt.positions.clone().requires_grad_(True)
    e = model(bt.x, pos, bt.radius_edge_index, bt.batch).sum()
    (f,) = torch.autograd.grad(e, pos)
    assert f.shape == pos.shape and float(f.abs().max()) == 0.0
    model2, bt2 = _painn_model_and_batch([7, 12, 3], seed=4)
    pos2 = bt2.positions.clone().requires_grad_(True)
    with torch.no_grad():
        a = model2(bt2.x, pos2, bt2.radius_edge_index, bt2.batch)
    b = model2(bt2.x, pos2, bt2.radius_edge_index, bt2.batch)
    assert not a.requires_grad and b.requires_grad and torch.equal(a, b.detach())


def test_trainer_gradients_are_cleared_across_graph_kinds_and_eager_steps():
    """The flat gradient buffer is cleared by the refresh launch of a replayed step (bucket graphs and per-structure graphs
    alike) and by the step itself when it runs eagerly: the same batch gives the same gradient whatever ran before."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    tr = _trainer(use_graph=True)
    ragged = [make_batch(0, seed=40 + i, sizes=_ragged_sizes(24, 40 + i)) for i in range(2)]
    uniform = make_batch(24, seed=50)
    probe, pn = ragged[0], draw_noise(ragged[0], seed=9)
    seen = []

    def grad_of_probe():
        bt = pg.Batch.from_numpy(probe, DEV)
        tr._graph_fwd_bwd(bt, {k: t(v, DEV) for k, v in pn.items()})
        return tr.flat.grad.clone()
    # bucket graph (captured on the probe), then another ragged batch, a uniform batch (its own graph), an eager step
    seen.append(grad_of_probe())
    tr._graph_fwd_bwd(pg.Batch.from_numpy(ragged[1], DEV), None)
    seen.append(grad_of_probe())
    tr._graph_fwd_bwd(pg.Batch.from_numpy(uniform, DEV), None)
    seen.append(grad_of_probe())
    tr._fwd_bwd(pg.Batch.from_numpy(ragged[1], DEV), None)
    seen.append(grad_of_probe())
    assert float(seen[0].abs().max()) > 0
    for g_ in seen[1:]:
        assert torch.equal(g_, seen[0])
    assert tr.step_graphs.captures == 2
