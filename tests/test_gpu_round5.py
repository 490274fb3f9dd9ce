"""GPU tests added in round 5: the advisor's findings of round 4 (single-atom molecule at the end of a ragged layer
loop, bucket tensor checks, stock Adam with weight decay, gradient clearing under a caller's own capture, the autograd
thread switch), BASELINE config 1 (finetune_qm9.py) on the HIP path against the reference's own train() / eval(), and
the capacity buckets of this round (molecules above 33 atoms at 10 A with the neighbour cap, PaiNN)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from helpers import (fill_module_, grad_summary, ncsn_oracle_params, product_ncsn, product_schnet,
                     schnet_oracle_params, t, unique_named_grads)

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


def _ragged_sizes(B, seed, lo=2, hi=33, mean=18.0, sd=4.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(mean, sd, size=B)), lo, hi).astype(np.int64)


def _trainer(cfg=SMALL, **kw):
    from geossl_amd import pretrain_GeoSSL as pg
    return pg.DDMTrainer(product_schnet(cfg, DEV), product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9),
                         lr=5e-4, **kw)


# ------------------------------------------------------------------------------------------- advisor, round 4
def test_ragged_layer_loop_with_a_trailing_single_atom_molecule(monkeypatch):
    """in the ragged layer loop (k_layer_loop<0>) a one-atom molecule at the END of the batch has
    pair_ptr[m] == P; its aggregation must write the zero row without requesting "slot 0", which lies one row past the
    [L, P, F] filter tensor.  The per-structure graph of such a batch (buckets off: the loop runs on exact-size tensors)
    gives the eager launches' losses and gradients bit for bit, single-atom molecules in the middle and at the end."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    monkeypatch.setenv("GEOSSL_NO_BUCKETS", "1")
    sizes = _ragged_sizes(12, 3)
    sizes[4], sizes[-1] = 1, 1
    b = make_batch(0, seed=70, sizes=sizes)
    nz = {k: t(v, DEV) for k, v in draw_noise(b, seed=71).items()}
    out = {}
    for use_graph in (False, True):
        tr = _trainer(use_graph=use_graph, graph_mode="structure")
        bt = pg.Batch.from_numpy(b, DEV)
        losses = [float(tr.step(bt, nz)) for _ in range(3)]
        out[use_graph] = (losses, tr.flat.flat.clone())
        if use_graph:
            assert tr.step_graphs.captures == 1
    assert out[True][0] == out[False][0] and torch.equal(out[True][1], out[False][1])
    assert all(np.isfinite(v) for v in out[True][0])


def test_bucket_refuses_tensors_it_cannot_copy():
    """Bucket.fill copies x / batch / super_edge_index by byte count as int64.  A batch whose x is
    int32, or whose super_edge_index was cut after the collation marked it canonical, is not bucket-eligible (it keeps
    its per-structure graph / the eager path) instead of making the copy read past the source."""
    from geossl_amd import bucket as bk
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    b = make_batch(0, seed=80, sizes=_ragged_sizes(8, 80))
    good = pg.Batch.from_numpy(b, DEV)
    assert bk.eligible(good, "schnet")
    bad = pg.Batch.from_numpy(b, DEV)
    bad.x = bad.x.to(torch.int32)
    assert not bk.eligible(bad, "schnet")
    cut = pg.Batch.from_numpy(b, DEV)
    cut.super_edge_index = cut.super_edge_index[:, :-3].contiguous()
    assert not bk.eligible(cut, "schnet")
    strided = pg.Batch.from_numpy(b, DEV)
    strided.positions = torch.cat([strided.positions, strided.positions], dim=1)[:, :3]
    assert not bk.eligible(strided, "schnet")
    # and a fill that is asked anyway raises instead of copying
    bkt = bk.Bucket(torch.device(DEV), 8, bk.capacities(*bk.batch_counts(bk.sizes_array(good), "combination"), B=8),
                    "combination")
    bkt.fill(good)
    with pytest.raises(ValueError):
        bkt.fill(strided)


def _ref_loop(graph, steps, batches, adam_kw=None, cfg=SMALL, seed=11, decay=0.0):
    from geossl_amd import pretrain_GeoSSL as pg
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
    pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
    args = types.SimpleNamespace(model_3d="schnet", GeoSSL_mu=0.0, GeoSSL_sigma=0.3, lr=5e-4, decay=decay, step_graph=graph)
    group = [{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()}, {"params": n2.parameters()}]
    optimizer = torch.optim.Adam(group, lr=args.lr, weight_decay=args.decay, **(adam_kw or {}))
    losses = []
    try:
        for step in range(steps):
            loss, acc = pg.do_DDM(args, batches[step % len(batches)], model, criterion=None, mu=0.0, sigma=0.3)
            losses.append(loss.detach().item())
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
    finally:
        pg.NCSN_model_01 = pg.NCSN_model_02 = None
    params = torch.cat([p.detach().reshape(-1) for m in (model, n1, n2) for p in m.parameters()]).cpu()
    return losses, params, optimizer


def test_stock_adam_with_weight_decay_on_one_launch_bit_for_bit():
    """the 'bit for bit' claim of the fused stock-Adam step was only checked with weight_decay = 0.
    --decay > 0 (examples/config.py:95, pretrain_GeoSSL.py:343): grad + decay * param as torch's foreach kernels form it."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    batches = [pg.Batch.from_numpy(make_batch(32, seed=90 + i), DEV) for i in range(3)]
    eager = _ref_loop(False, 5, batches, decay=1e-2)
    graph = _ref_loop(True, 5, batches, decay=1e-2)
    assert graph[0] == eager[0] and torch.equal(graph[1], eager[1])
    so, se = graph[2].state_dict(), eager[2].state_dict()
    for k in se["state"]:
        assert torch.equal(so["state"][k]["exp_avg"], se["state"][k]["exp_avg"])
        assert torch.equal(so["state"][k]["exp_avg_sq"], se["state"][k]["exp_avg_sq"])


def test_a_callers_own_capture_of_the_eager_step_clears_the_gradients():
    """DDMTrainer(use_graph=False)._fwd_bwd captured in the CALLER's CUDA graph: the fill that clears
    the flat gradient buffer must be recorded into that graph (only StepGraphs' own captures leave it to their refresh
    launch) - replays must not accumulate gradients."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(16, seed=95)
    bt = pg.Batch.from_numpy(b, DEV)
    nz = {k: t(v, DEV) for k, v in draw_noise(b, seed=96).items()}
    tr = _trainer(use_graph=False)
    tr._fwd_bwd(bt, nz)                     # warm-up: layouts, kernel attributes
    want = tr.flat.grad.clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tr._fwd_bwd(bt, nz)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(tr.flat.grad, want)


def test_autograd_thread_switch_is_put_back_after_the_step():
    """do_DDM's graph path runs ITS loss's backward on the calling thread (autograd's thread-local
    multithreading switch, turned off when the loss is handed out) - and the caller's setting is back after the optimizer
    step that follows, or at the latest at the next do_DDM; other autograd work of the caller sees its own setting."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    pg._restore_backward_threads()
    torch.autograd.set_multithreading_enabled(True)
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
    opt = torch.optim.SGD(list(model.parameters()) + [p for h in heads for p in h.parameters() if p.requires_grad], lr=1e-4)
    bt = pg.Batch.from_numpy(make_batch(16, seed=97), DEV)
    for _ in range(2):
        loss, _ = pg.do_DDM(pg.Args("schnet"), bt, model, NCSN_models=heads, graph=True)
        assert not torch.autograd.is_multithreading_enabled()
        opt.zero_grad()
        loss.backward()
        opt.step()                                  # (any optimizer: a global post-step hook)
        assert torch.autograd.is_multithreading_enabled()
    loss, _ = pg.do_DDM(pg.Args("schnet"), bt, model, NCSN_models=heads, graph=True)
    assert not torch.autograd.is_multithreading_enabled()
    del loss                                        # dropped without backward(): back at the next step's entry
    loss, _ = pg.do_DDM(pg.Args("schnet"), bt, model, NCSN_models=heads, graph=True)
    assert len(pg._MT_PENDING) == 1
    pg._restore_backward_threads()
    assert torch.autograd.is_multithreading_enabled()
    torch.cuda.synchronize()


# -------------------------------------------------------------------------------- BASELINE config 1 on the HIP path
def test_g14_finetune_qm9_on_the_hip_path():
    """examples/finetune_qm9.py:163-275,278-384,500-507 with the product SchNet (config.py defaults: cutoff 10 A, 51
    gaussians, readout mean; bs = 32 QM9-sized molecules): forward -> graph_pred_linear -> L1 on the normalised target ->
    stock Adam, CosineAnnealingLR per epoch, two epochs, then eval() under no_grad - per-step losses, parameters and
    evaluation predictions against the reference's own train() / eval() (fixture G14)."""
    from test_oracle_golden import finetune_epochs
    g = load_golden("g14_finetune_qm9_schnet")
    cfg = json.loads(str(g["cfg"]))
    model = product_schnet(cfg, DEV)
    head = fill_module_(torch.nn.Linear(128, 1)).to(DEV)
    model.train()
    losses, scores, lr = finetune_epochs(g, lambda z, pos, bat: model(z, pos, bat), list(model.parameters()),
                                         head.weight, head.bias, device=DEV)
    assert rel_err(losses, g["losses"]) < TOL_OUT and abs(lr - float(g["lr_after"])) < 1e-12
    # (predictions AFTER six Adam steps: the tolerance class of trained parameters, like the G12 trajectories)
    assert rel_err(scores, g["y_scores"]) < 5e-5
    named = dict(model.named_parameters())
    for k in g:
        if k.startswith("psum/"):
            assert rel_err(grad_summary(named[k[5:]].detach().cpu()), g[k]) < TOL_GRAD, k
    assert rel_err(head.weight.cpu(), g["head/weight"]) < TOL_GRAD and rel_err(head.bias.cpu(), g["head/bias"]) < TOL_GRAD


# ----------------------------------------------------------------- molecules above 33 atoms (Molecule3D with hydrogens)
def _sequential_index_add(xn, Wn, fn, pi, pj, swap):
    """propagate(aggr="add") (schnet.py:190,194-195) as a sequential index_add in ascending source order per target over
    the canonical edge list, in numpy fp32 (separate multiply and add), from the pair-slot form."""
    f = fn if not swap else (((fn & 1) << 1) | ((fn >> 1) & 1))
    src = np.concatenate([pj[(f & 1) != 0], pi[(f & 2) != 0]])
    tgt = np.concatenate([pi[(f & 1) != 0], pj[(f & 2) != 0]])
    slot = np.concatenate([np.nonzero((f & 1) != 0)[0], np.nonzero((f & 2) != 0)[0]])
    order = np.lexsort((src, tgt))
    src, tgt, slot = src[order], tgt[order], slot[order]
    ref = np.zeros_like(xn)
    start = np.concatenate([[0], np.nonzero(np.diff(tgt))[0] + 1])
    rank = np.arange(len(tgt)) - np.repeat(start, np.diff(np.concatenate([start, [len(tgt)]])))
    for r in range(int(rank.max()) + 1):        # one vectorised pass per list position keeps every target's order
        sel = rank == r
        ref[tgt[sel]] = (ref[tgt[sel]] + (xn[src[sel]] * Wn[slot[sel]]).astype(np.float32)).astype(np.float32)
    return ref


@pytest.mark.parametrize("by_targets", [False, True], ids=["parts", "targets"])
@pytest.mark.parametrize("sizes,lds_form", [([34, 18, 64, 1, 100, 33, 2, 47], True), ([255, 40, 3], False)],
                         ids=["upto100", "255"])
def test_aggregation_work_list_above_the_size_classes_is_the_sequential_index_add(sizes, lds_form, by_targets, monkeypatch):
    """Molecules of 34 .. 255 atoms go through the work list as one work item per target atom (aggregate_targets: no size
    class, partners 32 at a time): bit for bit the sequential index_add over the canonical edge list (schnet.py:190,
    194-195), for the graph and its transpose, with asymmetric flags (what the 32-neighbour cap produces); and the LDS
    form of geossl_cfconv_aggregate (molecules whose rows fit the LDS) gives the same bits.  by_targets: the form of small
    launches (the reference's batch size) - EVERY atom of every molecule its own work item."""
    monkeypatch.setenv("GEOSSL_AGG_TARGETS_MAX", "256" if by_targets else "0")
    from geossl_amd import ops
    from geossl_amd._lib import call, ptr, stream
    from geossl_amd.layout import MolLayout
    F = 128
    batch = torch.arange(len(sizes), device=DEV).repeat_interleave(torch.tensor(sizes, device=DEV))
    lay = MolLayout(batch, len(sizes), sizes=sizes)
    parts = lambda n: n if (n > 33 or by_targets) else (4 if n >= 31 else (2 if n >= 27 else 1))
    assert lay.agg_work is not None and int((lay.agg_work != -1).sum()) == sum(parts(n) for n in sizes)
    assert lay.agg_targets == by_targets and lay.agg_work.numel() % 8 == 0
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(lay.N, F, device=DEV, generator=g)
    W = torch.randn(lay.P, F, device=DEV, generator=g)
    flag = torch.randint(0, 4, (lay.P,), device=DEV, generator=g, dtype=torch.uint8)
    xn, Wn, fn = x.cpu().numpy(), W.cpu().numpy(), flag.cpu().numpy()
    pi, pj = lay.pair_i.cpu().numpy(), lay.pair_j.cpu().numpy()
    for swap in (False, True):
        out = ops.aggregate(x, W, flag, lay, swap=swap)
        if lds_form:
            lds = torch.empty_like(x)
            call("geossl_cfconv_aggregate", ptr(x), ptr(W), ptr(flag), ptr(lay.mol_ptr), ptr(lay.pair_ptr), None, lay.B,
                 lay.max_n, F, 1 if swap else 0, ptr(lds), stream())
            assert torch.equal(out, lds)
        assert np.array_equal(out.cpu().numpy(), _sequential_index_add(xn, Wn, fn, pi, pj, swap)), swap


def test_set_c_at_10_angstrom_through_the_bucket_graph_vs_oracle():
    """What the reference's DDM script feeds SchNet (submit_pretrain_GeoSSL_DDM.sh:3,8,22; datasets_Molecule3D.py:65;
    config.py:114): molecules with hydrogens (set C: a quarter above 33 atoms, some above 48), the default 10 A radius, where
    the 32-neighbour cap of radius_graph cuts lists and makes the graph asymmetric.  Through ONE bucket graph of
    DDMTrainer: the edge set of both views bit-exact against the oracle with the cap active, loss <= 1e-5 and backbone
    gradients <= 1e-4 against oracle.nets.do_ddm_schnet, replays on three different batches bit-identical to the eager
    launches on the same bucket."""
    from geossl_amd import bucket as bk
    from geossl_amd import ops
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import graph, nets
    cfg = dict(FULL, cutoff=10.0)
    B = 48
    from geossl_amd.synthetic import molecule_sizes
    specs = [molecule_sizes(B, "C", np.random.default_rng(900 + i)) for i in range(3)]
    for i, s_ in enumerate(specs):
        s_[3 + i], s_[10 + i], s_[-1] = 52 + i, 41, (1 if i == 1 else s_[-1])     # large molecules in every batch, a lone atom
    raws = [make_batch(B, seed=900 + i, sizes=s_) for i, s_ in enumerate(specs)]
    raws.sort(key=lambda b: -int((b["sizes"] * (b["sizes"] - 1)).sum()))          # (most pair slots first: one capture)
    assert all(40 < int(b["sizes"].max()) <= 64 for b in raws) and len({tuple(b["sizes"]) for b in raws}) == 3
    nzs = [draw_noise(b, seed=910 + i) for i, b in enumerate(raws)]
    # ---- the edge set with the cap active, both views of batch 0
    b, nz = raws[0], nzs[0]
    pos2 = np.concatenate([b["positions"], (b["positions"] + nz["pos_noise"]).astype(np.float32)])
    bat2 = np.concatenate([b["batch"], b["batch"] + B])
    want = graph.radius_graph_np(pos2, 10.0, bat2)
    deg = np.bincount(want[1], minlength=len(bat2))
    n2 = np.concatenate([b["sizes"], b["sizes"]])
    assert int((deg == 32).sum()) > 0 and want.shape[1] < int((n2 * (n2 - 1)).sum())      # the cap did cut lists
    got = ops.radius_graph(t(pos2, DEV), 10.0, t(bat2, DEV))
    assert torch.equal(got.cpu(), t(want))
    # ---- the step through the bucket graph
    tr = _trainer(cfg, use_graph=True)
    losses, grads = [], []
    for b_, nz_ in zip(raws, nzs):
        losses.append(tr._graph_fwd_bwd(pg.Batch.from_numpy(b_, DEV), {k: t(v, DEV) for k, v in nz_.items()}).clone())
        grads.append(tr.flat.grad.clone())
    assert tr.use_graph and tr.step_graphs.captures == 1 and len(tr._graphs) == 1
    bkt = next(iter(tr._graphs.values()))["bucket"]
    assert bkt.max_n in (64, 128)                    # (above the size classes: a bound with head room)
    # the same launches eagerly on a bucket of the same capacity
    te = _trainer(cfg, use_graph=False)
    eb = bk.Bucket(torch.device(DEV), B, bkt.caps(), "combination", max_n=bkt.max_n)
    f32 = dict(dtype=torch.float32, device=DEV)
    sn = {"pos_noise": torch.zeros(eb.N_cap, 3, **f32), "dist_noise_1": torch.zeros(eb.S_cap, 1, **f32),
          "dist_noise_2": torch.zeros(eb.S_cap, 1, **f32), "noise_level_1": torch.zeros(B, dtype=torch.long, device=DEV),
          "noise_level_2": torch.zeros(B, dtype=torch.long, device=DEV)}
    for i, (b_, nz_) in enumerate(zip(raws, nzs)):
        N, P, S, W = eb.fill(pg.Batch.from_numpy(b_, DEV))
        sn["pos_noise"][:N].copy_(t(nz_["pos_noise"], DEV))
        sn["dist_noise_1"][:S].copy_(t(nz_["dist_noise_1"], DEV))
        sn["dist_noise_2"][:S].copy_(t(nz_["dist_noise_2"], DEV))
        sn["noise_level_1"].copy_(t(nz_["noise_level_1"], DEV))
        sn["noise_level_2"].copy_(t(nz_["noise_level_2"], DEV))
        loss = te._fwd_bwd(eb.batch, sn)
        assert torch.equal(loss, losses[i]) and torch.equal(te.flat.grad, grads[i]), i
    # ---- against the oracle (batch 0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    Pm, P1, P2 = schnet_oracle_params(cfg), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 10.0, 6, 2, "mean")
    ref.backward()
    assert rel_err(losses[0].cpu(), ref.detach()) < TOL_OUT
    tr.flat.grad.copy_(grads[0])
    g = unique_named_grads(tr.model)
    for k in ("lin2.weight", "interactions.0.mlp.0.weight", "interactions.3.mlp.2.weight", "interactions.5.conv.lin1.weight",
              "interactions.2.conv.lin2.weight", "embedding.weight"):
        assert rel_err(g[k].cpu(), Pm[k].grad) < TOL_GRAD, k


# ----------------------------------------------------------------------------------------------- PaiNN on a bucket
def _painn_modules():
    from geossl_amd.Geom3D.models import PaiNN
    cfg = dict(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add")
    return cfg, fill_module_(PaiNN(**cfg)).to(DEV)


def _painn_batch(raw):
    from geossl_amd import ops
    from geossl_amd import pretrain_GeoSSL as pg
    bt = pg.Batch.from_numpy(raw, DEV)
    bt.radius_edge_index = ops.radius_graph(bt.positions, 5.0, bt.batch)     # datasets_3D_Radius.py:120, on the clean geometry
    return bt


def test_painn_edge_layout_kernel_builds_what_the_edge_layout_builds():
    """geossl_painn_edge_layout (one launch on the batch's own radius_edge_index) against layout.EdgeLayout over the
    concatenated two-view batch (two dozen launches): idx_i / idx_j, both incidence lists entry for entry, and the
    four-row group layout of the matrix-pipe forward molecule by molecule (same rows, same group codes; the kernel
    leaves gaps between molecules and ends a molecule's groups at mol_grp_end); atoms past the real count get empty
    lists; molecules with a single atom and without edges included."""
    from geossl_amd import bucket as bk
    from geossl_amd import ops
    from geossl_amd._lib import call, ptr, stream
    from geossl_amd.layout import EdgeLayout, MolLayout
    from geossl_amd.synthetic import make_batch
    sizes = np.array([18, 1, 33, 2, 60, 9, 1], dtype=np.int64)
    raw = make_batch(0, seed=31, sizes=sizes)
    raw["positions"][18] += 40.0                         # (the single atom stays alone anyway)
    raw["positions"][19 + 33:19 + 35] += np.array([[0, 0, 0], [30.0, 0, 0]], dtype=np.float32)   # the 2-atom molecule: no edge
    bt = _painn_batch(raw)
    e = bt.radius_edge_index
    E, N, B = int(e.size(1)), int(sizes.sum()), len(sizes)
    b2 = torch.cat([bt.batch, bt.batch + B])
    e2 = torch.cat([e, e + N], dim=1)
    lay2 = MolLayout(b2, 2 * B, sizes=list(sizes) + list(sizes))
    want = EdgeLayout(b2, e2, 2 * B)
    wr, wg, wp, wm = want.groups("i", lay2.mol_ptr)
    Ncap2, Ecap = 2 * N + 37, E + 100
    i64, i32 = dict(dtype=torch.int64, device=DEV), dict(dtype=torch.int32, device=DEV)
    from geossl_amd import _lib
    G = int(_lib.load().geossl_painn_group_capacity(2 * Ecap, Ncap2))
    idx_i, idx_j = torch.full((2 * Ecap,), -7, **i64), torch.full((2 * Ecap,), -7, **i64)
    ip_i, il_i = torch.full((Ncap2 + 1,), -7, **i64), torch.full((2 * Ecap,), -7, **i32)
    ip_j, il_j = torch.full((Ncap2 + 1,), -7, **i64), torch.full((2 * Ecap,), -7, **i32)
    row_edge, grp_atom = torch.full((4 * G,), -9, **i32), torch.full((G,), -9, **i32)
    mol_grp, mol_end, status = torch.zeros(2 * B + 1, **i32), torch.zeros(2 * B, **i32), torch.zeros(1, **i32)
    one_view = MolLayout(bt.batch, B, sizes=list(sizes))
    call("geossl_painn_edge_layout", ptr(e[0]), ptr(e[1]), E, ptr(one_view.mol_ptr), N, B, Ncap2, ptr(idx_i), ptr(idx_j),
         ptr(ip_i), ptr(il_i), ptr(ip_j), ptr(il_j), ptr(row_edge), ptr(grp_atom), ptr(mol_grp), ptr(mol_end), ptr(status),
         stream())
    assert int(status) == 0
    assert torch.equal(idx_i[:2 * E], e2[0]) and torch.equal(idx_j[:2 * E], e2[1]) and int(idx_i[2 * E]) == -7
    for (ip, il), side in (((ip_i, il_i), "i"), ((ip_j, il_j), "j")):
        wptr, widx = want.inc[side]
        assert torch.equal(ip[:2 * N + 1], wptr) and torch.equal(il[:2 * E], widx[:2 * E])
        assert bool((ip[2 * N:] == 2 * E).all())                      # atoms past the real count: empty lists
    mg, me, wmg = mol_grp.cpu().numpy(), mol_end.cpu().numpy(), wm.cpu().numpy()
    re_, ga, wre, wga = row_edge.cpu().numpy(), grp_atom.cpu().numpy(), wr.cpu().numpy(), wg.cpu().numpy()
    assert np.all(mg[1:2 * B] >= me[:2 * B - 1])                      # ranges in order, not overlapping
    for m in range(2 * B):
        ng = wmg[m + 1] - wmg[m]
        assert me[m] - mg[m] == ng, m
        assert np.array_equal(re_[4 * mg[m]:4 * me[m]], wre[4 * wmg[m]:4 * wmg[m + 1]]), m
        assert np.array_equal(ga[mg[m]:me[m]], wga[wmg[m]:wmg[m + 1]]), m
    # an edge that leaves its molecule is reported and left out
    bad = e.clone()
    bad[1, 5] = N - 1
    call("geossl_painn_edge_layout", ptr(bad[0]), ptr(bad[1]), E, ptr(one_view.mol_ptr), N, B, Ncap2, ptr(idx_i), ptr(idx_j),
         ptr(ip_i), ptr(il_i), ptr(ip_j), ptr(il_j), ptr(row_edge), ptr(grp_atom), ptr(mol_grp), ptr(mol_end), ptr(status),
         stream())
    assert int(status) == 1


def test_painn_bucket_replays_on_different_edge_lists_bit_for_bit_and_matches_the_oracle():
    """PaiNN's radius_edge_index is geometry-dependent (datasets_3D_Radius.py:120): no two batches of a loader share it.
    DDMTrainer(model_3d="painn", use_graph=True) serves them from ONE captured graph (a capacity bucket whose edge
    structures are rewritten per step by geossl_painn_edge_layout): three batches with three edge lists - one capture,
    losses and gradients bit-identical to the same launches made eagerly on the bucket, within 1e-5 / 1e-4 of
    oracle.nets.do_ddm_painn, and equal to the plain eager step on the batch itself within fp32 summation order."""
    from geossl_amd import bucket as bk
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    from test_oracle_golden import painn_params
    B = 24
    specs = [_ragged_sizes(B, 20 + i) for i in range(3)]
    specs[1][5], specs[2][-1] = 1, 1
    specs.sort(key=lambda s: -int((s * (s - 1)).sum()))
    raws = [make_batch(B, seed=700 + i, sizes=s) for i, s in enumerate(specs)]
    for r in raws:
        r["x"][::5, 0] = 0                                    # hydrogens: the padding row (painn.py:174)
    nzs = [draw_noise(r, seed=710 + i) for i, r in enumerate(raws)]
    bts = [_painn_batch(r) for r in raws]
    assert len({int(b.radius_edge_index.size(1)) for b in bts}) == 3

    def trainer(use_graph):
        cfg, model = _painn_modules()
        return pg.DDMTrainer(model, product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4,
                             model_3d="painn", use_graph=use_graph)
    tr = trainer(True)
    losses, grads = [], []
    for bt, nz in zip(bts, nzs):
        losses.append(tr._graph_fwd_bwd(bt, {k: t(v, DEV) for k, v in nz.items()}).clone())
        grads.append(tr.flat.grad.clone())
    assert tr.use_graph and tr.step_graphs.captures == 1 and len(tr._graphs) == 1
    key = next(iter(tr._graphs))
    assert key[0] == "bucket"
    bkt = tr._graphs[key]["bucket"]
    assert bkt.kind == "painn" and int(bkt.el.status) == 0
    # ---- the same launches eagerly on a bucket of the same capacity
    te = trainer(False)
    eb = bk.Bucket(torch.device(DEV), B, bkt.caps(), "combination", max_n=bkt.max_n, kind="painn", E_cap=bkt.E_cap)
    f32 = dict(dtype=torch.float32, device=DEV)
    sn = {"pos_noise": torch.zeros(eb.N_cap, 3, **f32), "dist_noise_1": torch.zeros(eb.S_cap, 1, **f32),
          "dist_noise_2": torch.zeros(eb.S_cap, 1, **f32), "noise_level_1": torch.zeros(B, dtype=torch.long, device=DEV),
          "noise_level_2": torch.zeros(B, dtype=torch.long, device=DEV)}
    for i, (bt, nz) in enumerate(zip(bts, nzs)):
        N, P, S, W = eb.fill(bt)
        sn["pos_noise"][:N].copy_(t(nz["pos_noise"], DEV))
        sn["dist_noise_1"][:S].copy_(t(nz["dist_noise_1"], DEV))
        sn["dist_noise_2"][:S].copy_(t(nz["dist_noise_2"], DEV))
        sn["noise_level_1"].copy_(t(nz["noise_level_1"], DEV))
        sn["noise_level_2"].copy_(t(nz["noise_level_2"], DEV))
        loss = te._fwd_bwd(eb.batch, sn)
        assert torch.equal(loss, losses[i]) and torch.equal(te.flat.grad, grads[i]), i
    # ---- the plain eager step on the batches themselves
    tp = trainer(False)
    for i, (bt, nz) in enumerate(zip(bts, nzs)):
        loss = tp._fwd_bwd(bt, {k: t(v, DEV) for k, v in nz.items()})
        assert abs(float(loss) - float(losses[i])) <= 2e-6 * abs(float(loss)), i
        assert rel_err(tp.flat.grad, grads[i]) < 1e-5, i
    # ---- the oracle (batch 1: a single-atom molecule in the middle)
    raw, nz, bt = raws[1], nzs[1], bts[1]
    cfg, _ = _painn_modules()
    Pm, P1, P2 = painn_params(cfg), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_painn(Pm, P1, P2, t(raw["x"]), t(raw["positions"]), t(raw["batch"]), bt.radius_edge_index.cpu(),
                            t(raw["super_edge_index"]), t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]),
                            t(nz["noise_level_2"]), t(nz["dist_noise_2"]), 128, 3, 5.0, 2)
    ref.backward()
    assert rel_err(losses[1].cpu(), ref.detach()) < TOL_OUT
    tr.flat.grad.copy_(grads[1])
    g = unique_named_grads(tr.model)
    for k in ("filter_net.weight", "interactions.0.interatomic_context_net.0.weight",
              "interactions.2.interatomic_context_net.1.weight", "mixing.1.mu_channel_mix.weight",
              "mixing.0.intraatomic_context_net.0.weight", "embedding.weight"):
        assert rel_err(g[k].cpu(), Pm[k].grad) < TOL_GRAD, k


# ------------------------------------------------------------------------------------------- eight ranks, one GPU
def test_bench_eight_ranks_share_one_gpu(tmp_path):
    """`python bench.py --gpus 8` as the driver launches it on an 8-GPU node, here with all eight ranks on the one GPU of
    the box over gloo (no 8-GPU node was ever available to a round): rendezvous on 127.0.0.1, eight different per-rank
    molecule sets and noise streams (eight different losses), ONE flat all-reduce per step with the 1 / world factor in the
    Adam launch - the parameters of all eight ranks are bit-identical after three steps - and rank 0's JSON line says
    n_gpus 8, dp8, world_size_initialised 8."""
    import subprocess
    import sys
    env = dict(os.environ, GEOSSL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GEOSSL_BENCH_RANK_LOSS=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--mols", "128", "--dataset-mols", "8192"]
    with open(tmp_path / "out.log", "w") as fo, open(tmp_path / "err.log", "w") as fe:
        p = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, cwd=REPO)
        try:
            code = p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            code = None
            p.kill()
            p.wait()
    out_text, err_text = open(tmp_path / "out.log").read(), open(tmp_path / "err.log").read()
    assert code == 0, out_text[-1500:] + "\n" + err_text[-1500:]
    lines = [ln for ln in out_text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp8"
    assert out["config"]["backend"] == "gloo" and out["config"]["world_size_initialised"] == 8
    assert np.isfinite(out["value"]) and out["value"] > 0
    assert abs(out["value"] - 8 * 128 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-4 * out["value"]   # (the line's 6 digits)
    losses = [float(open(tmp_path / ("loss_rank%d.txt" % r)).read()) for r in range(8)]
    assert all(np.isfinite(v) for v in losses) and len(set(losses)) == 8     # eight molecule sets, eight noise streams
    params = [open(tmp_path / ("params_rank%d.txt" % r)).read() for r in range(8)]
    assert len(set(params)) == 1 and len(params[0]) == 64                    # one all-reduced gradient, one update


# --------------------------------------------------------------------------------------- inference as one graph launch
def test_graphed_forward_is_the_eager_forward_bit_for_bit():
    """geossl_amd.graphed.GraphedForward: SchNet.forward under no_grad as one HIP-graph replay per call (the evaluation
    loop of finetune_qm9.py:278-384; BASELINE config 2) - equal-sized molecules share a graph captured at first sight (its
    pass takes the layer loop), ragged size sequences are captured at their second sighting; outputs equal the eager
    forward's bit for bit, a new batch's atom types and positions reach the replay."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.graphed import GraphedForward
    from geossl_amd.synthetic import make_batch
    model = product_schnet(FULL, DEV)
    gf = GraphedForward(model)
    uni = [pg.Batch.from_numpy(make_batch(64, seed=30 + i), DEV) for i in range(3)]
    for bt in uni:
        with torch.no_grad():
            want = model(bt.x[:, 0], bt.positions, bt.batch)
        assert torch.equal(gf(bt), want)
    assert gf.captures == 1 and len(gf.graphs) == 1
    rag = pg.Batch.from_numpy(make_batch(0, seed=40, sizes=_ragged_sizes(32, 41)), DEV)
    with torch.no_grad():
        want = model(rag.x[:, 0], rag.positions, rag.batch)
    assert torch.equal(gf(rag), want) and gf.captures == 1          # first sighting: eager
    assert torch.equal(gf(rag), want) and gf.captures == 2          # second: captured
    assert torch.equal(gf(rag), want) and gf.captures == 2
    lat = GraphedForward(model, return_latent=True)
    out, h = lat(uni[0])
    with torch.no_grad():
        o2, h2 = model(uni[0].x[:, 0], uni[0].positions, uni[0].batch, return_latent=True)
    assert torch.equal(out, o2) and torch.equal(h, h2)


def test_painn_oversized_molecules_go_to_the_per_atom_kernels_alone(monkeypatch):
    """Molecule3D with hydrogens has molecules above the LDS rows of the molecule-staged interaction kernels (44 atoms
    for the matrix-pipe forward, 73 for the backward).  The batch keeps those kernels for the molecules that fit; the
    atoms of the others are covered by the per-atom kernel (geossl_painn_interaction_fwd_atoms / _bwd_atoms) - features
    and parameter gradients against oracle.nets.painn_forward, and equal to the all-per-atom path within the summation
    order of the matrix-pipe forward."""
    from geossl_amd import _lib, ops
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    from oracle import nets
    from test_oracle_golden import painn_params
    import geossl_amd.Geom3D.models.painn as pm
    sizes = np.array([18, 50, 7, 80, 30, 46, 2, 44, 76], dtype=np.int64)
    raw = make_batch(0, seed=55, sizes=sizes)
    raw["x"][::7, 0] = 0
    bt = _painn_batch(raw)
    pos2 = bt.positions + 0.2 * torch.randn(bt.positions.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    calls = []
    real = pm.call

    def counting(name, *a):
        calls.append(name)
        return real(name, *a)

    def run():
        cfg, model = _painn_modules()
        out, q = model(bt.x, pos2, bt.radius_edge_index, bt.batch, return_latent=True)
        ((out ** 2).sum() + 0.5 * (q ** 2).sum()).backward()
        return cfg, out.detach(), q.detach(), unique_named_grads(model)

    monkeypatch.setattr(pm, "call", counting)
    # default: the backward is split (molecule-staged up to 73 atoms + the per-atom kernel over the 76- and 80-atom
    # molecules); the forward keeps one form for the batch (the split forward measured slower: layout.painn_stage_caps)
    cfg, out0, q0, grads0 = run()
    assert calls.count("geossl_painn_interaction_fwd_atoms") == 0 and calls.count("geossl_painn_interaction_fwd_mma_dyn") == 0
    assert calls.count("geossl_painn_interaction_bwd_mol_skip") == 3 and calls.count("geossl_painn_interaction_bwd_atoms") == 3
    del calls[:]
    monkeypatch.setenv("GEOSSL_PAINN_MMA_CAP", "44")        # ... and with the forward split as well
    cfg, out, q, grads = run()
    monkeypatch.delenv("GEOSSL_PAINN_MMA_CAP")
    assert calls.count("geossl_painn_interaction_fwd_mma_dyn") == 3 and calls.count("geossl_painn_interaction_fwd_atoms") == 3
    assert calls.count("geossl_painn_interaction_bwd_mol_skip") == 3 and calls.count("geossl_painn_interaction_bwd_atoms") == 3
    assert rel_err(q, q0) < 5e-6 and rel_err(out, out0) < 5e-6
    for k in grads:
        assert rel_err(grads[k], grads0[k]) < 2e-5, k
    monkeypatch.setattr(pm, "call", real)
    monkeypatch.setenv("GEOSSL_PAINN_NO_SPLIT", "1")        # no split at all: per-atom kernels for everything at 80 atoms
    _, out_v, q_v, grads_v = run()
    monkeypatch.delenv("GEOSSL_PAINN_NO_SPLIT")
    assert rel_err(q, q_v) < 5e-6 and rel_err(out, out_v) < 5e-6
    for k in grads:
        assert rel_err(grads[k], grads_v[k]) < 2e-5, k
    P = painn_params(cfg)
    o_ref, q_ref = nets.painn_forward(P, t(raw["x"]), pos2.cpu(), bt.radius_edge_index.cpu(), t(raw["batch"]), 128, 3, 5.0,
                                      "add", return_latent=True)
    ((o_ref ** 2).sum() + 0.5 * (q_ref ** 2).sum()).backward()
    assert rel_err(q.cpu(), q_ref.detach()) < TOL_OUT and rel_err(out.cpu(), o_ref.detach()) < TOL_OUT
    for k in ("filter_net.weight", "filter_net.bias", "interactions.1.interatomic_context_net.1.weight",
              "mixing.2.mu_channel_mix.weight", "embedding.weight"):
        assert rel_err(grads[k].cpu(), P[k].grad) < TOL_GRAD, k


def test_painn_backward_with_equal_shares_of_the_edges_per_block(monkeypatch):
    """k_painn_interaction_bwd_mol on batches whose largest molecule is far above the average (Molecule3D with hydrogens):
    every block owns an equal share of the edge array - a contiguous atom range found in the incidence-list offsets,
    molecules at a boundary staged by both neighbours - instead of whole molecules in turn.  Against the round-4 form
    (GEOSSL_PAINN_BALANCE=0): the atom-row gradients do not depend on who computes them (bit-identical input gradient of the
    embedding rows, positions untouched), the filter-network gradient only in the order of the block partials; run twice,
    the balanced form gives the same bits; sizes incl. single atoms, a molecule above the LDS rows and runs of tiny
    molecules that several blocks' ranges skip entirely."""
    from geossl_amd.synthetic import make_batch, molecule_sizes
    rng = np.random.default_rng(5)
    sizes = np.concatenate([molecule_sizes(70, "C", rng), np.array([1, 1, 2, 80, 1, 72, 3], dtype=np.int64)])
    raw = make_batch(0, seed=56, sizes=sizes)
    bt = _painn_batch(raw)
    assert int(sizes.max()) * len(sizes) > 2 * int(sizes.sum())      # the rule that turns the balanced form on by itself

    def run():
        cfg, model = _painn_modules()
        out, q = model(bt.x, bt.positions, bt.radius_edge_index, bt.batch, return_latent=True)
        ((out ** 2).sum() + 0.5 * (q ** 2).sum()).backward()
        return unique_named_grads(model)

    auto = run()
    monkeypatch.setenv("GEOSSL_PAINN_BALANCE", "1")
    forced, again = run(), run()
    monkeypatch.setenv("GEOSSL_PAINN_BALANCE", "0")
    old = run()
    for k in old:
        assert torch.equal(forced[k], again[k]) and torch.equal(forced[k], auto[k]), k
        assert rel_err(forced[k], old[k]) < 5e-6, k
    # (the embedding gradient is a sum of atom-row gradients in atom order: no block partials in it)
    assert torch.equal(forced["embedding.weight"], old["embedding.weight"])


@pytest.mark.parametrize("fwd_split", [False, True])
def test_painn_bucket_with_oversized_molecules_replays_bit_for_bit(fwd_split, monkeypatch):
    """The same split inside a capacity bucket: the lists of oversized molecules' atoms are device data rewritten per step
    (two lists: above 44 atoms for the forward, above 73 for the backward).  Two set-C-like batches, one of them without
    any molecule above 73 atoms (an empty list): one capture, replays bit-identical to the eager launches on the bucket,
    losses equal to the plain eager step within fp32 summation order."""
    from geossl_amd import bucket as bk
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch, molecule_sizes
    if fwd_split:
        monkeypatch.setenv("GEOSSL_PAINN_MMA_CAP", "44")
    B = 16
    specs = [molecule_sizes(B, "C", np.random.default_rng(70 + i)) for i in range(2)]
    specs[0][2], specs[0][9] = 90, 47
    specs[1][:] = np.minimum(specs[1], 60)
    specs[1][4] = 58
    raws = [make_batch(B, seed=720 + i, sizes=s_) for i, s_ in enumerate(specs)]
    nzs = [draw_noise(r, seed=730 + i) for i, r in enumerate(raws)]
    bts = [_painn_batch(r) for r in raws]

    def trainer(use_graph):
        cfg, model = _painn_modules()
        return pg.DDMTrainer(model, product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4,
                             model_3d="painn", use_graph=use_graph)
    tr = trainer(True)
    losses, grads = [], []
    for bt, nz in zip(bts, nzs):
        losses.append(tr._graph_fwd_bwd(bt, {k: t(v, DEV) for k, v in nz.items()}).clone())
        grads.append(tr.flat.grad.clone())
    assert tr.step_graphs.captures == 1
    bkt = next(iter(tr._graphs.values()))["bucket"]
    assert bkt.max_n == 128 and bkt.big_caps == ((44, 73) if fwd_split else (73,)) and int(bkt.el.status) == 0
    te = trainer(False)
    eb = bk.Bucket(torch.device(DEV), B, bkt.caps(), "combination", max_n=bkt.max_n, kind="painn", E_cap=bkt.E_cap)
    f32 = dict(dtype=torch.float32, device=DEV)
    sn = {"pos_noise": torch.zeros(eb.N_cap, 3, **f32), "dist_noise_1": torch.zeros(eb.S_cap, 1, **f32),
          "dist_noise_2": torch.zeros(eb.S_cap, 1, **f32), "noise_level_1": torch.zeros(B, dtype=torch.long, device=DEV),
          "noise_level_2": torch.zeros(B, dtype=torch.long, device=DEV)}
    tp = trainer(False)
    for i, (bt, nz) in enumerate(zip(bts, nzs)):
        N, P, S, W = eb.fill(bt)
        sn["pos_noise"][:N].copy_(t(nz["pos_noise"], DEV))
        sn["dist_noise_1"][:S].copy_(t(nz["dist_noise_1"], DEV))
        sn["dist_noise_2"][:S].copy_(t(nz["dist_noise_2"], DEV))
        sn["noise_level_1"].copy_(t(nz["noise_level_1"], DEV))
        sn["noise_level_2"].copy_(t(nz["noise_level_2"], DEV))
        loss = te._fwd_bwd(eb.batch, sn)
        assert torch.equal(loss, losses[i]) and torch.equal(te.flat.grad, grads[i]), i
        plain = tp._fwd_bwd(bt, {k: t(v, DEV) for k, v in nz.items()})
        assert abs(float(plain) - float(losses[i])) <= 2e-6 * abs(float(plain)), i
        assert rel_err(tp.flat.grad, grads[i]) < 1e-5, i


def test_painn_bucket_reports_an_edge_that_leaves_its_molecule():
    """A radius_edge_index whose edges are not grouped by molecule (never produced by the reference's dataset) makes the
    eager path raise at once (layout.EdgeLayout validates); on the bucket path geossl_painn_edge_layout leaves such edges
    out and flags them in a device word that the fill reads without draining the stream - the ValueError surfaces a few
    steps late, not never."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    raw = make_batch(0, seed=81, sizes=_ragged_sizes(12, 81))
    bt = _painn_batch(raw)
    bad = bt.radius_edge_index.clone()
    bad[1, 7] = bt.positions.size(0) - 1          # the other end in the last molecule
    bt.radius_edge_index = bad
    cfg, model = _painn_modules()
    tr = pg.DDMTrainer(model, product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4,
                       model_3d="painn", use_graph=True)
    with pytest.raises(ValueError, match="grouped by molecule"):
        for _ in range(20):
            tr.step(bt, None)
            torch.cuda.synchronize()


@pytest.mark.parametrize("opt_name", ["adam", "sgd"])
def test_reference_loop_with_painn_on_a_shuffled_loader_captures_once(opt_name):
    """examples/pretrain_GeoSSL.py:248-260 with --model_3d painn on the product modules over the reference's own loader
    surface (MoleculeDataset3DRadius-style molecules: a per-molecule radius_edge_index of the clean geometry,
    datasets_3D_Radius.py:120; AtomTupleExtractor; DataLoaderAtomTuple(shuffle=True); batch.to(device)): every batch has
    its own molecules, sizes and edge list; do_DDM captures ONE pair of graphs and replays it (a PaiNN bucket).  The
    bucket's partial sums are cut at capacity block boundaries, so a step equals the eager step within fp32 summation order
    (1e-6 of the loss); with plain SGD the whole trajectory stays that close; under the reference's Adam (:343) the first
    steps divide every gradient component by its own magnitude - components that are rounding noise get updates of size
    lr - and the two trajectories separate like any two fp32 evaluations do (steps 1-3 compared tightly, the rest loosely)."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, Data, DataLoaderAtomTuple
    from geossl_amd.synthetic import make_batch
    from oracle.graph import radius_graph_np
    sizes = _ragged_sizes(64, 17)
    b = make_batch(64, seed=78, sizes=sizes)
    ext = AtomTupleExtractor(ratio=1, option="combination")
    off, dataset = 0, []
    for n in sizes.tolist():
        pos = b["positions"][off:off + n]
        d = Data(x=torch.from_numpy(b["x"][off:off + n]), positions=torch.from_numpy(pos),
                 radius_edge_index=torch.from_numpy(radius_graph_np(pos, 5.0)))
        dataset.append(ext(d))
        off += n
    out = {}
    for graph in (False, True):
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        cfg, model = _painn_modules()
        n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
        pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
        args = types.SimpleNamespace(model_3d="painn", GeoSSL_mu=0.0, GeoSSL_sigma=0.3, lr=5e-4, decay=0.0, step_graph=graph)
        group = [{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()}, {"params": n2.parameters()}]
        optimizer = (torch.optim.Adam(group, lr=args.lr, weight_decay=args.decay) if opt_name == "adam"
                     else torch.optim.SGD(group, lr=1e-7))   # (the filler weights have large gradients)
        gen = torch.Generator()
        gen.manual_seed(1)
        loader = DataLoaderAtomTuple(dataset, batch_size=16, shuffle=True, generator=gen)
        losses, edges = [], set()
        try:
            for epoch in range(2):
                for batch in loader:
                    batch = batch.to(DEV)
                    edges.add(int(batch.radius_edge_index.size(1)))
                    loss, acc = pg.do_DDM(args, batch, model, criterion=None, mu=args.GeoSSL_mu, sigma=args.GeoSSL_sigma)
                    losses.append(loss.detach().item())
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
        finally:
            pg.NCSN_model_01 = pg.NCSN_model_02 = None
        assert len(losses) == 8 and len(edges) > 4
        eng = model.__dict__.get("_geossl_autograd_step")
        caps = sum(sg.captures for sg in eng.graphs.values()) if eng is not None else 0
        out[graph] = (losses, torch.cat([p.detach().reshape(-1) for m in (model, n1, n2) for p in m.parameters()]).cpu(), caps)
    assert out[False][2] == 0 and 1 <= out[True][2] <= 2, out[True][2]
    for k, (a, c) in enumerate(zip(out[True][0], out[False][0])):
        tol = 5e-6 if (opt_name == "sgd" or k < 3) else 5e-2
        assert abs(a - c) <= tol * abs(c), (k, out[True][0], out[False][0])
    assert rel_err(out[True][1], out[False][1]) < (1e-4 if opt_name == "sgd" else 2e-2)


# ------------------------------------------------------------------------------- every product at 24 bits (a switch)
def test_weight_gradient_gemm_on_three_bf16_pieces_vs_fp64(monkeypatch):
    """geossl_linear_wgrad under GEOSSL_ARITH_24BIT (k_wgrad_split<..., 3>: three bf16 pieces, six MFMAs per product, no
    operand scales) against fp64, next to the default two-fp16-piece form: both within 2e-6 of the tensor scale, inputs of
    ordinary size, x 1e-9 and x 1e+5 (bf16 pieces carry fp32's exponent range)."""
    from geossl_amd import ops
    from conftest import max_abs_rel
    g = torch.Generator(device=DEV).manual_seed(11)
    R = 5000
    for sa, sb in ((1.0, 1.0), (1e-9, 1e5)):
        A = torch.randn(R, 128, device=DEV, generator=g) * sa
        Bm = torch.randn(R, 128, device=DEV, generator=g) * sb
        ref = A.double().t() @ Bm.double()
        refb = A.double().sum(0)
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("GEOSSL_ARITH_24BIT", env)
            else:
                monkeypatch.delenv("GEOSSL_ARITH_24BIT", raising=False)
            dW, db = torch.empty(128, 128, device=DEV), torch.empty(128, device=DEV)
            ops.linear_wgrad([(A, Bm, dW, db)], R, 128, 128)
            assert max_abs_rel(dW.cpu(), ref.cpu()) < 2e-6, (sa, sb, env)
            assert max_abs_rel(db.cpu(), refb.cpu()) < 2e-6
    monkeypatch.delenv("GEOSSL_ARITH_24BIT", raising=False)


def test_step_with_every_product_at_24_bits_vs_oracle(monkeypatch):
    """GEOSSL_ARITH_24BIT: filter network, atom-row layers, weight gradients and both heads with three bf16 pieces per
    operand (six MFMAs, 24-bit products - fp32's own product width) - the DDM step against oracle.nets.do_ddm_schnet at the
    tolerances of the default path (loss 1e-5, gradients 1e-4), and within 2e-6 of the default path's loss."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle import nets
    b = make_batch(48, seed=12, mode="B")
    nz = draw_noise(b, seed=13)
    noise = {k: t(v, DEV) for k, v in nz.items()}
    out = {}
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("GEOSSL_ARITH_24BIT", env)
        tr = _trainer(FULL, use_graph=False)
        loss = tr._fwd_bwd(pg.Batch.from_numpy(b, DEV), noise)
        out[env] = (float(loss), unique_named_grads(tr.model), tr)
    monkeypatch.delenv("GEOSSL_ARITH_24BIT")
    assert abs(out["1"][0] - out[None][0]) <= 2e-6 * abs(out[None][0])
    Pm, P1, P2 = schnet_oracle_params(FULL), ncsn_oracle_params(128, 50), ncsn_oracle_params(128, 50, 0.9)
    ref = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                             t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
                             t(nz["dist_noise_2"]), 5.0, 6, 2, "mean")
    ref.backward()
    assert abs(out["1"][0] - float(ref.detach())) <= TOL_OUT * abs(float(ref.detach()))
    for k in ("lin2.weight", "interactions.0.mlp.0.weight", "interactions.3.mlp.2.weight", "interactions.5.conv.lin1.weight",
              "interactions.2.conv.lin2.weight", "interactions.4.lin.weight", "embedding.weight"):
        assert rel_err(out["1"][1][k].cpu(), Pm[k].grad) < TOL_GRAD, k
