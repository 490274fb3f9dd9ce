"""GPU tests added in round 3: step graphs found by a content-derived fingerprint, the reference's own loop on the
graph path of do_DDM, bench.py's N > 1 branch on one GPU."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import rel_err
from helpers import product_ncsn, product_schnet, t, unique_named_grads

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


def _reference_loop(graph, steps, batches, cfg=SMALL, seed=11):
    """examples/pretrain_GeoSSL.py:248-260 + :332-343 on the product modules; returns losses and final parameters."""
    from geossl_amd import pretrain_GeoSSL as pg
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    model = product_schnet(cfg, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9)
    pg.NCSN_model_01, pg.NCSN_model_02 = n1, n2
    args = types.SimpleNamespace(model_3d="schnet", GeoSSL_mu=0.0, GeoSSL_sigma=0.3, lr=5e-4, decay=0.0,
                                 step_graph=graph)
    group = [{"params": model.parameters(), "lr": args.lr}, {"params": n1.parameters()}, {"params": n2.parameters()}]
    optimizer = torch.optim.Adam(group, lr=args.lr, weight_decay=args.decay)
    losses = []
    try:
        for step in range(steps):
            batch = batches[step % len(batches)]
            loss, acc = pg.do_DDM(args, batch, model, criterion=None, mu=args.GeoSSL_mu, sigma=args.GeoSSL_sigma)
            losses.append(loss.detach().item())
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
    finally:
        pg.NCSN_model_01 = pg.NCSN_model_02 = None
    eng = model.__dict__.get("_geossl_autograd_step")
    captures = sum(sg.captures for sg in eng.graphs.values()) if eng is not None else 0
    params = torch.cat([p.detach().reshape(-1) for m in (model, n1, n2) for p in m.parameters()]).cpu()
    return losses, params, captures


def test_reference_loop_on_the_graph_path_is_the_eager_loop_bit_for_bit():
    """The reference's loop body, unchanged, with do_DDM replaying a captured forward+backward from the second sighting
    of a batch structure on: same seeds -> the same five draws per step (host position noise, device head noise), the same
    losses and, through stock torch.optim.Adam over the reference's three groups, the same parameters as eager launches.
    Fresh tensor objects every step (a loader's collation): the graph is found by the molecule sizes."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    out = {}
    for graph in (False, True):
        batches = [pg.Batch.from_numpy(make_batch(48, seed=40 + i), DEV) for i in range(5)]
        out[graph] = _reference_loop(graph, 5, batches)
    assert out[True][2] == 1 and out[False][2] == 0, "one capture, at the second step"
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    assert len(set(out[True][0])) == 5


def test_graph_path_keeps_autograds_contract():
    """The loss do_DDM returns from a replay is an ordinary differentiable scalar: torch.autograd.grad over a subset of
    parameters, a scaled loss, gradient accumulation over two steps in flight and parameter hooks all see what the
    eager path gives them."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(32, seed=7)
    batch = pg.Batch.from_numpy(b, DEV)
    nz = [{k: t(v, DEV) for k, v in draw_noise(b, seed=70 + i).items()} for i in range(2)]
    torch.manual_seed(3)
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
    args = pg.Args("schnet")
    params = [p for m in (model,) + heads for p in m.parameters() if p.requires_grad]

    def grads(graph):
        for p in params:
            p.grad = None
        l0, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[0], graph=graph)
        l1, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[1], graph=graph)
        (0.5 * l0 + 2.0 * l1).backward()      # two steps in flight, scaled
        return float(l0), float(l1), [None if p.grad is None else p.grad.clone() for p in params]

    pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[0], graph=True)   # first sighting (eager)
    seen = []
    hook = params[3].register_hook(lambda g: seen.append(g.clone()))
    e0, e1, ge = grads(False)
    g0, g1, gg = grads(True)
    hook.remove()
    assert (e0, e1) == (g0, g1)
    assert [x is None for x in ge] == [x is None for x in gg]
    for a, c in zip(ge, gg):
        if a is not None:
            assert float((a - c).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-30   # 0.5 g0 + 2 g1: one rounding apart
    assert len(seen) == 2 and torch.allclose(seen[0], seen[1], rtol=1e-5, atol=0)
    sub = [params[0], params[5]]
    l, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[0], graph=True)
    ga = torch.autograd.grad(l, sub)
    l, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[0], graph=False)
    gb = torch.autograd.grad(l, sub)
    assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    with torch.no_grad():                                  # no gradients wanted: plain forward, no replay
        l2, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz[0], graph=True)
    assert not l2.requires_grad and float(l2) == e0


def test_trainer_graphs_follow_the_batch_structure_not_a_callers_key():
    """two ragged batches with the same N and S but different molecule sizes, stepped under ONE
    caller key, must not replay each other's index structures; batches with equal sizes share a graph whatever tensor
    objects they arrive as."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    specs = [[10, 20, 14], [20, 10, 14], [10, 20, 14], [14, 20, 10]]
    raw = [make_batch(3, seed=80 + i, sizes=s) for i, s in enumerate(specs)]
    assert len({b["x"].shape[0] for b in raw}) == 1 and len({b["super_edge_index"].shape[1] for b in raw}) == 1
    losses, n_graphs = {}, None
    for use_graph in (False, True):
        tr = pg.DDMTrainer(product_schnet(SMALL, DEV), product_ncsn(128, 50, 2, DEV),
                           product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4, use_graph=use_graph,
                           graph_mode="structure")  # (the default mode serves ragged batches from ONE bucket graph)
        out = []
        for step in range(8):
            b = raw[step % 4]
            noise = {k: t(v, DEV) for k, v in draw_noise(b, seed=300 + step).items()}
            out.append(float(tr.step(pg.Batch.from_numpy(b, DEV), noise, structure_key="same-key-for-everything")))
        losses[use_graph] = out
        if use_graph:
            assert tr.use_graph, "capture fell back to eager"
            n_graphs = len(tr._graphs)
    assert losses[True] == losses[False], losses
    assert n_graphs == 3      # [10,20,14] twice -> one graph


def test_trainer_graph_cache_is_lru_and_losses_survive_eviction():
    """beyond max_graphs the least recently used graph is dropped (with a warning), a loss returned
    earlier stays valid (it is a copy, not the graph's static scalar), and a structure that comes back is captured again
    with the right result."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    raw = [make_batch(3, seed=90 + i, sizes=s) for i, s in enumerate([[6, 9, 12], [9, 6, 12], [12, 9, 6]])]

    def run(use_graph):
        tr = pg.DDMTrainer(product_schnet(SMALL, DEV), product_ncsn(128, 50, 2, DEV),
                           product_ncsn(128, 50, 2, DEV, scale=0.9), lr=5e-4, use_graph=use_graph, max_graphs=2,
                           graph_mode="structure")
        kept = []
        for step in range(7):
            b = raw[step % 3]
            noise = {k: t(v, DEV) for k, v in draw_noise(b, seed=500 + step).items()}
            kept.append(tr.step(pg.Batch.from_numpy(b, DEV), noise))
        return [float(x) for x in kept], tr

    eager, _ = run(False)
    with pytest.warns(UserWarning, match="least recently used"):
        replayed, tr = run(True)
    assert replayed == eager
    assert len(tr._graphs) == 2 and tr.step_graphs.captures == 7


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it - the command shape the driver uses: the parent (which never
    touches the GPU) starts two fresh ranks itself; they run bench.py's N > 1 branch (init_distributed, per-rank
    molecules and noise streams, barrier + max-over-ranks timing, all-reduce + Adam, the JSON line of rank 0) on the one
    GPU of the box over gloo."""
    env = dict(os.environ, GEOSSL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", GEOSSL_BENCH_RANK_LOSS=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--mols", "256", "--dataset-mols", "8192"]
    with open(tmp_path / "out.log", "w") as fo, open(tmp_path / "err.log", "w") as fe:
        p = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, cwd=REPO)
        try:
            code = p.wait(timeout=300)
        except subprocess.TimeoutExpired:
            code = None
            p.kill()
            p.wait()
    out_text, err_text = open(tmp_path / "out.log").read(), open(tmp_path / "err.log").read()
    assert code == 0, out_text[-1500:] + "\n" + err_text[-1500:]
    lines = [ln for ln in out_text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not any(ln.startswith("{") for ln in err_text.splitlines())
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["parallelism"] == "dp2"
    assert out["config"]["backend"] == "gloo" and out["config"]["world_size_initialised"] == 2
    assert np.isfinite(out["value"]) and out["value"] > 0 and out["cpu_baseline"] is None
    assert abs(out["value"] - 2 * 256 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-4 * out["value"]   # (the line's 6 digits)
    # what an N > 1 run needs to be read: the collective alone, both ranks' own step times, the loader's share of a step
    mg = out["multi_gpu"]
    assert mg["allreduce_ms"] > 0 and mg["allreduce_bytes"] == 4 * 520324
    assert 0 < mg["rank_ms_per_step"]["min"] <= mg["rank_ms_per_step"]["max"] <= out["ms_per_step"] * 1.001
    assert mg["input_ms"]["host"] > 0 and mg["input_ms"]["device"] > 0
    l0, l1 = (float(open(tmp_path / ("loss_rank%d.txt" % r)).read()) for r in range(2))
    assert np.isfinite(l0) and np.isfinite(l1) and l0 != l1      # each rank has its own molecules and noise stream


def test_bench_gpus_flag_must_match_the_launched_world(tmp_path):
    """--gpus N under a launcher that started another number of ranks fails loudly instead of printing n_gpus: 1."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, cwd=REPO, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


# --------------------------------------------------------------------------- PaiNN interaction on the matrix pipe
def _painn_edge_case(sizes, seed, R=20):
    from geossl_amd import _lib, ops
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.layout import MolLayout, get_edge_layout
    from geossl_amd.synthetic import make_batch
    b = make_batch(len(sizes), seed=seed, sizes=sizes)
    bt = pg.Batch.from_numpy(b, DEV)
    lay = MolLayout(bt.batch, len(sizes), sizes=list(sizes))
    rei = ops.radius_graph(bt.positions, 5.0, bt.batch)
    el = get_edge_layout(bt.batch, rei, lay.B)
    E, N, Fd = el.E, bt.positions.size(0), 128
    f32 = dict(dtype=torch.float32, device=DEV)
    dirv, fcut, phi = torch.empty(max(E, 1), 3, **f32), torch.empty(max(E, 1), **f32), torch.empty(max(E, 1), R, **f32)
    offsets = torch.linspace(0.0, 5.0, R, device=DEV)
    widths = torch.abs(offsets[1] - offsets[0]) * torch.ones_like(offsets)
    _lib.call("geossl_painn_edge_geom", bt.positions.data_ptr(), el.idx_i.data_ptr(), el.idx_j.data_ptr(), E, 5.0,
              offsets.data_ptr(), widths.data_ptr(), R, dirv.data_ptr(), fcut.data_ptr(), phi.data_ptr(), _lib.stream())
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV, generator=gen) * scale).contiguous()
    return dict(lay=lay, el=el, N=N, E=E, R=R, dirv=dirv, fcut=fcut, phi=phi, Wf=rnd(3 * Fd, R, scale=0.3),
                bf=rnd(3 * Fd, scale=0.2), q=rnd(N, Fd), mu=rnd(N, 3, Fd), xc=rnd(N, 3 * Fd))


@pytest.mark.parametrize("sizes", [[18] * 24, [2, 26, 1, 7, 18, 1, 12, 25, 3, 20, 9, 33], [1, 1, 2]], ids=["setA", "ragged", "tiny"])
@pytest.mark.parametrize("R", [20, 8])
def test_painn_interaction_forward_on_the_matrix_pipe_matches_the_vector_kernel(sizes, R):
    """geossl_painn_interaction_fwd_mma (filter as a GEMM per tile of 32 edge rows, groups of four rows per target atom)
    against geossl_painn_interaction_fwd_mol on the same inputs: q_out / mu_out within 2e-6 of the tensor scale (another
    summation order, two-piece fp16 products), atoms without edges and single-atom molecules included; bit-reproducible."""
    from geossl_amd import _lib
    c = _painn_edge_case(sizes, seed=3, R=R)
    lay, el, N, Fd = c["lay"], c["el"], c["N"], 128
    st = _lib.stream()
    inc_ptr, inc_idx = el.inc["i"]
    q_ref, mu_ref = torch.empty_like(c["q"]), torch.empty_like(c["mu"])
    _lib.call("geossl_painn_interaction_fwd_mol", c["q"].data_ptr(), c["mu"].data_ptr(), c["xc"].data_ptr(),
              el.idx_j.data_ptr(), inc_ptr.data_ptr(), inc_idx.data_ptr(), c["phi"].data_ptr(), c["fcut"].data_ptr(),
              c["dirv"].data_ptr(), c["Wf"].data_ptr(), c["bf"].data_ptr(), lay.mol_ptr.data_ptr(), lay.B, lay.max_n, N, Fd,
              R, q_ref.data_ptr(), mu_ref.data_ptr(), st)
    row_edge, grp_atom, grp_ptr, mol_grp = el.groups("i", lay.mol_ptr)
    outs = []
    for _ in range(2):
        q_out = torch.full_like(c["q"], float("nan"))
        mu_out = torch.full_like(c["mu"], float("nan"))
        _lib.call("geossl_painn_interaction_fwd_mma", c["q"].data_ptr(), c["mu"].data_ptr(), c["xc"].data_ptr(),
                  el.idx_j.data_ptr(), row_edge.data_ptr(), grp_atom.data_ptr(), mol_grp.data_ptr(), c["phi"].data_ptr(),
                  c["fcut"].data_ptr(), c["dirv"].data_ptr(), c["Wf"].data_ptr(), c["bf"].data_ptr(), lay.mol_ptr.data_ptr(),
                  lay.B, lay.max_n, N, Fd, R, q_out.data_ptr(), mu_out.data_ptr(), st)
        outs.append((q_out, mu_out))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    scale = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert scale(outs[0][0], q_ref) < 2e-6, scale(outs[0][0], q_ref)
    assert scale(outs[0][1], mu_ref) < 2e-6, scale(outs[0][1], mu_ref)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_head_gradients_meet_in_one_buffer_without_a_concatenation():
    """The two heads write their gradients at h into the two halves of one buffer (GradSlot on the split views), so the
    backward of the view split hands that buffer on instead of concatenating (12 us of a 2.8 ms step).  Same values as
    the concatenation; a second backward through a retained graph gets a buffer of its own."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(24, seed=11)
    batch = pg.Batch.from_numpy(b, DEV)
    nz = {k: t(v, DEV) for k, v in draw_noise(b, seed=12).items()}
    torch.manual_seed(5)
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 2, DEV, scale=0.9))
    args = pg.Args("schnet")
    params = [p for m in (model,) + heads for p in m.parameters() if p.requires_grad]

    def run(with_slots):
        for p in params:
            p.grad = None
        orig = pg.split_views
        if not with_slots:
            pg.split_views = lambda h, n: pg._SplitViews.apply(h, n)
        try:
            loss, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=nz, graph=False)
            loss.backward(retain_graph=with_slots)
            first = [p.grad.clone() for p in params]
            if with_slots:  # the retained graph once more: gradients double, nothing of the first pass is overwritten
                loss.backward()
                assert all(torch.allclose(p.grad, 2 * g, rtol=1e-6, atol=0) for p, g in zip(params, first))
        finally:
            pg.split_views = orig
        return float(loss), first

    stats = pg._SplitViews.STATS
    c0, a0 = stats["cat"], stats["adjacent"]
    l_cat, g_cat = run(False)
    assert stats["cat"] == c0 + 1 and stats["adjacent"] == a0
    l_adj, g_adj = run(True)
    assert stats["adjacent"] == a0 + 2 and stats["cat"] == c0 + 1
    assert l_cat == l_adj
    assert all(torch.equal(x, y) for x, y in zip(g_cat, g_adj))


def test_ddm_views_is_perturb_concatenation_and_both_distance_sets_bit_for_bit():
    """geossl_ddm_views (one launch) against the launches it replaces: geossl_axpy, the concatenation and two
    geossl_pair_distance calls (pretrain_GeoSSL.py:68-74,199-205)."""
    from geossl_amd import ops
    g = torch.Generator().manual_seed(3)
    N, S = 1000, 7777
    pos = (torch.randn(N, 3, generator=g) * 3).to(DEV)
    noise = (torch.randn(N, 3, generator=g) * 0.3).to(DEV)
    sei = torch.randint(0, N, (2, S), generator=g).to(DEV)
    pos2, d01, d02 = ops.ddm_views(pos, noise, sei[0], sei[1])
    p2 = ops.add_scaled(pos, noise, 1.0)
    assert torch.equal(pos2, torch.cat([pos, p2]))
    assert torch.equal(d01, ops.pair_distance(pos, sei[0], sei[1]))
    assert torch.equal(d02, ops.pair_distance(p2, sei[0], sei[1]))


@pytest.mark.parametrize("rows", [37, 5000])
def test_chain_silu_epilogues_match_separate_silu_launches(rows):
    """Dense(F, F, silu) -> Dense(F, 3F) as one five-stage launch with GEOSSL_EPI_SILU (pre-activation and silu of it both
    stored), and the backward form (sum of three passes) * silu'(u) -> next Dense with GEOSSL_EPI_MUL_DSILU, against the
    same layers with geossl_silu_fwd / geossl_silu_bwd launched between the chains (painn_utils.py:27-35)."""
    from geossl_amd import ops, _lib
    from geossl_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(rows)
    F = 128
    W = [(torch.randn(F, F, generator=g) / 11).to(DEV) for _ in range(5)]
    b = [(torch.randn(F, generator=g) * 0.1).to(DEV) for _ in range(4)]
    x = torch.randn(rows, F, generator=g).to(DEV)
    x2 = torch.randn(rows, F, generator=g).to(DEV)
    img = ops.prepare_chain(W, transB=True)
    # forward: [W0 x + b0 (no store), + W1 x2 -> u, silu(u) -> s, then three fan-out layers on s]
    u, s = torch.empty(rows, F, device=DEV), torch.empty(rows, F, device=DEV)
    outs = [torch.empty(rows, F, device=DEV) for _ in range(3)]
    ops.linear_chain(x, [dict(image=img[0], bias=b[0], store=False),
                         dict(image=img[1], x=x2, add_prev=True, out=u, out_act=s, flags=_lib.EPI_SILU)] +
                     [dict(image=img[2 + c], bias=b[1 + c], out=outs[c], same_input=(c > 0)) for c in range(3)])
    u_ref = ops.linear_chain(x, [dict(image=img[0], bias=b[0], store=False), dict(image=img[1], x=x2, add_prev=True)])[1]
    s_ref = torch.empty_like(u_ref)
    call("geossl_silu_fwd", ptr(u_ref), u_ref.numel(), ptr(s_ref), stream())
    outs_ref = ops.linear_chain(s_ref, [dict(image=img[2 + c], bias=b[1 + c], same_input=(c > 0)) for c in range(3)])
    assert torch.equal(u, u_ref)
    assert rel_err(s, s_ref) < 1e-6
    for a, r in zip(outs, outs_ref):
        assert rel_err(a, r) < 2e-6
    # against fp64
    u64 = x.double() @ W[0].double().T + b[0].double() + x2.double() @ W[1].double().T
    s64 = u64 * torch.sigmoid(u64)
    assert rel_err(outs[1], (s64 @ W[3].double().T + b[2].double()).float()) < 3e-6
    # backward form: (x W0^T + x2 W1^T) * silu'(u) -> d, then d W2^T
    d = torch.empty(rows, F, device=DEV)
    y = ops.linear_chain(x, [dict(image=img[0], store=False),
                             dict(image=img[1], x=x2, add_prev=True, tprev=u, out=d, flags=_lib.EPI_MUL_DSILU),
                             dict(image=img[2])])[2]
    pre = ops.linear_chain(x, [dict(image=img[0], store=False), dict(image=img[1], x=x2, add_prev=True)])[1]
    d_ref = torch.empty_like(pre)
    call("geossl_silu_bwd", ptr(u), ptr(pre), pre.numel(), ptr(d_ref), stream())
    y_ref = ops.linear_chain(d_ref, [dict(image=img[2])])[0]
    assert rel_err(d, d_ref) < 1e-6 and rel_err(y, y_ref) < 2e-6
    # a strided weight block (column half of a [F, 2F] matrix) gives the image of its contiguous copy
    wide = torch.randn(F, 2 * F, generator=g).to(DEV)
    for tb in (True, False):
        a = ops.prepare_chain([wide[:, F:]], transB=tb)[0]
        c = ops.prepare_chain([wide[:, F:].contiguous()], transB=tb)[0]
        used = 4 * 8 * 2 * 64 * 4 + 4  # two fp16 pieces of every fragment + the four block exponents (the rest is unused)
        assert torch.equal(a[:used], c[:used])


@pytest.mark.parametrize("sizes", [[18] * 40, [20] * 9, [7] * 600, [5, 9, 18, 20, 2, 1, 13, 17, 20, 11, 3, 16] * 3,
                                   [33, 1, 27, 30, 2, 22, 24, 26, 28, 8, 12, 31] * 20, [18, 25, 9, 33] * 300])
def test_layer_loop_is_the_separate_launches_bit_for_bit(sizes, monkeypatch):
    """geossl_schnet_layer_loop (every chain and aggregation of the backbone between the filter network and the heads as
    ONE launch per pass, a block carrying its molecules through all of them) against the 26 separate launches: atom
    features, and every parameter gradient, bit for bit - uniform batches of 18-, 20- and 7-atom molecules (the last one
    with four molecules per block); ragged batches of up to 256 molecules take geossl_schnet_layer_loop_ragged (one
    molecule per block, aggregated by its four waves; 1 .. 33 atoms); a larger ragged batch keeps the separate launches."""
    from geossl_amd import _lib
    from geossl_amd.Geom3D.dataloaders.dataloaders_AtomTuple import BatchAtomTuple
    g = torch.Generator().manual_seed(len(sizes))
    N = sum(sizes)
    x = torch.randint(0, 9, (N, 1), generator=g).to(DEV)
    pos = (torch.randn(N, 3, generator=g) * 2).to(DEV)
    data = BatchAtomTuple.from_sizes(x, pos, sizes, option="combination")
    torch.manual_seed(1)
    model = product_schnet(FULL, DEV)
    w = torch.randn(N, 128, generator=g).to(DEV)

    def run(loop):
        monkeypatch.setenv("GEOSSL_LAYER_LOOP", "1" if loop else "0")  # (by default: only while a graph is captured)
        for p in model.parameters():
            p.grad = None
        _lib.CALLS = 0
        _, h = model(x[:, 0], pos, data.batch, return_latent=True)
        (h * w).sum().backward()
        calls, _lib.CALLS = _lib.CALLS, None
        return h.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None], calls

    h0, g0, c0 = run(False)
    h1, g1, c1 = run(True)
    assert torch.equal(h0, h1)
    assert len(g0) == len(g1) and all(torch.equal(a, b) for a, b in zip(g0, g1))
    if len(set(sizes)) == 1 or len(sizes) <= 256:
        assert c1 == c0 - 2 * (2 * 6 + 2) + 2      # 14 launches per pass became one
    else:
        assert c1 == c0                              # large ragged batches: the separate launches


def test_two_heads_in_the_same_launches_are_the_two_single_head_calls(monkeypatch):
    """NCSN.ddm_heads_loss (both heads of pretrain_GeoSSL.py:207-210 as one autograd node on geossl_ddm_loss_fwd2 /
    _bwd_fused2) against NCSN_model_01(...) + NCSN_model_02(...): the same arithmetic per row; the pair gives each head
    half of the chip's blocks, so block partial sums are cut elsewhere - loss within 1e-6, every gradient within 2e-6 of
    the tensor's scale.  With the heads' own draws (same generator calls in the same order) as well as injected noise."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import draw_noise, make_batch
    b = make_batch(40, seed=21, mode="B")
    batch = pg.Batch.from_numpy(b, DEV)
    torch.manual_seed(9)
    model = product_schnet(SMALL, DEV)
    heads = (product_ncsn(128, 50, 2, DEV), product_ncsn(128, 50, 5, DEV, scale=0.9))
    args = pg.Args("schnet")
    params = [p for m in (model,) + heads for p in m.parameters() if p.requires_grad]
    full = {k: t(v, DEV) for k, v in draw_noise(b, seed=22).items()}

    def run(separate, noise):
        if separate:
            monkeypatch.setenv("GEOSSL_NCSN_SEPARATE_HEADS", "1")
        else:
            monkeypatch.delenv("GEOSSL_NCSN_SEPARATE_HEADS", raising=False)
        for p in params:
            p.grad = None
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        loss, _ = pg.do_DDM(args, batch, model, NCSN_models=heads, noise=noise, graph=False)
        loss.backward()
        return float(loss), [p.grad.clone() for p in params if p.grad is not None]

    for noise in (full, {"pos_noise": full["pos_noise"]}):   # (second: the heads draw their own levels and distance noise)
        l0, g0 = run(True, noise)
        l1, g1 = run(False, noise)
        assert np.isfinite(l0) and abs(l0 - l1) <= 1e-6 * abs(l0)
        assert len(g0) == len(g1)
        for x, y in zip(g0, g1):
            assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()) + 1e-30


def test_small_entry_points_of_round_3_through_the_c_abi():
    """geossl_copy2 (two copies, one launch), geossl_loss_reduce_partials (the mean from the row pass's block partials)
    against geossl_loss_reduce over the per-edge losses, and geossl_schnet_layer_loop's refusal of shapes it does not
    take (the caller then launches the operations one by one)."""
    import ctypes as C
    from geossl_amd import _lib
    from geossl_amd._lib import ptr, stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    a = torch.randint(0, 100, (1000, 2), generator=g).to(DEV)
    b = torch.randn(1000, 3, generator=g).to(DEV)
    da, db = torch.zeros_like(a), torch.zeros_like(b)
    assert lib.geossl_copy2(ptr(da), ptr(a), a.numel() * 8, ptr(db), ptr(b), b.numel() * 4, stream()) == 0
    assert torch.equal(da, a) and torch.equal(db, b)
    assert lib.geossl_copy2(ptr(da), ptr(a), 6, ptr(db), ptr(b), 4, stream()) != 0           # not a multiple of 4
    # loss from block partials == loss from the per-edge values (same total, other partial sums: 1e-6)
    S = 50_000
    le = torch.rand(S, generator=g).to(DEV)
    stats = torch.tensor([40, 0], dtype=torch.int64, device=DEV)
    part = torch.zeros(256, device=DEV)
    part[:200] = le.view(200, 250).sum(1)
    l1, l2, ws = torch.zeros((), device=DEV), torch.zeros((), device=DEV), torch.empty(256, device=DEV)
    assert lib.geossl_loss_reduce(ptr(le), S, ptr(stats), 0.5, ptr(l1), ptr(ws), 0, stream()) == 0
    assert lib.geossl_loss_reduce_partials(ptr(part), ptr(stats), 0.5, ptr(l2), 0, stream()) == 0
    ref = float(le.double().sum() / 40 * 0.5)
    assert abs(float(l1) - ref) <= 1e-6 * ref and abs(float(l2) - ref) <= 1e-6 * ref
    l3 = torch.zeros((), device=DEV)
    assert lib.geossl_loss_reduce_partials2(ptr(part), ptr(part), ptr(stats), 0.5, 0.25, ptr(l3), stream()) == 0
    assert abs(float(l3) - 1.5 * ref) <= 1e-6 * ref
    # the layer loop refuses ragged batches, other widths, too many operations
    ops_ = (_lib.LoopOp * 1)()
    plan = torch.tensor([[0, 18, 0, 1]], dtype=torch.int32, device=DEV)
    x = torch.zeros(18, 128, device=DEV)
    ops_[0].kind, ops_[0].X, ops_[0].Wf, ops_[0].out = 1, ptr(x), ptr(x), ptr(x)
    args = (C.byref(ops_), 1, ptr(plan), 1, ptr(plan), ptr(plan), ptr(plan))
    assert lib.geossl_schnet_layer_loop(*args, 18, 0, 18, 128, 0, stream()) != 0     # not uniform
    assert lib.geossl_schnet_layer_loop(*args, 18, 1, 18, 64, 0, stream()) != 0      # F = 64
    assert lib.geossl_schnet_layer_loop(*args, 21, 1, 18, 128, 0, stream()) != 0     # molecules above 20 atoms
    assert lib.geossl_schnet_layer_loop(C.byref(ops_), 15, ptr(plan), 1, ptr(plan), ptr(plan), ptr(plan), 18, 1, 18, 128, 0,
                                        stream()) != 0                               # more than 14 operations
    torch.cuda.synchronize()


@pytest.mark.parametrize("power,upstream", [(2.0, 1.0), (10.0, 1.0), (0.05, 1.0), (2.0, 1e-12), (2.0, 1e6), (10.0, 1e-9)])
def test_two_piece_ncsn_backward_follows_the_scale_of_the_row_gradient(power, upstream, monkeypatch):
    """ncsn_bwd.hip on two fp16 pieces: the upstream row gradient spans sigma^power between molecules (15 orders of
    magnitude at power 10 over the 0.01 .. 10 noise ladder) and whatever the caller's loss scaling adds - the running
    exponent published with the row scalars, the weight-only bound of dz1 and the rescaled accumulators have to carry
    that.  One pass against the two-pass form on three bf16 pieces (ncsn_rows.hip + the column GEMMs), both heads'
    pair launches against the single-head launches, tensor-level 3e-6; zero upstream gradient gives exact zeros."""
    from geossl_amd.Geom3D.dataloaders.dataloaders_AtomTuple import BatchAtomTuple
    gen = torch.Generator().manual_seed(int(power * 100) + 3)
    sizes = torch.randint(2, 27, (300,), generator=gen).tolist()
    N = sum(sizes)
    x = torch.randint(0, 9, (N, 1), generator=gen)
    pos = torch.randn(N, 3, generator=gen)
    data = BatchAtomTuple.from_sizes(x.to(DEV), pos.to(DEV), sizes, option="combination")
    sei = data.super_edge_index
    S = sei.size(1)
    h = (torch.randn(N, 128, generator=gen) * 0.5).to(DEV)
    dist = (pos.to(DEV)[sei[0]] - pos.to(DEV)[sei[1]]).norm(dim=-1, keepdim=True)
    nl = torch.randint(0, 50, (len(sizes),), generator=gen).to(DEV)
    dn = torch.randn(S, 1, generator=gen).to(DEV)
    head = product_ncsn(128, 50, power, DEV)

    def run(split, c):
        if split:
            monkeypatch.setenv("GEOSSL_NCSN_SPLIT_BWD", "1")
        else:
            monkeypatch.delenv("GEOSSL_NCSN_SPLIT_BWD", raising=False)
        for p in head.parameters():
            p.grad = None
        hh = h.clone().requires_grad_()
        loss = head(data, hh, dist, noise_level=nl, distance_noise=dn)
        (loss * c).backward()
        out = {k: v.detach().clone() for k, v in unique_named_grads(head).items()}
        out["h"] = hh.grad.clone()
        return out

    one, two = run(False, upstream), run(True, upstream)
    errs = {k: rel_err(one[k], two[k]) for k in one}
    print("rel errs", power, upstream, {k: "%.1e" % v for k, v in errs.items()})
    # (power 0.05 weighs all noise levels alike: the 1 -> F -> 1 distance embedding's gradients are sums of rows of both
    # signs that cancel to a few per cent of their terms - 22-bit against 24-bit products show there)
    tol = 3e-6 if power >= 1.0 else 2e-5
    for k in one:
        assert torch.isfinite(one[k]).all(), k
        assert errs[k] < tol, (k, errs[k])
    zero = run(False, 0.0)
    assert all(float(v.abs().max()) == 0.0 for v in zero.values())


def test_one_launch_noise_of_the_trainer():
    """geossl_ddm_noise (the trainer's five draws in one launch, Philox keyed by a seed from torch's generator): the same
    seed gives the same tensors, another seed others; positions noise ~ N(mu, sigma), distance noise ~ N(0, 1), levels
    uniform on [0, K); nothing is written past the tensors."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    b = make_batch(700, seed=2)
    batch = pg.Batch.from_numpy(b, DEV)
    n1, n2 = product_ncsn(128, 50, 2, DEV), product_ncsn(128, 30, 2, DEV)

    def draw(seed):
        torch.cuda.manual_seed(seed)
        return {k: v.clone() for k, v in pg.draw_step_noise_fused(batch, n1, n2, 0.25, 0.3).items()}

    a, a2, c = draw(5), draw(5), draw(6)
    assert all(torch.equal(a[k], a2[k]) for k in a) and not any(torch.equal(a[k], c[k]) for k in a)
    pn = a["pos_noise"].double()
    assert pn.shape == batch.positions.shape and abs(float(pn.mean()) - 0.25) < 5e-3 and abs(float(pn.std()) - 0.3) < 5e-3
    for k in ("dist_noise_1", "dist_noise_2"):
        d = a[k].double().view(-1)
        assert d.numel() == batch.super_edge_index.size(1)
        assert abs(float(d.mean())) < 1e-2 and abs(float(d.std()) - 1.0) < 1e-2
        assert abs(float((d ** 3).mean())) < 3e-2 and abs(float((d ** 4).mean()) - 3.0) < 0.1      # skewness, kurtosis
        assert float(d.abs().max()) > 3.5
    assert float((a["dist_noise_1"] * a["dist_noise_2"]).double().mean().abs()) < 1e-2                  # independent streams
    for k, K in (("noise_level_1", 50), ("noise_level_2", 30)):
        lv = a[k]
        assert lv.dtype == torch.long and lv.numel() == 700 and int(lv.min()) >= 0 and int(lv.max()) < K
        assert len(torch.unique(lv)) >= K - 3
    # in place into a graph's static inputs, with guard elements behind them
    into = {k: torch.full((v.numel() + 8,), 7, dtype=v.dtype, device=DEV) for k, v in a.items()}
    views = {k: into[k][:a[k].numel()].view(a[k].shape) for k in a}
    torch.cuda.manual_seed(5)
    pg.draw_step_noise_fused(batch, n1, n2, 0.25, 0.3, into=views)
    assert all(torch.equal(views[k], a[k]) for k in a) and all(bool((into[k][-8:] == 7).all()) for k in a)
