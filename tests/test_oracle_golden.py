"""Pin the CPU oracle (oracle/) against the golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only.  Tolerance: restatement vs golden <= 1e-6 relative
(same torch CPU kernels underneath) — SURVEY.md §8c."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden, max_rel, rel_err
from filler import fill_dict, grad_summary
from oracle import graph, nets

TOL = 1e-6


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def schnet_shapes(cfg):
    F, G, L, C = cfg["hidden_channels"], cfg["num_gaussians"], cfg["num_interactions"], cfg["node_class"]
    s = {"embedding.weight": (C, F)}
    for l in range(L):
        p = "interactions.%d." % l
        s.update({p + "mlp.0.weight": (F, G), p + "mlp.0.bias": (F,), p + "mlp.2.weight": (F, F),
                  p + "mlp.2.bias": (F,), p + "conv.lin1.weight": (F, F), p + "conv.lin2.weight": (F, F),
                  p + "conv.lin2.bias": (F,), p + "lin.weight": (F, F), p + "lin.bias": (F,)})
    s.update({"lin1.weight": (F, F), "lin1.bias": (F,), "lin2.weight": (F, F), "lin2.bias": (F,)})
    return s


def schnet_params(cfg, requires_grad=True):
    P = fill_dict(schnet_shapes(cfg))
    for v in P.values():
        v.requires_grad_(requires_grad)
    P["distance_expansion.offset"] = nets.smearing_constants(cfg["cutoff"], cfg["num_gaussians"])[0]
    return P


def ncsn_shapes(F):
    return {"input_distance_mlp.layers.0.weight": (F, 1), "input_distance_mlp.layers.0.bias": (F,),
            "input_distance_mlp.layers.1.weight": (1, F), "input_distance_mlp.layers.1.bias": (1,),
            "output_mlp.layers.0.weight": (F, F + 1), "output_mlp.layers.0.bias": (F,),
            "output_mlp.layers.1.weight": (F // 2, F), "output_mlp.layers.1.bias": (F // 2,),
            "output_mlp.layers.2.weight": (1, F // 2), "output_mlp.layers.2.bias": (1,)}


def ncsn_params(F, K, scale=1.0):
    P = {k: (v * scale).requires_grad_() for k, v in fill_dict(ncsn_shapes(F)).items()}
    P["sigmas"] = nets.ncsn_sigmas(10.0, 0.01, K)
    return P


def test_g1_smearing_and_coeff():
    g = load_golden("g1_g2_smearing_ssp")
    for r, G in [(10.0, 51), (5.0, 51), (10.0, 50)]:
        key = "%g_%d" % (r, G)
        off, coeff = nets.smearing_constants(r, G)
        assert coeff == float(g["coeff_" + key])
        assert torch.equal(off, t(g["offset_" + key]))
        y = nets.gaussian_smearing(t(g["d_" + key]), off, coeff)
        assert torch.equal(y, t(g["y_" + key]))
    # the two scalars quoted in SURVEY §8a S4
    assert abs(nets.smearing_constants(10.0, 51)[1] - (-12.4999996275)) < 1e-9
    assert abs(nets.smearing_constants(5.0, 51)[1] - (-49.9999985099)) < 1e-9


def test_g2_shifted_softplus():
    g = load_golden("g1_g2_smearing_ssp")
    assert nets.SSP_SHIFT == float(g["ssp_shift"])
    assert torch.equal(nets.shifted_softplus(t(g["ssp_x"])), t(g["ssp_y"]))


@pytest.mark.parametrize("r", [5.0, 10.0, 1.5])
def test_g3_radius_graph_bit_exact(r):
    g = load_golden("g3_radius_graph")
    e = graph.radius_graph_np(g["positions"], r, g["batch"])
    assert e.dtype == np.int64 and np.array_equal(e, g["edge_index_%g" % r])
    pos = t(g["positions"])
    w = (pos[t(e[0])] - pos[t(e[1])]).norm(dim=-1)
    assert torch.equal(w, t(g["edge_weight_%g" % r]))


def test_g3_cap_triggers_and_is_asymmetric():
    g = load_golden("g3_radius_graph")
    e = g["edge_index_10"]
    deg = np.bincount(e[1], minlength=len(g["batch"]))
    assert deg.max() == 33 and (deg == 32).sum() > 0
    s = set(map(tuple, e.T.tolist()))
    assert any((b, a) not in s for a, b in s)  # capped graph is not symmetric


def test_collate_matches_reference_semantics():
    g = load_golden("g5_ncsn_comb_K50_p5_last1")
    sizes = np.bincount(g["batch"])
    mols, off = [], 0
    for n in sizes:
        mols.append((g["x"][off:off + n], g["positions"][off:off + n]))
        off += n
    c = graph.collate_np(mols, "combination")
    for k in ("x", "positions", "batch", "super_edge_index"):
        assert np.array_equal(c[k], g[k]), k
    assert graph.super_edges_np(1).shape == (2, 0)
    assert graph.super_edges_np(3, "permutation").T.tolist() == [[0, 1], [0, 2], [1, 0], [1, 2], [2, 0], [2, 1]]


@pytest.mark.parametrize("tag", ["reduced", "full_r5", "full_r10"])
def test_g4_schnet_forward_and_grads(tag):
    g = load_golden("g4_schnet_" + tag)
    cfg = json.loads(str(g["cfg"]))
    P = schnet_params(cfg)
    out, h = nets.schnet_forward(P, t(g["x"])[:, 0], t(g["positions"]), t(g["batch"]), cfg["cutoff"],
                                 cfg["num_interactions"], cfg["readout"], return_latent=True)
    assert max_rel(out, g["out"]) < TOL and max_rel(h, g["h"]) < TOL
    loss = (out ** 2).sum() + (h ** 2).sum() * 0.5
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(P[k[5:]].grad), g[k]) < 5e-6, k
        if k.startswith("grad/"):
            assert rel_err(P[k[5:]].grad, g[k]) < 5e-6, k


@pytest.mark.parametrize("tag", ["reduced", "full_r5"])
def test_g9_schnet_forces(tag):
    """Position gradient (finetune_md17.py:46) of the oracle against the forces of the unmodified reference."""
    g = load_golden("g9_schnet_forces_" + tag)
    cfg = json.loads(str(g["cfg"]))
    P = schnet_params(cfg)
    pos = t(g["positions"]).clone().requires_grad_(True)
    out = nets.schnet_forward(P, t(g["x"])[:, 0], pos, t(g["batch"]), cfg["cutoff"], cfg["num_interactions"],
                              cfg["readout"])
    energy = (out * torch.cos(torch.arange(out.size(1), dtype=torch.float32))).sum(dim=1)
    assert rel_err(energy, g["energy"]) < 1e-5  # a cos-weighted sum over the features: cancellation, not drift
    force = -torch.autograd.grad(energy.sum(), pos)[0]
    assert rel_err(force, g["force"]) < 2e-5  # two fp32 evaluations in different summation orders


@pytest.mark.parametrize("tag", ["comb_K50_p2", "comb_K30_p0.05", "comb_K50_p5_last1", "perm_K30_p10"])
def test_g5_ncsn(tag):
    g = load_golden("g5_ncsn_" + tag)
    P = ncsn_params(128, int(g["K"]))
    assert torch.equal(P["sigmas"], t(g["sigmas"]))
    h = t(g["h"]).clone().requires_grad_()
    loss = nets.ncsn_v03_forward(P, t(g["batch"]), t(g["super_edge_index"]), h, t(g["distance"]),
                                 t(g["noise_level"]), t(g["distance_noise"]), float(g["anneal_power"]))
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    assert rel_err(h.grad, g["grad_h"]) < 5e-6
    for k in g:
        if k.startswith("grad/"):
            assert rel_err(P[k[5:]].grad, g[k]) < 5e-6, k


def test_g5_last_graph_single_atom_quirk():
    """NCSN.py:210-212: the mean divides by max(edge2graph)+1, which is B-1 when the last
    molecule has one atom (no super-edges)."""
    g = load_golden("g5_ncsn_comb_K50_p5_last1")
    b, sei = t(g["batch"]), t(g["super_edge_index"])
    assert int(b.max()) + 1 == 4 and int(b[sei[0]].max()) + 1 == 3


@pytest.mark.parametrize("tag", ["reduced", "full"])
def test_g6_do_ddm(tag):
    g = load_golden("g6_ddm_" + tag)
    cfg = json.loads(str(g["cfg"]))
    F = cfg["hidden_channels"]
    Pm, P1, P2 = schnet_params(cfg), ncsn_params(F, 50), ncsn_params(F, 50, 0.9)
    loss = nets.do_ddm_schnet(Pm, P1, P2, t(g["x"]), t(g["positions"]), t(g["batch"]), t(g["super_edge_index"]),
                              t(g["pos_noise"]), t(g["noise_level_1"]), t(g["dist_noise_1"]),
                              t(g["noise_level_2"]), t(g["dist_noise_2"]), cfg["cutoff"],
                              cfg["num_interactions"], 2, cfg["readout"])
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    for k in g:
        if k.startswith("gsum/") or k.startswith("grad/"):
            _, m, name = k.split("/", 2)
            grad = {"model": Pm, "ncsn1": P1, "ncsn2": P2}[m][name].grad
            got = grad_summary(grad) if k.startswith("gsum/") else grad
            assert rel_err(got, g[k]) < 2e-5, k


def painn_params(cfg):
    F, L, R, Z = cfg["n_atom_basis"], cfg["n_interactions"], cfg["n_rbf"], cfg["max_z"]
    nf = 1 if cfg.get("shared_filters") else L          # painn.py:178-187
    s = {"embedding.weight": (Z, F), "filter_net.weight": (nf * 3 * F, R), "filter_net.bias": (nf * 3 * F,)}
    if cfg.get("shared_interactions"):                   # one block, replicated (painn_utils.py:92-93)
        L = 1
    for i in range(L):
        p = "interactions.%d.interatomic_context_net." % i
        s.update({p + "0.weight": (F, F), p + "0.bias": (F,), p + "1.weight": (3 * F, F), p + "1.bias": (3 * F,)})
    for i in range(L):
        m = "mixing.%d." % i
        s.update({m + "intraatomic_context_net.0.weight": (F, 2 * F), m + "intraatomic_context_net.0.bias": (F,),
                  m + "intraatomic_context_net.1.weight": (3 * F, F), m + "intraatomic_context_net.1.bias": (3 * F,),
                  m + "mu_channel_mix.weight": (2 * F, F)})
    P = {k: v.requires_grad_() for k, v in fill_dict(s).items()}
    off = torch.linspace(0.0, cfg["cutoff"], R)
    P["radial_basis.offsets"] = off
    P["radial_basis.widths"] = torch.abs(off[1] - off[0]) * torch.ones_like(off)
    P["cutoff_fn.cutoff"] = torch.tensor([cfg["cutoff"]], dtype=torch.float32)
    return P


def test_g7_painn_forward_grads_and_ddm():
    g = load_golden("g7_painn")
    cfg = json.loads(str(g["cfg"]))
    P = painn_params(cfg)
    out, q = nets.painn_forward(P, t(g["x"]), t(g["positions_perturbed"]), t(g["radius_edge_index"]), t(g["batch"]),
                                cfg["n_atom_basis"], cfg["n_interactions"], cfg["cutoff"], cfg["readout"],
                                return_latent=True)
    assert max_rel(out, g["out"]) < TOL and max_rel(q, g["q"]) < TOL
    # padding_idx=0: hydrogens' embedding row gets no gradient (painn.py:174)
    loss = (out ** 2).sum() + 0.5 * (q ** 2).sum()
    loss.backward()
    assert float(P["embedding.weight"].grad[0].abs().max()) == 0.0
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(P[k[5:]].grad), g[k]) < 5e-6, k
    d = load_golden("g7_painn_ddm")
    for v in P.values():
        v.grad = None
    P1, P2 = ncsn_params(128, 50), ncsn_params(128, 50)
    loss = nets.do_ddm_painn(P, P1, P2, t(g["x"]), t(g["positions"]), t(g["batch"]), t(g["radius_edge_index"]),
                             t(g["super_edge_index"]), t(d["pos_noise"]), t(d["noise_level_1"]), t(d["dist_noise_1"]),
                             t(d["noise_level_2"]), t(d["dist_noise_2"]), cfg["n_atom_basis"],
                             cfg["n_interactions"], cfg["cutoff"], 2, cfg["readout"])
    assert rel_err(loss, d["loss"]) < TOL
    loss.backward()
    for k in d:
        if k.startswith("gsum/"):
            _, m, name = k.split("/", 2)
            assert rel_err(grad_summary({"model": P, "ncsn1": P1, "ncsn2": P2}[m][name].grad), d[k]) < 2e-5, k


@pytest.mark.parametrize("tag", ["reduced", "full_r5"])
def test_g10_force_training_double_backward(tag):
    """finetune_md17.py:46-54 on the oracle: force = -grad(E, pos, create_graph=True), a loss on energy and force,
    backward into the parameters (second differentiation) - against the unmodified reference."""
    g = load_golden("g10_schnet_force_training_" + tag)
    cfg = json.loads(str(g["cfg"]))
    P = schnet_params(cfg)
    pos = t(g["positions"]).clone().requires_grad_(True)
    out = nets.schnet_forward(P, t(g["x"])[:, 0], pos, t(g["batch"]), cfg["cutoff"], cfg["num_interactions"],
                              cfg["readout"])
    from filler import fill_module_
    head = fill_module_(torch.nn.Linear(out.size(1), 1))  # graph_pred_linear, finetune_md17.py:33
    energy = head(out).squeeze(1)
    force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    crit = torch.nn.MSELoss()
    loss = 1.0 * crit(energy, t(g["actual_energy"])) + 10.0 * crit(force, t(g["actual_force"]))
    assert rel_err(loss, g["loss"]) < 2e-5
    loss.backward()
    assert rel_err(pos.grad, g["grad_pos"]) < 5e-5
    assert rel_err(head.weight.grad, g["head_grad/weight"]) < 5e-5 and rel_err(head.bias.grad, g["head_grad/bias"]) < 5e-5
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(P[k[5:]].grad), g[k]) < 5e-5, k
        if k.startswith("grad/"):
            assert rel_err(P[k[5:]].grad, g[k]) < 5e-5, k


def test_g11_collate_and_tuple_extractor_vs_reference_loader():
    """oracle.graph.collate_np against BatchAtomTuple.from_data_list / AtomTupleExtractor of the unmodified reference
    (ratio = 1, both options, with a per-molecule radius_edge_index)."""
    g = load_golden("g11_loader")
    sizes = g["sizes"].tolist()
    mols, off = [], 0
    for n in sizes:
        mols.append((g["x"][off:off + n], g["positions"][off:off + n]))
        off += n
    for option in ("combination", "permutation"):
        c = graph.collate_np(mols, option, radius=5.0)
        assert np.array_equal(c["super_edge_index"], g["sei/%s_1" % option])
        assert np.array_equal(c["batch"], g["batch/%s_1" % option])
        assert np.array_equal(c["radius_edge_index"], g["rei/%s_1" % option])
        assert int(g["num_graphs/%s_1" % option]) == len(sizes)


@pytest.mark.parametrize("tag", ["reduced", "full"])
def test_g12_three_step_trajectory(tag):
    """Three DDM steps with stock torch.optim.Adam over the three parameter groups (pretrain_GeoSSL.py:258-260,
    333-343) on the oracle: losses and parameters after step 3 against the unmodified reference."""
    g = load_golden("g12_ddm_trajectory_" + tag)
    cfg = json.loads(str(g["cfg"]))
    F = cfg["hidden_channels"]
    Pm, P1, P2 = schnet_params(cfg), ncsn_params(F, 50), ncsn_params(F, 50, 0.9)
    groups = [{"params": [p for p in P.values() if p.requires_grad], "lr": 5e-4} for P in (Pm, P1, P2)]
    opt = torch.optim.Adam(groups, lr=5e-4, weight_decay=0)
    for step in range(3):
        loss = nets.do_ddm_schnet(Pm, P1, P2, t(g["x"]), t(g["positions"]), t(g["batch"]), t(g["super_edge_index"]),
                                  t(g["pos_noise/%d" % step]), t(g["noise_level_1/%d" % step]),
                                  t(g["dist_noise_1/%d" % step]), t(g["noise_level_2/%d" % step]),
                                  t(g["dist_noise_2/%d" % step]), cfg["cutoff"], cfg["num_interactions"], 2,
                                  cfg["readout"])
        assert rel_err(loss, g["loss/%d" % step]) < 1e-5, step
        opt.zero_grad()
        loss.backward()
        opt.step()
    for k in g:
        if k.startswith("psum/") or k.startswith("param/"):
            _, m, name = k.split("/", 2)
            p = {"model": Pm, "ncsn1": P1, "ncsn2": P2}[m][name].detach()
            got = grad_summary(p) if k.startswith("psum/") else p
            assert rel_err(got, g[k]) < 1e-5, k


def test_g13_painn_force_training():
    """finetune_md17.py:38-54 with PaiNN on the oracle (forces, loss on energy and force, second differentiation)."""
    from filler import fill_module_
    g = load_golden("g13_painn_force_training")
    cfg = json.loads(str(g["cfg"]))
    P = painn_params(cfg)
    head = fill_module_(torch.nn.Sequential(torch.nn.Linear(128, 64), torch.nn.SiLU(), torch.nn.Linear(64, 1)))
    head_ref_names = {"0.weight": "0.weight", "0.bias": "0.bias", "1.weight": "2.weight", "1.bias": "2.bias"}
    # create_output_layers() = Sequential(Dense(F, F/2, silu), Dense(F/2, 1)): the filler keys are the Sequential indices
    # 0 / 1 there; refill under those names so that both heads carry identical weights
    from filler import fill_value
    with torch.no_grad():
        for ref_name, mine in head_ref_names.items():
            p = dict(head.named_parameters())[mine]
            p.copy_(fill_value(ref_name, tuple(p.shape)))
    pos = t(g["positions"]).clone().requires_grad_(True)
    rep = nets.painn_forward(P, t(g["x"]), pos, t(g["radius_edge_index"]), t(g["batch"]), 128, 3, 5.0, "add")
    assert max_rel(rep, g["rep"]) < TOL
    energy = head(rep).squeeze(1)
    force = -torch.autograd.grad(energy, pos, torch.ones_like(energy), create_graph=True, retain_graph=True)[0]
    assert rel_err(energy, g["energy"]) < 1e-5 and rel_err(force, g["force"]) < 2e-5
    crit = torch.nn.MSELoss()
    loss = 1.0 * crit(energy, t(g["actual_energy"])) + 10.0 * crit(force, t(g["actual_force"]))
    assert rel_err(loss, g["loss"]) < 2e-5
    loss.backward()
    assert rel_err(pos.grad, g["grad_pos"]) < 5e-5
    for k in g:
        if k.startswith("gsum/"):
            assert rel_err(grad_summary(P[k[5:]].grad), g[k]) < 5e-5, k
    for ref_name, mine in head_ref_names.items():
        assert rel_err(dict(head.named_parameters())[mine].grad, g["head_grad/" + ref_name]) < 5e-5, ref_name


def finetune_epochs(g, forward, params, head_w, head_b, device="cpu"):
    """The loop of examples/finetune_qm9.py:163-275 (forward -> graph_pred_linear -> squeeze -> L1 on the normalised
    target -> zero_grad / backward / Adam step; CosineAnnealingLR(T_max = 100) stepped per epoch, :500-507) over the
    fixture's three training batches for two epochs, then eval() (:278-384) over its two evaluation batches:
    `forward(x0, positions, batch) -> [B, F]`.  Returns (per-step losses, eval predictions)."""
    tm, ts, task = float(g["TRAIN_mean"]), float(g["TRAIN_std"]), int(g["task_id"])
    opt = torch.optim.Adam([{"params": params, "lr": 5e-4}, {"params": [head_w, head_b], "lr": 5e-4}], lr=5e-4,
                           weight_decay=0)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 100)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    losses = []
    for epoch in (1, 2):
        for i in range(3):
            x, pos, bat, y = (dev(g["train/%d/%s" % (i, k)]) for k in ("x", "positions", "batch", "y"))
            rep = forward(x[:, 0], pos, bat)                                             # :179
            pred = torch.nn.functional.linear(rep, head_w, head_b).squeeze()             # :250
            B = pred.size()[0]
            yy = (y.view(B, -1)[:, task] - tm) / ts                                      # :254-257
            loss = torch.nn.functional.l1_loss(pred, yy)                                 # :259, --loss mae
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        sched.step()                                                                      # :270-271
    scores = []
    with torch.no_grad():                                                                 # :278
        for i in range(2):
            x, pos, bat = (dev(g["eval/%d/%s" % (i, k)]) for k in ("x", "positions", "batch"))
            pred = torch.nn.functional.linear(forward(x[:, 0], pos, bat), head_w, head_b).squeeze()
            scores.append(pred * ts + tm)                                                 # :374
    return losses, torch.cat(scores).cpu(), float(opt.param_groups[0]["lr"])


def test_g14_finetune_qm9_schnet_two_epochs_and_eval():
    """BASELINE config 1 (examples/finetune_qm9.py, SchNet at config.py's defaults: cutoff 10, 51 gaussians, readout
    mean, batches of 32 QM9-sized molecules) on the oracle against the reference's own train() / eval(): per-step L1
    losses, the parameters after two epochs, the evaluation predictions and their MAE."""
    g = load_golden("g14_finetune_qm9_schnet")
    cfg = json.loads(str(g["cfg"]))
    P = schnet_params(cfg)
    hw = fill_dict({"weight": (1, 128), "bias": (1,)})
    head_w, head_b = hw["weight"].requires_grad_(), hw["bias"].requires_grad_()
    fwd = lambda z, pos, bat: nets.schnet_forward(P, z, pos, bat, cfg["cutoff"], cfg["num_interactions"], cfg["readout"])
    losses, scores, lr = finetune_epochs(g, fwd, [p for p in P.values() if p.requires_grad], head_w, head_b)
    assert rel_err(losses, g["losses"]) < 1e-5 and abs(lr - float(g["lr_after"])) < 1e-12
    assert rel_err(scores, g["y_scores"]) < 1e-5
    y_true = torch.cat([t(g["eval/%d/y" % i]).view(32, -1)[:, int(g["task_id"])] for i in range(2)])
    assert torch.equal(y_true, t(g["y_true"]))
    assert abs(float((scores - y_true).abs().mean()) - float(g["mae"])) < 1e-5 * float(g["mae"])
    for k in g:
        if k.startswith("psum/"):
            assert rel_err(grad_summary(P[k[5:]].detach()), g[k]) < 1e-5, k
    assert rel_err(head_w, g["head/weight"]) < 1e-5 and rel_err(head_b, g["head/bias"]) < 1e-5


def _check_gsums(g, grads, tol=2e-5):
    for key in g:
        if key.startswith("gsum/"):   # (l2-relative over the summary, like the other fixtures: fp32 sums depend on the thread count)
            assert rel_err(grad_summary(grads[key[5:]]), g[key]) < tol, key


def test_g15_configurations_off_the_defaults():
    """G15 (round 6): SchNet with hidden_channels != num_filters at widths that are no multiple of 32 and more than 64
    gaussians; PaiNN at an odd width / radial basis and with shared_filters + shared_interactions - the oracle against
    fixtures of the unmodified reference (schnet.py:17-30; painn.py:140-141,178-202,242-243)."""
    from helpers import schnet_shapes
    for tag in ("a", "b"):
        g = load_golden("g15_schnet_widths_" + tag)
        cfg = json.loads(str(g["cfg"]))
        H, Fl, G, L, C = (cfg[k] for k in ("hidden_channels", "num_filters", "num_gaussians", "num_interactions", "node_class"))
        shapes = {"embedding.weight": (C, H), "lin1.weight": (H, H), "lin1.bias": (H,), "lin2.weight": (H, H), "lin2.bias": (H,)}
        for l in range(L):
            p = "interactions.%d." % l
            shapes.update({p + "mlp.0.weight": (Fl, G), p + "mlp.0.bias": (Fl,), p + "mlp.2.weight": (Fl, Fl),
                           p + "mlp.2.bias": (Fl,), p + "conv.lin1.weight": (Fl, H), p + "conv.lin2.weight": (H, Fl),
                           p + "conv.lin2.bias": (H,), p + "lin.weight": (H, H), p + "lin.bias": (H,)})
        P = {k: v.requires_grad_() for k, v in fill_dict(shapes).items()}
        P["distance_expansion.offset"] = nets.smearing_constants(cfg["cutoff"], G)[0]
        out, h = nets.schnet_forward(P, t(g["x"])[:, 0], t(g["positions"]), t(g["batch"]), cfg["cutoff"], L, cfg["readout"],
                                     return_latent=True)
        assert rel_err(out, g["out"]) < 1e-6 and rel_err(h, g["h"]) < 1e-6
        w = lambda t_: torch.cos(0.1 * torch.arange(t_.numel(), dtype=torch.float32)).view(t_.shape)
        ((out * w(out)).sum() + (h * w(h)).sum()).backward()
        _check_gsums(g, {k: v.grad for k, v in P.items() if v.grad is not None})
    for tag in ("a", "b"):
        g = load_golden("g15_painn_variants_" + tag)
        cfg = json.loads(str(g["cfg"]))
        P = painn_params(cfg)
        out, q = nets.painn_forward(P, t(g["x"]), t(g["positions"]), t(g["radius_edge_index"]), t(g["batch"]),
                                    cfg["n_atom_basis"], cfg["n_interactions"], cfg["cutoff"], cfg["readout"],
                                    return_latent=True, shared_filters=bool(cfg.get("shared_filters")),
                                    shared_interactions=bool(cfg.get("shared_interactions")))
        assert rel_err(out, g["out"]) < 1e-6 and rel_err(q, g["q"]) < 1e-6
        w = lambda t_: torch.cos(0.1 * torch.arange(t_.numel(), dtype=torch.float32)).view(t_.shape)
        ((out * w(out)).sum() + (q * w(q)).sum()).backward()
        _check_gsums(g, {k: v.grad for k, v in P.items() if v.grad is not None})
