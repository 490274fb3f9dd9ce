"""fp32 torch-CPU restatement of the network arithmetic on the hot path.

Test infrastructure only — see oracle/__init__.py.  Parameters are passed as a flat
``{state_dict key: tensor}`` dict using the reference's own key names, so golden weights and
product-module weights drop straight in, and ``torch.autograd`` supplies reference gradients.
All random draws of the reference are explicit inputs here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .graph import radius_graph_np

# schnet.py:213 — torch.log(torch.tensor(2.0)).item(): an fp32 log(2) widened to a Python float
SSP_SHIFT = float(torch.log(torch.tensor(2.0)).item())


def shifted_softplus(x):
    """ShiftedSoftplus.forward, schnet.py:215-216 (softplus beta=1, threshold=20)."""
    return F.softplus(x) - SSP_SHIFT


def smearing_constants(cutoff, num_gaussians, start=0.0):
    """GaussianSmearing.__init__, schnet.py:201-203: offset = linspace(fp32); coeff is a Python
    double computed from the fp32 difference offset[1]-offset[0]."""
    offset = torch.linspace(start, cutoff, num_gaussians)
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    return offset, coeff


def gaussian_smearing(dist, offset, coeff):
    """GaussianSmearing.forward, schnet.py:205-207."""
    d = dist.view(-1, 1) - offset.view(1, -1)
    return torch.exp(coeff * torch.pow(d, 2))


def segment_reduce(src, index, reduce):
    """torch_scatter.scatter(src, index, dim=0, reduce) as used at schnet.py:115 / painn.py:266:
    dim_size = index.max()+1, mean = sum / clamp(count, 1) (third-party; parity unpinned)."""
    size = int(index.max()) + 1
    out = torch.zeros((size,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
    if reduce in ("add", "sum"):
        return out
    cnt = torch.zeros(size, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def schnet_forward(P, z, pos, batch, cutoff, num_interactions, readout="mean",
                   return_latent=False, edge_index=None, return_trace=False):
    """SchNet.forward, schnet.py:85-125, with the inactive dipole/mean/std/atomref/scale
    branches (103-121) left out — every caller in examples/ leaves them off.

    P keys: embedding.weight, distance_expansion.offset, interactions.{l}.mlp.{0,2}.{weight,bias},
    interactions.{l}.conv.lin1.weight, interactions.{l}.conv.lin2.{weight,bias},
    interactions.{l}.lin.{weight,bias}, lin1.*, lin2.*
    """
    assert z.dim() == 1 and z.dtype == torch.long  # schnet.py:86
    batch = torch.zeros_like(z) if batch is None else batch
    h = F.embedding(z, P["embedding.weight"])  # :89
    if edge_index is None:
        edge_index = torch.from_numpy(radius_graph_np(pos.detach().numpy(), cutoff, batch.numpy()))  # :91
    row, col = edge_index  # :92  row = source j, col = target i
    edge_weight = (pos[row] - pos[col]).norm(dim=-1)  # :93
    offset = P["distance_expansion.offset"]
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2  # :202
    edge_attr = gaussian_smearing(edge_weight, offset, coeff)  # :94
    trace = []
    for l in range(num_interactions):  # :96-97
        p = "interactions.%d." % l
        C = 0.5 * (torch.cos(edge_weight * math.pi / cutoff) + 1.0)  # :186 (no d<r mask)
        W = F.linear(shifted_softplus(F.linear(edge_attr, P[p + "mlp.0.weight"], P[p + "mlp.0.bias"])),
                     P[p + "mlp.2.weight"], P[p + "mlp.2.bias"]) * C.view(-1, 1)  # :187
        x = F.linear(h, P[p + "conv.lin1.weight"])  # :189
        msg = x[row] * W  # :194-195
        agg = torch.zeros_like(x).index_add(0, col, msg)  # :190 propagate(aggr="add")
        x = F.linear(agg, P[p + "conv.lin2.weight"], P[p + "conv.lin2.bias"])  # :191
        x = shifted_softplus(x)  # :165
        x = F.linear(x, P[p + "lin.weight"], P[p + "lin.bias"])  # :166
        h = h + x  # :97
        trace.append(h)
    h = F.linear(h, P["lin1.weight"], P["lin1.bias"])  # :99
    h = shifted_softplus(h)  # :100
    h = F.linear(h, P["lin2.weight"], P["lin2.bias"])  # :101
    out = segment_reduce(h, batch, readout)  # :115
    res = (out, h) if return_latent else out  # :123-125
    if return_trace:
        return res, dict(edge_index=edge_index, edge_weight=edge_weight, edge_attr=edge_attr, layers=trace)
    return res


def ncsn_sigmas(sigma_begin, sigma_end, num_noise_level):
    """NCSN_version_03.__init__, NCSN.py:178: fp64 geometric ladder rounded to fp32."""
    return torch.tensor(np.exp(np.linspace(np.log(sigma_begin), np.log(sigma_end), num_noise_level)),
                        dtype=torch.float32)


def _mlp(P, prefix, n_layers, x):
    """MultiLayerPerceptron.forward, NCSN.py:33-43 (relu between layers, none after the last)."""
    for i in range(n_layers):
        x = F.linear(x, P["%slayers.%d.weight" % (prefix, i)], P["%slayers.%d.bias" % (prefix, i)])
        if i < n_layers - 1:
            x = F.relu(x)
    return x


def ncsn_v03_forward(P, batch, super_edge_index, node_feature, distance, noise_level,
                     distance_noise, anneal_power, return_parts=False):
    """NCSN_version_03.forward, NCSN.py:183-220, with the two draws (`torch.randint` :190,
    `torch.randn_like` :194) passed in as `noise_level` [num_graphs] int64 and
    `distance_noise` [S,1] fp32."""
    edge2graph = batch[super_edge_index[0]]  # :187
    used_sigmas = P["sigmas"][noise_level]  # :191
    used_sigmas = used_sigmas[edge2graph].unsqueeze(-1)  # :192
    perturbed = distance + distance_noise * used_sigmas  # :196
    distance_emb = _mlp(P, "input_distance_mlp.", 2, perturbed)  # :197  (S,1)
    target = -1 / (used_sigmas ** 2) * (perturbed - distance)  # :199
    h_row, h_col = node_feature[super_edge_index[0]], node_feature[super_edge_index[1]]  # :201
    feat = torch.cat([h_row + h_col, distance_emb], dim=-1)  # :203  (S,F+1)
    scores = _mlp(P, "output_mlp.", 3, feat)  # :204
    scores = scores * (1.0 / used_sigmas)  # :205
    target = target.view(-1)
    scores = scores.view(-1)
    loss_e = 0.5 * ((scores - target) ** 2) * (used_sigmas.squeeze(-1) ** anneal_power)  # :209
    size = int(edge2graph.max()) + 1  # scatter_add dim_size (:210; quirk §9.8)
    loss_g = torch.zeros(size, dtype=loss_e.dtype).index_add(0, edge2graph, loss_e)  # :210
    loss = loss_g.mean()  # :212
    if return_parts:
        return loss, dict(loss_e=loss_e, loss_g=loss_g, scores=scores, target=target)
    return loss


def ncsn_relu_margin(P, batch, super_edge_index, node_feature, distance, noise_level, distance_noise):
    """Conditioning of a NCSN_version_03 evaluation for GRADIENT comparisons: the smallest |pre-activation| of any relu
    unit (NCSN.py:33-43) relative to the sum of the magnitudes of its terms.  A unit whose pre-activation is below the
    rounding noise of its own sum (ratio <~ 1e-7 in fp32) has an undetermined relu'(.) in {0, 1}: two correct fp32
    evaluations may then differ by a finite amount in the gradient of that row.  Tests that compare gradients evaluate
    this (in fp64) and require a safe margin for their inputs."""
    edge2graph = batch[super_edge_index[0]]
    sig = P["sigmas"][noise_level][edge2graph].unsqueeze(-1)
    pert = distance + distance_noise * sig
    margins = []
    zi = F.linear(pert, P["input_distance_mlp.layers.0.weight"], P["input_distance_mlp.layers.0.bias"])
    mi = F.linear(pert.abs(), P["input_distance_mlp.layers.0.weight"].abs(), P["input_distance_mlp.layers.0.bias"].abs())
    margins.append(zi.abs() / mi)
    emb = F.linear(F.relu(zi), P["input_distance_mlp.layers.1.weight"], P["input_distance_mlp.layers.1.bias"])
    x = torch.cat([node_feature[super_edge_index[0]] + node_feature[super_edge_index[1]], emb], dim=-1)
    for i in range(2):
        w, b = P["output_mlp.layers.%d.weight" % i], P["output_mlp.layers.%d.bias" % i]
        z, m = F.linear(x, w, b), F.linear(x.abs(), w.abs(), b.abs())
        margins.append(z.abs() / m)
        x = F.relu(z)
    return float(min(r.min() for r in margins))


def super_edge_distance(pos, super_edge_index):
    """pretrain_GeoSSL.py:199-201 / 203-205."""
    u = torch.index_select(pos, 0, super_edge_index[0])
    v = torch.index_select(pos, 0, super_edge_index[1])
    return torch.sqrt(torch.sum((u - v) ** 2, dim=1)).unsqueeze(1)


def do_ddm_schnet(P_model, P_ncsn1, P_ncsn2, x, positions, batch, super_edge_index, pos_noise,
                  noise_level_1, dist_noise_1, noise_level_2, dist_noise_2, cutoff,
                  num_interactions, anneal_power, readout="mean", normalize=False):
    """do_DDM, pretrain_GeoSSL.py:179-212 (SchNet branch), `perturb` (68-74) with the CPU
    normal draw passed in as `pos_noise` (already scaled: mu + sigma*eps)."""
    x_01 = x[:, 0]  # :182
    pos_01 = positions
    pos_02 = positions + pos_noise  # :72
    _, h1 = schnet_forward(P_model, x_01, pos_01, batch, cutoff, num_interactions, readout, True)  # :187
    _, h2 = schnet_forward(P_model, x_01, pos_02, batch, cutoff, num_interactions, readout, True)  # :188
    if normalize:  # :193-195
        h1 = F.normalize(h1, dim=-1)
        h2 = F.normalize(h2, dim=-1)
    d1 = super_edge_distance(pos_01, super_edge_index)  # :199-201
    d2 = super_edge_distance(pos_02, super_edge_index)  # :203-205
    l1 = ncsn_v03_forward(P_ncsn1, batch, super_edge_index, h1, d2, noise_level_1, dist_noise_1, anneal_power)  # :207
    l2 = ncsn_v03_forward(P_ncsn2, batch, super_edge_index, h2, d1, noise_level_2, dist_noise_2, anneal_power)  # :208
    return (l1 + l2) / 2  # :210


# ----------------------------------------------------------------------------- PaiNN (config 5)

def painn_forward(P, x, positions, radius_edge_index, batch, n_atom_basis, n_interactions,
                  cutoff, readout="add", epsilon=1e-8, return_latent=False, shared_filters=False,
                  shared_interactions=False):
    """PaiNN.forward, painn.py:216-269.  shared_filters: ONE filter of width 3F for all interactions (painn.py:178-181,
    242-243); shared_interactions: every interaction / mixing block is the same module (painn_utils.py:92-93 via
    painn.py:189-202) - block 0's parameters are read for every block."""
    F_ = n_atom_basis
    z = x[:, 0] if x.dim() == 2 else x  # :226-229
    idx_i, idx_j = radius_edge_index[0], radius_edge_index[1]  # :230
    r_ij = positions[idx_i] - positions[idx_j]  # :232
    n_atoms = z.size(0)
    d_ij = torch.norm(r_ij, dim=1, keepdim=True)  # :236  (E,1)
    dir_ij = r_ij / d_ij  # :237
    offsets, widths = P["radial_basis.offsets"], P["radial_basis.widths"]
    coeff = -0.5 / torch.pow(widths, 2)  # painn_utils.py:100
    phi = torch.exp(coeff * torch.pow(d_ij[..., None] - offsets, 2))  # painn_utils.py:101-102 (E,1,R)
    cut = P["cutoff_fn.cutoff"]
    fcut = 0.5 * (torch.cos(d_ij * math.pi / cut) + 1.0)  # painn_utils.py:152
    fcut = fcut * (d_ij < cut).to(d_ij.dtype)  # :154 (.float() there; dtype-generic so the oracle also runs in fp64)
    filters = F.linear(phi, P["filter_net.weight"], P["filter_net.bias"]) * fcut[..., None]  # :241
    filter_list = [filters] * n_interactions if shared_filters else torch.split(filters, 3 * F_, dim=-1)  # :242-245
    emb = P["embedding.weight"]
    q = F.embedding(z, emb, padding_idx=0)[:, None]  # :247
    mu = torch.zeros((q.shape[0], 3, q.shape[2]), dtype=q.dtype)  # :249
    for i in range(n_interactions):  # :251-253
        blk = 0 if shared_interactions else i
        p = "interactions.%d.interatomic_context_net." % blk
        xx = F.linear(F.silu(F.linear(q, P[p + "0.weight"], P[p + "0.bias"])), P[p + "1.weight"], P[p + "1.bias"])  # :53
        xj = xx[idx_j]  # :54
        muj = mu[idx_j]  # :55
        xx = filter_list[i] * xj  # :56
        dq, dmuR, dmumu = torch.split(xx, F_, dim=-1)  # :58
        dq = torch.zeros((n_atoms,) + tuple(dq.shape[1:]), dtype=dq.dtype).index_add(0, idx_i, dq)  # :59
        dmu = dmuR * dir_ij[..., None] + dmumu * muj  # :60
        dmu = torch.zeros((n_atoms,) + tuple(dmu.shape[1:]), dtype=dmu.dtype).index_add(0, idx_i, dmu)  # :61
        q = q + dq  # :63
        mu = mu + dmu  # :64
        m = "mixing.%d." % blk
        mu_mix = F.linear(mu, P[m + "mu_channel_mix.weight"])  # :100
        mu_V, mu_W = torch.split(mu_mix, F_, dim=-1)  # :101
        mu_Vn = torch.sqrt(torch.sum(mu_V ** 2, dim=-2, keepdim=True) + epsilon)  # :102
        ctx = torch.cat([q, mu_Vn], dim=-1)  # :104
        c = m + "intraatomic_context_net."
        xx = F.linear(F.silu(F.linear(ctx, P[c + "0.weight"], P[c + "0.bias"])), P[c + "1.weight"], P[c + "1.bias"])  # :105
        dq_intra, dmu_intra, dqmu_intra = torch.split(xx, F_, dim=-1)  # :107
        dmu_intra = dmu_intra * mu_W  # :108
        dqmu_intra = dqmu_intra * torch.sum(mu_V * mu_W, dim=1, keepdim=True)  # :110
        q = q + dq_intra + dqmu_intra  # :112
        mu = mu + dmu_intra  # :113
    q = q.squeeze(1)  # :255
    h = segment_reduce(q, batch, readout)  # :266
    return (h, q) if return_latent else h  # :267-269


def do_ddm_painn(P_model, P_ncsn1, P_ncsn2, x, positions, batch, radius_edge_index,
                 super_edge_index, pos_noise, noise_level_1, dist_noise_1, noise_level_2,
                 dist_noise_2, n_atom_basis, n_interactions, cutoff, anneal_power, readout="add"):
    """do_DDM, pretrain_GeoSSL.py:179-212 (PaiNN branch :190-191): the precomputed
    radius_edge_index of the clean geometry is reused for the perturbed view."""
    x_01 = x[:, 0]
    pos_02 = positions + pos_noise
    _, h1 = painn_forward(P_model, x_01, positions, radius_edge_index, batch, n_atom_basis, n_interactions, cutoff, readout, return_latent=True)
    _, h2 = painn_forward(P_model, x_01, pos_02, radius_edge_index, batch, n_atom_basis, n_interactions, cutoff, readout, return_latent=True)
    d1 = super_edge_distance(positions, super_edge_index)
    d2 = super_edge_distance(pos_02, super_edge_index)
    l1 = ncsn_v03_forward(P_ncsn1, batch, super_edge_index, h1, d2, noise_level_1, dist_noise_1, anneal_power)
    l2 = ncsn_v03_forward(P_ncsn2, batch, super_edge_index, h2, d1, noise_level_2, dist_noise_2, anneal_power)
    return (l1 + l2) / 2
