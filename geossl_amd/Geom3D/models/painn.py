"""PaiNN behind the reference's interface (Geom3D/models/painn.py:117-269) — BASELINE config 5.

Round-1 status: the constructor, parameter registration order and state_dict keys mirror the
reference so checkpoints interchange; the HIP message/mixing kernels are not
built yet, and because the product has no CPU/PyTorch fallback ``forward`` raises instead of
silently computing in ATen.
"""
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import xavier_uniform_, zeros_


class Dense(nn.Linear):
    """painn_utils.py:9-35: Linear with xavier weight / zero bias init and an activation attribute."""

    def __init__(self, in_features, out_features, bias=True, activation=None, weight_init=xavier_uniform_,
                 bias_init=zeros_):
        self.weight_init = weight_init
        self.bias_init = bias_init
        super().__init__(in_features, out_features, bias)
        self.activation = activation if activation is not None else nn.Identity()

    def reset_parameters(self):
        self.weight_init(self.weight)
        if self.bias is not None:
            self.bias_init(self.bias)


class GaussianRBF(nn.Module):
    """painn_utils.py:106-136 (non-trainable): offsets = linspace(start, cutoff, n_rbf), widths = |Δ|."""

    def __init__(self, n_rbf, cutoff, start=0.0):
        super().__init__()
        self.n_rbf = n_rbf
        offset = torch.linspace(start, cutoff, n_rbf)
        widths = torch.FloatTensor(torch.abs(offset[1] - offset[0]) * torch.ones_like(offset))
        self.register_buffer("widths", widths)
        self.register_buffer("offsets", offset)


class CosineCutoff(nn.Module):
    """painn_utils.py:158-177: buffer only; the cutoff function lives in the kernels."""

    def __init__(self, cutoff):
        super().__init__()
        self.register_buffer("cutoff", torch.FloatTensor([cutoff]))


class PaiNNInteraction(nn.Module):
    def __init__(self, n_atom_basis, activation):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.interatomic_context_net = nn.Sequential(
            Dense(n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None))


class PaiNNMixing(nn.Module):
    def __init__(self, n_atom_basis, activation, epsilon=1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.intraatomic_context_net = nn.Sequential(
            Dense(2 * n_atom_basis, n_atom_basis, activation=activation),
            Dense(n_atom_basis, 3 * n_atom_basis, activation=None))
        self.mu_channel_mix = Dense(n_atom_basis, 2 * n_atom_basis, activation=None, bias=False)
        self.epsilon = epsilon


def _replicate(factory, n, share):
    if share:
        return nn.ModuleList([factory()] * n)
    return nn.ModuleList([factory() for _ in range(n)])


class PaiNN(nn.Module):
    def __init__(self, n_atom_basis: int, n_interactions: int, n_rbf: int, cutoff: float, n_out: int, readout: str,
                 n_out_hidden: int = None, n_out_layers: int = 2, activation: Optional[Callable] = F.silu,
                 max_z: int = 100, shared_interactions: bool = False, shared_filters: bool = False,
                 epsilon: float = 1e-8):
        super().__init__()
        self.n_atom_basis = n_atom_basis
        self.n_interactions = n_interactions
        self.n_out = n_out
        self.n_out_hidden = n_out_hidden
        self.n_out_layers = n_out_layers
        self.activation = activation
        self.cutoff = cutoff
        self.cutoff_fn = CosineCutoff(cutoff)
        self.radial_basis = GaussianRBF(n_rbf=n_rbf, cutoff=cutoff)
        self.readout = readout
        self.embedding = nn.Embedding(max_z, n_atom_basis, padding_idx=0)
        self.share_filters = shared_filters
        if shared_filters:
            self.filter_net = Dense(self.radial_basis.n_rbf, 3 * n_atom_basis, activation=None)
        else:
            self.filter_net = Dense(self.radial_basis.n_rbf, self.n_interactions * n_atom_basis * 3, activation=None)
        self.interactions = _replicate(lambda: PaiNNInteraction(self.n_atom_basis, activation), self.n_interactions,
                                       shared_interactions)
        self.mixing = _replicate(lambda: PaiNNMixing(self.n_atom_basis, activation, epsilon), self.n_interactions,
                                 shared_interactions)

    def create_output_layers(self):
        """build_mlp(n_in, n_out, n_hidden, n_layers, activation), painn_utils.py:38-70."""
        n_in, n_out, n_layers = self.n_atom_basis, self.n_out, self.n_out_layers
        if self.n_out_hidden is None:
            c, neurons = n_in, []
            for _ in range(n_layers):
                neurons.append(c)
                c = max(n_out, c // 2)
            neurons.append(n_out)
        else:
            hid = [self.n_out_hidden] * (n_layers - 1) if isinstance(self.n_out_hidden, int) else list(self.n_out_hidden)
            neurons = [n_in] + hid + [n_out]
        layers = [Dense(neurons[i], neurons[i + 1], activation=self.activation) for i in range(n_layers - 1)]
        layers.append(Dense(neurons[-2], neurons[-1], activation=None))
        return nn.Sequential(*layers)

    def forward(self, x, positions, radius_edge_index, batch, return_latent=False):
        raise NotImplementedError(
            "PaiNN's HIP interaction/mixing kernels are not built yet (BASELINE config 5, SURVEY.md §8a P2-P4); "
            "geossl_amd has no PyTorch fallback by design")
