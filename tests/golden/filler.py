"""Closed-form weight filler shared by make_golden.py and the tests, so no weight blob is stored.

Parameter ``name`` (its state_dict key) with ``numel`` entries gets
``w.flat[k] = scale * sin(0.37*k + phase(name))`` where ``scale`` depends on the shape only.
"""
import math
import zlib

import numpy as np
import torch


def _phase(name):
    return (zlib.crc32(name.encode()) % 1000) * 0.01


def fill_value(name, shape, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    k = np.arange(n, dtype=np.float64)
    if len(shape) >= 2:
        scale = 1.2 / math.sqrt(shape[-1])
    else:
        scale = 0.1
    if "embedding" in name:
        scale = 0.8
    v = scale * np.sin(0.37 * k + _phase(name))
    return torch.tensor(v.reshape(shape), dtype=dtype)


def fill_module_(module, skip=("sigmas",)):
    """In-place fill of every trainable parameter of an nn.Module (by state_dict key)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if any(s in name for s in skip):
                continue
            p.copy_(fill_value(name, tuple(p.shape), p.dtype))
    return module


def fill_dict(shapes, skip=("sigmas",)):
    """{key: shape} -> {key: tensor} using the same rule."""
    return {k: fill_value(k, tuple(s)) for k, s in shapes.items() if not any(x in k for x in skip)}


def grad_summary(g, nsample=64):
    """Compact, order-sensitive summary of a gradient tensor: [sum, l2, abs-sum] + strided samples."""
    f = g.detach().double().reshape(-1)
    idx = np.unique(np.linspace(0, f.numel() - 1, min(nsample, f.numel())).astype(np.int64))
    head = torch.stack([f.sum(), f.pow(2).sum().sqrt(), f.abs().sum()])
    return torch.cat([head, f[idx]]).numpy()
