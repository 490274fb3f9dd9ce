import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().reshape(-1)
    b = torch.as_tensor(b).detach().double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b, floor=1e-6):
    """max |a-b| / (|b| + floor*max|b|) — elementwise relative error with an absolute floor."""
    a = torch.as_tensor(a).detach().double().reshape(-1)
    b = torch.as_tensor(b).detach().double().reshape(-1)
    return float(((a - b).abs() / (b.abs() + floor * b.abs().max().clamp_min(1e-30))).max())


def max_abs_rel(a, b):
    """max |a-b| / max |b| — error relative to the tensor's scale (near-zero elements of an fp32
    result carry absolute, not relative, rounding error)."""
    a = torch.as_tensor(a).detach().double().reshape(-1)
    b = torch.as_tensor(b).detach().double().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol, what=""):
    r, m = rel_err(a, b), max_abs_rel(a, b)
    assert r < tol and m < tol, "%s: l2-rel %.3e, max-abs/max %.3e (tol %.1e)" % (what, r, m, tol)
