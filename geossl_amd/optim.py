"""Flat-buffer parameters + fused Adam (torch.optim.Adam semantics, pretrain_GeoSSL.py:343) and
the single-bucket gradient all-reduce for data parallelism."""
import torch

from . import _lib
from ._lib import call, ptr, stream


class FlatParams:
    """Re-homes the parameters of several modules into ONE contiguous fp32 buffer (parameters
    become views), with a matching flat gradient buffer: one Adam launch, one all-reduce."""

    def __init__(self, modules):
        seen, self.params = set(), []
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    self.params.append(p)
        self.trainable = [p for p in self.params if p.requires_grad]
        dev, n = self.trainable[0].device, sum(p.numel() for p in self.trainable)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.trainable:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.numel = n

    def zero_grad(self):
        self.grad.zero_()

    def rebind_grads(self):
        """Make sure every .grad still aliases the flat buffer (autograd may have replaced it)."""
        off = 0
        for p in self.trainable:
            k = p.numel()
            want = self.grad[off:off + k].view_as(p)
            if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                if p.grad is not None:
                    want.copy_(p.grad)
                p.grad = want
            off += k


class FusedAdam:
    """One geossl_adam_step launch over the flat buffer.  Matches torch.optim.Adam (amsgrad off)."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.fp = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0

    def step(self, grad_scale=1.0):
        _lib.require_cuda(self.fp.flat)
        self.step_count += 1
        call("geossl_adam_step", ptr(self.fp.flat), ptr(self.fp.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
             self.fp.numel, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             float(self.weight_decay), self.step_count, float(grad_scale), stream())

    def zero_grad(self):
        self.fp.zero_grad()


def cosine_annealing_lr(base_lr, epoch, T_max, eta_min=0.0):
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (pretrain_GeoSSL.py:350, stepped per epoch)."""
    import math
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / T_max)) / 2
