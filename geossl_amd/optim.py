"""Flat-buffer parameters + fused Adam (torch.optim.Adam semantics, pretrain_GeoSSL.py:343) and
the single-bucket gradient all-reduce for data parallelism."""
import os

import torch
from .switches import env as _env

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


# ---- the reference's own optimizer on the flat path ----------------------------------------------------------------------
# The reference loop ends in ``optimizer.step()`` of a stock ``torch.optim.Adam`` over three parameter groups
# (examples/pretrain_GeoSSL.py:258-260,332-343).  Its default (foreach) path costs ~0.6 ms of Python and two dozen
# launches per step - at the reference's own batch size (128) more than the whole forward + backward of the step.  When
# do_DDM's graph path owns the parameters' memory (``ParamHome``: every trainable parameter a view of ONE flat buffer,
# every gradient it hands to autograd a view of ONE flat buffer at the same offsets), the same update is ONE
# ``geossl_adam_step`` launch.  A global optimizer step pre-hook recognises that situation on an UNMODIFIED
# ``torch.optim.Adam`` instance, runs the launch, and leaves the stock ``step()`` nothing to do (it skips parameters
# without a gradient); a post-hook puts the gradients back.  The arithmetic is torch's foreach Adam bit for bit
# (csrc/ddm.hip: k_adam; tools/probes/adam_probe.hip), ``optimizer.state`` carries ``step`` / ``exp_avg`` /
# ``exp_avg_sq`` per parameter as torch lays them out (views of flat buffers), ``state_dict()`` / ``load_state_dict()`` work,
# and anything the hook does not recognise - another optimizer class, amsgrad, a closure, gradients that are not the flat
# views (a clipped copy, an eager step) - takes the stock path untouched.  GEOSSL_NO_FUSED_ADAM disables the hook.
_HOMES = None
_HOOKED = [False]


class ParamHome:
    """The trainable parameters of a (backbone, head, head) triple re-homed into one flat fp32 buffer, in the order of
    the flat gradient buffer do_DDM's graph path hands to autograd."""

    def __init__(self, params):
        import weakref
        global _HOMES
        self.params = list(params)
        self.index = {id(p): k for k, p in enumerate(self.params)}
        self.numels = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in self.numels:
            self.offsets.append(self.offsets[-1] + n)
        self.numel = self.offsets[-1]
        dev = self.params[0].device
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        for p, a, b in zip(self.params, self.offsets[:-1], self.offsets[1:]):
            self.flat[a:b].copy_(p.data.reshape(-1))
            p.data = self.flat[a:b].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = torch.zeros(len(self.params), dtype=torch.float32)  # host, like torch's non-capturable `step`
        self.step_count = 0
        if _HOMES is None:
            _HOMES = weakref.WeakSet()
        _HOMES.add(self)
        install_adam_hook()

    @staticmethod
    def of(params):
        """A home for these parameters, or None when they cannot be re-homed (not fp32 / CUDA, or already views of a
        larger storage - someone else, e.g. a DDMTrainer, owns their memory)."""
        params = list(params)
        if not params or _env("GEOSSL_NO_FUSED_ADAM"):
            return None
        for p in params:
            if (not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous()
                    or p.untyped_storage().nbytes() != p.numel() * 4):
                return None
        return ParamHome(params)

    def intact(self, full=False):
        """The parameters still live in the flat buffer (first and last as per-step sentinels; `full`: every one)."""
        base = self.flat.data_ptr()
        if full:
            return all(p.data_ptr() == base + 4 * a for p, a in zip(self.params, self.offsets))
        return (self.params[0].data_ptr() == base + 4 * self.offsets[0]
                and self.params[-1].data_ptr() == base + 4 * self.offsets[-2])


class _AdamPlan:
    """What the step hook knows about one torch.optim.Adam instance whose trainable parameters are exactly a home's.
    The per-step checks are a handful of pointer comparisons; the full walks (hyperparameters per parameter, optimizer
    state) are redone only when their cheap fingerprints change."""

    def __init__(self, optimizer, home):
        self.home = home
        self.stash = None
        self._hkey, self._runs, self._shared = None, None, None
        self._state_ok = False

    def hyper(self, optimizer):
        """[(first param, end param, lr)] runs of equal lr in home order + shared (betas, eps, weight_decay), or None."""
        groups = optimizer.param_groups
        key = tuple((g["lr"], g["betas"], g["eps"], g["weight_decay"], g.get("amsgrad"), g.get("maximize"),
                     g.get("capturable"), g.get("differentiable"), g.get("fused"), len(g["params"])) for g in groups)
        if key == self._hkey:
            return self._runs, self._shared
        self._hkey, self._runs, self._shared = key, None, None
        lr_of, shared = {}, None
        for g in groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or g.get("fused"):
                return None, None
            h = (tuple(g["betas"]), g["eps"], g["weight_decay"])
            if any(isinstance(v, torch.Tensor) for v in (g["lr"],) + h[0] + h[1:]):
                return None, None
            if shared is None:
                shared = h
            elif shared != h:
                return None, None
            for p in g["params"]:
                if p.requires_grad:
                    lr_of[id(p)] = g["lr"]
        if shared is None or set(lr_of) != set(self.home.index):
            return None, None
        runs = []
        for k, p in enumerate(self.home.params):
            lr = lr_of[id(p)]
            if runs and runs[-1][2] == lr:
                runs[-1][1] = k + 1
            else:
                runs.append([k, k + 1, lr])
        self._runs, self._shared = runs, shared
        return runs, shared

    def _ours(self, optimizer, k):
        home, p = self.home, self.home.params[k]
        st = optimizer.state.get(p)
        a = home.offsets[k]
        return (st is not None and len(st) == 3 and st["exp_avg"].data_ptr() == home.exp_avg.data_ptr() + 4 * a
                and st["exp_avg_sq"].data_ptr() == home.exp_avg_sq.data_ptr() + 4 * a
                and st["step"].data_ptr() == home.steps.data_ptr() + 4 * k)

    def adopt_state(self, optimizer):
        """optimizer.state of our parameters as views of the home's flat buffers (existing stock state is copied in).
        Once adopted, the first and the last parameter's entries are the per-step sentinels (load_state_dict replaces
        every entry)."""
        home = self.home
        last = len(home.params) - 1
        if self._state_ok and self._ours(optimizer, 0) and self._ours(optimizer, last):
            return True
        self._state_ok = False
        steps = set()
        for k, p in enumerate(home.params):
            a, b = home.offsets[k], home.offsets[k + 1]
            if not self._ours(optimizer, k):
                st = optimizer.state.get(p)
                if st is not None and len(st) > 0:  # state made by stock steps (or loaded): move it in
                    home.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
                    home.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
                    home.steps[k] = float(st["step"])
                else:
                    home.exp_avg[a:b].zero_()
                    home.exp_avg_sq[a:b].zero_()
                    home.steps[k] = 0.0
                optimizer.state[p] = {"step": home.steps[k], "exp_avg": home.exp_avg[a:b].view_as(p),
                                      "exp_avg_sq": home.exp_avg_sq[a:b].view_as(p)}
            steps.add(float(home.steps[k]))
        if len(steps) != 1:
            return False  # parameters at different step counts: not one launch's worth
        home.step_count = int(steps.pop())
        self._state_ok = True
        return True

    def try_step(self, optimizer):
        done = self._try_step(optimizer)
        if not done:
            # the stock step runs on this call and advances the per-parameter `step` tensors itself: the next fused
            # step reads the count back from them
            self._state_ok = False
        return done

    def _try_step(self, optimizer):
        home = self.home
        grads = [p.grad for p in home.params]
        g0 = grads[0]
        if g0 is None or not home.intact(full=True):  # (every parameter: a re-pointed middle one must not go unnoticed)
            return False
        if g0.dtype != torch.float32:
            return False
        base = g0.data_ptr() - 4 * home.offsets[0]
        for g, a in zip(grads, home.offsets):
            # every gradient where the flat buffer do_DDM's backward handed to autograd has it (a gradient that was
            # replaced, cloned or never arrived is somewhere else, or None)
            if g is None or g.data_ptr() != base + 4 * a:
                return False
        last = grads[-1]
        if last.numel() != home.numels[-1] or not last.is_contiguous():
            return False
        runs, shared = self.hyper(optimizer)
        if runs is None or not self.adopt_state(optimizer):
            return False
        (b1, b2), eps, wd = shared
        home.step_count += 1
        st = stream()
        for k0, k1, lr in runs:
            a, b = home.offsets[k0], home.offsets[k1]
            call("geossl_adam_step", home.flat.data_ptr() + 4 * a, base + 4 * a, home.exp_avg.data_ptr() + 4 * a,
                 home.exp_avg_sq.data_ptr() + 4 * a, b - a, float(lr), float(b1), float(b2), float(eps), float(wd),
                 home.step_count, 1.0, st)
        home.steps.add_(1.0)
        self.stash = grads
        for p in home.params:   # the stock step() that follows finds nothing to update
            p.grad = None
        return True


def _find_plan(optimizer):
    if type(optimizer) is not torch.optim.Adam or _HOMES is None:
        return False
    ids = {id(p) for g in optimizer.param_groups for p in g["params"] if p.requires_grad}
    for home in list(_HOMES):
        if ids == set(home.index):
            return _AdamPlan(optimizer, home)
    return None  # (not yet: the engine that re-homes the parameters may be created later)


def _adam_pre_hook(optimizer, args, kwargs):
    try:
        # (torch hands the hook the step's full argument tuple, the optimizer itself first; anything beyond it - a
        # closure - is the stock path's business)
        if (len(args) > 1 or kwargs or type(optimizer) is not torch.optim.Adam
                or _env("GEOSSL_NO_FUSED_ADAM")):
            return None
        plan = optimizer.__dict__.get("_geossl_plan")
        if plan is None or plan is False or not plan.home.intact(full=True):
            plan = _find_plan(optimizer)
            optimizer.__dict__["_geossl_plan"] = plan
        if plan:
            plan.stash = None
            plan.try_step(optimizer)
    except Exception as e:  # never in the way of a step: the stock path runs on whatever is still there
        import warnings
        warnings.warn("geossl_amd: fused Adam hook disabled for this optimizer (%s: %s)" % (type(e).__name__, e))
        optimizer.__dict__["_geossl_plan"] = False
    return None


def _adam_post_hook(optimizer, args, kwargs):
    # the step's backward has run by now: autograd's thread switch goes back to the caller's setting (pretrain_GeoSSL)
    try:
        from .pretrain_GeoSSL import _MT_PENDING, _restore_backward_threads
        if _MT_PENDING:
            _restore_backward_threads()
    except Exception:
        pass
    plan = optimizer.__dict__.get("_geossl_plan")
    if plan and plan.stash is not None:
        for p, g in zip(plan.home.params, plan.stash):
            p.grad = g
        plan.stash = None


def install_adam_hook():
    if not _HOOKED[0]:
        from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
        register_optimizer_step_pre_hook(_adam_pre_hook)
        register_optimizer_step_post_hook(_adam_post_hook)
        _HOOKED[0] = True


def cosine_annealing_lr(base_lr, epoch, T_max, eta_min=0.0):
    """torch.optim.lr_scheduler.CosineAnnealingLR closed form (pretrain_GeoSSL.py:350, stepped per epoch)."""
    import math
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / T_max)) / 2
