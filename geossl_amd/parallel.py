"""Data parallelism for the DDM step: one process per GPU, whole molecules sharded across ranks,
ONE all-reduce over the flat gradient buffer per step (RCCL over xGMI when the backend is "nccl";
the same code runs on gloo/CPU tensors for the world_size-2 tests).

Equal per-rank molecule counts make the mean of per-rank batch-mean losses equal the global batch
mean (NCSN.py:210-212), so N ranks x B molecules is numerically a B*N reference step up to
summation order.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # a process group is created for world > 1, and for a single rank when a backend is named explicitly
    # (GEOSSL_DIST_BACKEND / backend=): the 1-rank RCCL group is how a 1-GPU box exercises the real transport
    explicit = backend or os.environ.get("GEOSSL_DIST_BACKEND")
    if (world > 1 or explicit) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GEOSSL_DIST_BACKEND=gloo: the N>1 code path on a box with fewer GPUs than ranks (ranks then share
            # devices, see local_device) - a test aid, RCCL ("nccl") is the production transport
            backend = os.environ.get("GEOSSL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def local_device(local_rank):
    """Device index of this rank: LOCAL_RANK, wrapped only when ranks outnumber the visible GPUs (gloo test runs)."""
    return local_rank % max(1, torch.cuda.device_count())


def shard_molecules(num_mols, rank, world):
    """Contiguous, equal-sized shard [lo, hi) of whole molecules for this rank (remainder dropped,
    like a drop_last loader)."""
    per = num_mols // world
    return rank * per, (rank + 1) * per


def shard_batch_numpy(b, rank, world):
    """Slice a collated numpy batch (geossl_amd.synthetic.make_batch layout) to this rank's
    molecules and renumber: batch from 0, index tensors offset by the shard's first atom — the rule
    of dataloaders_AtomTuple.py:64-65 applied per rank."""
    sizes = np.asarray(b["sizes"])
    lo, hi = shard_molecules(len(sizes), rank, world)
    aoff = np.concatenate([[0], np.cumsum(sizes)])
    a0, a1 = int(aoff[lo]), int(aoff[hi])
    out = {"x": b["x"][a0:a1], "positions": b["positions"][a0:a1], "batch": b["batch"][a0:a1] - lo,
           "sizes": sizes[lo:hi]}
    for key in ("super_edge_index", "radius_edge_index"):
        if key in b:
            e = b[key]
            sel = (e[0] >= a0) & (e[0] < a1)
            out[key] = e[:, sel] - a0
    return out


class GradAllReduce:
    """Sum-all-reduce of one flat gradient buffer, averaged over ranks."""

    def __init__(self, flat_grad, world=None, async_op=False):
        self.buf = flat_grad
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # with an initialised group the collective is always issued, also for a single rank (a 1-rank sum is the
        # identity, bit for bit): the same code path as N ranks
        self.active = dist.is_initialized() or self.world > 1

    def __call__(self):
        if self.active:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
            return 1.0 / self.world  # folded into the optimizer's grad_scale: no extra pass over the buffer
        return 1.0
