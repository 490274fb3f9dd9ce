"""Stand-ins for torch_scatter.{scatter, scatter_add} (schnet.py:13, painn.py:9, NCSN.py:6)."""
import torch


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim in (0, -1) and index.dim() == 1
    if dim == -1:
        assert src.dim() == 1
    size = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[0] = size
    return torch.zeros(shape, dtype=src.dtype, device=src.device).index_add(0, index, src)


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    assert reduce == "mean"
    s = scatter_add(src, index, dim, out, dim_size)
    cnt = scatter_add(torch.ones_like(index, dtype=src.dtype), index, 0, None, s.size(0)).clamp(min=1)
    return s / cnt.view(-1, *([1] * (s.dim() - 1)))
