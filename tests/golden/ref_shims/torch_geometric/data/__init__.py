import re

import torch


class Data:
    """Stand-in for torch_geometric.data.Data (pyg 2.0.x) as far as dataloaders_AtomTuple.py uses it (:5,40-73):
    attributes as items, the ``keys`` property (names of the non-None attributes), ``__cat_dim__`` (index-like keys
    concatenate along the last dimension) and ``contiguous()``."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __cat_dim__(self, key, value):
        return -1 if bool(re.search("(index|face)", key)) else 0

    def contiguous(self):
        for k in self.keys:
            if torch.is_tensor(self[k]):
                self[k] = self[k].contiguous()
        return self
