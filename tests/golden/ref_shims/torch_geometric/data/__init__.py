class Data(dict):
    """Attribute-dict stand-in for torch_geometric.data.Data (dataloaders_AtomTuple.py:5)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
