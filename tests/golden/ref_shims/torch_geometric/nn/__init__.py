"""Stand-ins for torch_geometric.nn.{MessagePassing, radius_graph} (schnet.py:12)."""
import numpy as np
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add"):
        super().__init__()
        assert aggr == "add"

    def propagate(self, edge_index, x, W):
        msg = self.message(x_j=x[edge_index[0]], W=W)
        out = torch.zeros(x.size(0), msg.size(1), dtype=msg.dtype, device=msg.device)
        return out.index_add(0, edge_index[1], msg)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    assert flow == "source_to_target"
    pos = x.detach().cpu().numpy().astype(np.float32)
    n = pos.shape[0]
    b = np.zeros(n, np.int64) if batch is None else batch.detach().cpu().numpy()
    r2 = np.float32(float(r) * float(r))
    cap = max_num_neighbors if loop else max_num_neighbors + 1
    src, dst = [], []
    for i in range(n):
        found = 0
        for j in range(n):
            if b[j] != b[i]:
                continue
            d = pos[j] - pos[i]
            d2 = np.float32(np.float32(d[0] * d[0]) + np.float32(d[1] * d[1]))
            d2 = np.float32(d2 + np.float32(d[2] * d[2]))
            if d2 < r2:
                found += 1
                if loop or j != i:
                    src.append(j)
                    dst.append(i)
                if found >= cap:
                    break
    return torch.tensor(np.array([src, dst], dtype=np.int64).reshape(2, -1), device=x.device)
