"""Index/byte side of the oracle: radius graph, super-edges, collation (numpy, integer exact).

Test infrastructure only — see oracle/__init__.py.
"""
import itertools

import numpy as np


def pair_dist2_f32(pi, pj):
    """fl32(fl32(fl32(dx*dx)+fl32(dy*dy))+fl32(dz*dz)) with d = pj - pi, no fused multiply-add.

    Restates the accumulation loop of torch_cluster's radius kernel (third-party, absent;
    reached from schnet.py:91) — parity unpinned, see package docstring.
    """
    d = (pj.astype(np.float32) - pi.astype(np.float32)).astype(np.float32)
    d2 = (d[..., 0] * d[..., 0]).astype(np.float32) + (d[..., 1] * d[..., 1]).astype(np.float32)
    return (d2.astype(np.float32) + (d[..., 2] * d[..., 2]).astype(np.float32)).astype(np.float32)


def radius_graph_np(pos, r, batch=None, max_num_neighbors=32, loop=False):
    """radius_graph(pos, r, batch) as called at schnet.py:91 and datasets_3D_Radius.py:120.

    Returns int64 [2, E] = [source j; target i], target-major, sources ascending (canonical
    order).  Cap: scan stops after max_num_neighbors(+1 when loop=False) hits,
    the self hit included, then the self edge is dropped.
    """
    pos = np.asarray(pos, dtype=np.float32)
    n = pos.shape[0]
    b = np.zeros(n, np.int64) if batch is None else np.asarray(batch, dtype=np.int64)
    r2 = np.float32(float(r) * float(r))
    cap = max_num_neighbors if loop else max_num_neighbors + 1
    src, dst = [], []
    # graphs are contiguous runs in a collated batch, but do not assume it
    order = np.arange(n)
    for g in np.unique(b):
        idx = order[b == g]
        p = pos[idx]
        d2 = pair_dist2_f32(p[:, None, :], p[None, :, :])  # [target i, source j]
        hit = d2 < r2
        for ti in range(len(idx)):
            js = np.nonzero(hit[ti])[0][:cap]
            if not loop:
                js = js[js != ti]
            src.extend(idx[js].tolist())
            dst.extend([idx[ti]] * len(js))
    e = np.array([src, dst], dtype=np.int64).reshape(2, -1)
    # target-major over the whole batch (graphs may be interleaved in the general case)
    key = np.lexsort((e[0], e[1]))
    return e[:, key]


def super_edges_np(n, option="combination"):
    """AtomTupleExtractor.__call__ with ratio=1 (dataloaders_AtomTuple.py:15-37)."""
    if n < 2:
        return np.empty((2, 0), dtype=np.int64)
    if option == "permutation":
        p = list(itertools.permutations(range(n), 2))
    else:
        p = list(itertools.combinations(range(n), 2))
    return np.array(p, dtype=np.int64).T


def collate_np(mols, option="combination", radius=None):
    """BatchAtomTuple.from_data_list (dataloaders_AtomTuple.py:46-73).

    mols: list of (x[n,2] int64, positions[n,3] f32).  Index keys are offset by the cumulative
    node count (64-65); batch = full((n_i,), i) (61).  radius -> per-molecule
    radius_edge_index on the unperturbed positions (datasets_3D_Radius.py:120).
    """
    xs, ps, bs, ses, res = [], [], [], [], []
    cum = 0
    for i, (x, p) in enumerate(mols):
        n = x.shape[0]
        xs.append(x)
        ps.append(p.astype(np.float32))
        bs.append(np.full((n,), i, dtype=np.int64))
        ses.append(super_edges_np(n, option) + cum)
        if radius is not None:
            res.append(radius_graph_np(p, radius) + cum)
        cum += n
    out = {
        "x": np.concatenate(xs, 0),
        "positions": np.concatenate(ps, 0),
        "batch": np.concatenate(bs, 0),
        "super_edge_index": np.concatenate(ses, 1),
    }
    if radius is not None:
        out["radius_edge_index"] = np.concatenate(res, 1)
    return out
