"""Synthetic Molecule3D-shaped batches (host side, numpy).

Record layout follows the reference's data path: per molecule ``x[n,2] int64`` (column 0 = atom
type in 0..8, vocabulary of 9: Geom3D/datasets/datasets_utils.py:16,112-176) and
``positions[n,3] fp32``; a batch is the PyG-style concatenation with a sorted ``batch`` vector and
node-offset index tensors (Geom3D/dataloaders/dataloaders_AtomTuple.py:46-73).

Geometry: seeded random-tree growth — atom 0 at the origin, atom k at 1.4 Å from
a uniformly chosen earlier atom in a uniformly random direction, redrawn while it is closer than
1.0 Å to any placed atom.
"""
import numpy as np

BOND = 1.4
EXCL = 1.0


def molecule_sizes(num_mols, mode="A", rng=None, n_fixed=18):
    """Set A: n=18 fixed.  Set B: n ~ clip(round(N(18,4)), 2, 33) (n<=33 keeps clear of the
    32-neighbour cap of radius_graph).  Set C: n ~ clip(round(N(26,10)), 4, 72) (molecules with hydrogens)."""
    if mode == "A":
        return np.full(num_mols, n_fixed, dtype=np.int64)
    rng = np.random.default_rng(0) if rng is None else rng
    if mode == "C":
        # Molecule3D WITH hydrogens (datasets_Molecule3D.py:65 removeHs=False), what the reference's DDM script trains on
        # (submit_pretrain_GeoSSL_DDM.sh:3): n ~ clip(round(N(26, 10)), 4, 72) - about a quarter of the molecules above the
        # 33 atoms of the aggregation's size classes, about one in a hundred above 48; at the reference's 10 A radius
        # (config.py:114) the 32-neighbour cap of radius_graph cuts the lists of the compact ones
        return np.clip(np.rint(rng.normal(26.0, 10.0, size=num_mols)), 4, 72).astype(np.int64)
    return np.clip(np.rint(rng.normal(18.0, 4.0, size=num_mols)), 2, 33).astype(np.int64)


def grow_positions(sizes, rng):
    """Vectorised random-tree growth for all molecules at once. Returns list-free padded array
    [M, nmax, 3] fp32 (rows >= sizes[m] are zero)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    M, nmax = sizes.shape[0], int(sizes.max())
    pos = np.zeros((M, nmax, 3), dtype=np.float64)
    for k in range(1, nmax):
        todo = np.nonzero(sizes > k)[0]
        while todo.size:
            parent = rng.integers(0, k, size=todo.size)
            v = rng.normal(size=(todo.size, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            cand = pos[todo, parent] + BOND * v
            d = np.linalg.norm(pos[todo, :k] - cand[:, None, :], axis=2)
            ok = (d >= EXCL).all(axis=1)
            pos[todo[ok], k] = cand[ok]
            todo = todo[~ok]
    return pos.astype(np.float32)


def combination_pairs(n):
    """Lexicographic i<j pairs, shape [2, n(n-1)/2] (AtomTupleExtractor 'combination',
    dataloaders_AtomTuple.py:22-23)."""
    i, j = np.triu_indices(n, k=1)
    return np.stack([i, j]).astype(np.int64)


def permutation_pairs(n):
    """All ordered pairs i!=j in itertools.permutations order (dataloaders_AtomTuple.py:20)."""
    i, j = np.nonzero(~np.eye(n, dtype=bool))
    return np.stack([i, j]).astype(np.int64)


def make_molecules(num_mols, seed=0, mode="A", n_fixed=18, sizes=None):
    """A dataset of molecules without any index tensor: x [N,2] i64, positions [N,3] f32, sizes [M] i64 - the same
    molecules make_batch(num_mols, seed, mode) collates (same random stream)."""
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = molecule_sizes(num_mols, mode, rng, n_fixed)
    sizes = np.asarray(sizes, dtype=np.int64)
    padded = grow_positions(sizes, rng)
    N = int(sizes.sum())
    mask = np.arange(padded.shape[1])[None, :] < sizes[:, None]
    x = np.zeros((N, 2), dtype=np.int64)
    x[:, 0] = rng.integers(0, 9, size=N)
    return {"x": x, "positions": padded[mask], "sizes": sizes}


def make_batch(num_mols, seed=0, mode="A", option="combination", n_fixed=18, sizes=None):
    """Collated synthetic batch as a dict of numpy arrays:
    x [N,2] i64, positions [N,3] f32, batch [N] i64, super_edge_index [2,S] i64, sizes [B] i64."""
    rng = np.random.default_rng(seed)
    if sizes is None:
        sizes = molecule_sizes(num_mols, mode, rng, n_fixed)
    sizes = np.asarray(sizes, dtype=np.int64)
    padded = grow_positions(sizes, rng)
    N = int(sizes.sum())
    off = np.concatenate([[0], np.cumsum(sizes)])
    mask = np.arange(padded.shape[1])[None, :] < sizes[:, None]
    positions = padded[mask]
    x = np.zeros((N, 2), dtype=np.int64)
    x[:, 0] = rng.integers(0, 9, size=N)
    batch = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)
    cache, se = {}, []
    for m, n in enumerate(sizes.tolist()):
        if n not in cache:
            cache[n] = combination_pairs(n) if option == "combination" else permutation_pairs(n)
        se.append(cache[n] + off[m])
    sei = np.concatenate(se, axis=1) if se else np.empty((2, 0), np.int64)
    return {"x": x, "positions": positions, "batch": batch, "super_edge_index": sei, "sizes": sizes}


_SEI_CACHE = {}


def collate_subset(pool, mol_ids, option="combination"):
    """The molecules `mol_ids` of the collated pool `pool` (a make_batch dict), collated in that order - what a shuffled
    loader over a dataset of molecules hands over (dataloaders_AtomTuple.py:46-78, 81-88): same dict layout as
    make_batch."""
    sizes_all = np.asarray(pool["sizes"], dtype=np.int64)
    off_all = np.concatenate([[0], np.cumsum(sizes_all)])
    mol_ids = np.asarray(mol_ids, dtype=np.int64)
    sizes = sizes_all[mol_ids]
    off = np.concatenate([[0], np.cumsum(sizes)])
    # atom rows of the chosen molecules, in order
    rows = np.repeat(off_all[mol_ids] - off[:-1], sizes) + np.arange(int(off[-1]), dtype=np.int64)
    se = []
    for m, n in enumerate(sizes.tolist()):
        key = (option, n)
        if key not in _SEI_CACHE:
            _SEI_CACHE[key] = combination_pairs(n) if option == "combination" else permutation_pairs(n)
        se.append(_SEI_CACHE[key] + off[m])
    sei = np.concatenate(se, axis=1) if se else np.empty((2, 0), np.int64)
    return {"x": pool["x"][rows], "positions": pool["positions"][rows],
            "batch": np.repeat(np.arange(len(sizes), dtype=np.int64), sizes), "super_edge_index": sei, "sizes": sizes}


def draw_noise(batch, seed, num_noise_level=50, sigma=0.3, mu=0.0):
    """The three random draws of one DDM step per view pair, as explicit tensors:
    pos_noise ~ N(mu, sigma^2) [N,3]; per NCSN head noise_level ~ U{0..K-1} [B] and
    distance_noise ~ N(0,1) [S,1]."""
    rng = np.random.default_rng(seed)
    N, S, B = batch["positions"].shape[0], batch["super_edge_index"].shape[1], len(batch["sizes"])
    return {
        "pos_noise": (mu + sigma * rng.normal(size=(N, 3))).astype(np.float32),
        "noise_level_1": rng.integers(0, num_noise_level, size=B).astype(np.int64),
        "dist_noise_1": rng.normal(size=(S, 1)).astype(np.float32),
        "noise_level_2": rng.integers(0, num_noise_level, size=B).astype(np.int64),
        "dist_noise_2": rng.normal(size=(S, 1)).astype(np.float32),
    }
