"""CPU-only tests of the host side: C-ABI surface, module/parameter mirror of the reference interface,
error behaviour, synthetic generator, multi-process gradient all-reduce (gloo, world_size 2)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, load_golden


def test_c_abi_exports_every_declared_symbol():
    from geossl_amd import _lib, build
    build.build(verbose=False)
    h = open(os.path.join(REPO, "include", "geossl_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    declared = set(re.findall(r"\b(?:int|int64_t|void)\s+(geossl_\w+)\s*\(", h))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (geossl_\w+)", out))
    assert declared <= exported, declared - exported
    lib = _lib.load()  # binds argtypes for every prototype; no compute call without a GPU
    assert lib.geossl_abi_version() == 1
    # host-only planning helpers are callable on CPU
    import ctypes as C
    chunk, nblk = C.c_int(), C.c_int()
    lib.geossl_tn_plan(36864, 20, C.byref(chunk), C.byref(nblk))
    assert chunk.value % 64 == 0 and chunk.value * nblk.value >= 36864
    assert lib.geossl_tn_workspace_floats(36864, 128, 128, 20) == 20 * nblk.value * (128 * 128 + 256)


def test_header_argument_counts_match_ctypes():
    from geossl_amd import _lib
    h = open(os.path.join(REPO, "include", "geossl_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    for ret, name, args in re.findall(r"\b(int|int64_t|void)\s+(geossl_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        a = [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]
        assert len(a) == len(_lib.PROTOTYPES[name][1]), name


def test_state_dict_mirrors_reference():
    from geossl_amd.Geom3D.models import PaiNN, SchNet
    from geossl_amd.NCSN import NCSN_version_03
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    mods = {
        "SchNet": SchNet(hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=51, cutoff=10.0,
                         node_class=9, readout="mean"),
        "PaiNN": PaiNN(n_atom_basis=128, n_interactions=3, n_rbf=20, cutoff=5.0, max_z=9, n_out=1, readout="add"),
        "NCSN_version_03": NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2),
    }
    for name, m in mods.items():
        got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert got == spec[name]["state_dict"], name
        gotp = [[k, list(p.shape), bool(p.requires_grad)] for k, p in m.named_parameters()]
        assert gotp == spec[name]["named_parameters"], name
        assert sum(p.numel() for p in m.parameters()) == spec[name]["num_params"]


def test_schnet_init_quirks():
    from geossl_amd.Geom3D.models import SchNet
    torch.manual_seed(0)
    s = SchNet(node_class=9)
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["SchNet_init"]
    blk = s.interactions[0]
    assert blk.mlp is blk.conv.nn  # one module under two names (schnet.py:141-148)
    assert float(blk.mlp[0].bias.abs().max()) == 0.0
    # same seed, same construction order -> bit-identical init, incl. the never-zeroed mlp[2].bias
    assert abs(float(blk.mlp[2].bias.abs().max()) - spec["mlp2_bias_absmax"]) < 1e-12
    assert s.atomic_mass.dtype == torch.float64 and s.atomic_mass.shape == (119,)
    with pytest.raises(AssertionError):
        SchNet(node_class=9, readout="max")


def test_ncsn_sigma_ladder_and_frozen_parameter():
    from geossl_amd.NCSN import NCSN_version_03
    g = load_golden("g5_ncsn_comb_K50_p2")
    m = NCSN_version_03(128, 10.0, 0.01, 50, "symmetry", 2)
    assert torch.equal(m.sigmas.data, torch.from_numpy(g["sigmas"])) and not m.sigmas.requires_grad


def test_no_cpu_fallback_and_no_oracle_in_product():
    from geossl_amd import _lib, ops
    with pytest.raises(_lib.GeosslHipError):
        ops.radius_graph(torch.zeros(4, 3), 5.0)
    pkg = os.path.join(REPO, "geossl_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "/root/reference" not in src, f


def test_painn_forward_fails_loudly():
    from geossl_amd.Geom3D.models import PaiNN
    from geossl_amd import _lib
    p = PaiNN(128, 3, 20, 5.0, 1, "add", max_z=9)
    with pytest.raises(_lib.GeosslHipError):  # CPU tensors: no fallback
        p(torch.zeros(2, dtype=torch.long), torch.zeros(2, 3), torch.zeros(2, 0, dtype=torch.long),
          torch.zeros(2, dtype=torch.long))
    assert isinstance(p.create_output_layers(), torch.nn.Sequential)


def test_install_as_reference_aliases():
    import geossl_amd
    geossl_amd.install_as_reference()
    from Geom3D.models import PaiNN, SchNet  # noqa: F401
    from NCSN import NCSN_version_03  # noqa: F401


def test_synthetic_batch_statistics():
    from geossl_amd.synthetic import draw_noise, make_batch
    from oracle.graph import radius_graph_np
    b = make_batch(256, seed=0)
    assert b["positions"].dtype == np.float32 and b["x"].shape == (256 * 18, 2)
    assert b["super_edge_index"].shape == (2, 256 * 153)
    e = radius_graph_np(b["positions"], 5.0, b["batch"])
    assert 270 < e.shape[1] / 256 < 300  # ~284 directed edges / molecule at 5 A
    p = b["positions"].reshape(256, 18, 3)
    d = np.linalg.norm(p[:, :, None] - p[:, None], axis=-1) + np.eye(18)[None] * 10
    assert d.min() >= 1.0 - 1e-5
    bb = make_batch(512, seed=1, mode="B")
    assert bb["sizes"].min() >= 2 and bb["sizes"].max() <= 33
    nz = draw_noise(b, 3)
    assert nz["pos_noise"].shape == b["positions"].shape and nz["noise_level_1"].max() < 50


def test_cosine_lr_closed_form():
    from geossl_amd.optim import cosine_annealing_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 100)
    for epoch in range(1, 6):
        opt.step()
        sch.step()
        assert abs(opt.param_groups[0]["lr"] - cosine_annealing_lr(5e-4, epoch, 100)) < 1e-12


_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, {golden!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from geossl_amd.parallel import init_distributed, shard_batch_numpy, GradAllReduce
from geossl_amd.synthetic import make_batch, draw_noise
from helpers import schnet_oracle_params, ncsn_oracle_params, t
from oracle import nets
torch.set_num_threads(2)
rank, _, world = init_distributed("gloo")
cfg = dict(hidden_channels=32, num_filters=32, num_interactions=2, num_gaussians=8, cutoff=5.0, node_class=9, readout="mean")
full = make_batch(8, seed=5, mode="B")
def grads_of(b, seed):
    nz = draw_noise(b, seed)
    Pm, P1, P2 = schnet_oracle_params(cfg), ncsn_oracle_params(32, 50), ncsn_oracle_params(32, 50, 0.9)
    loss = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
        t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]), t(nz["dist_noise_2"]),
        5.0, 2, 2)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for P in (Pm, P1, P2) for k, p in sorted(P.items()) if p.requires_grad])
    return loss.detach(), flat, nz
mine = shard_batch_numpy(full, rank, world)
assert mine["batch"].min() == 0 and len(mine["sizes"]) == 4
assert mine["super_edge_index"].min() == 0 and mine["super_edge_index"].max() == mine["x"].shape[0] - 1
loss, flat, nz = grads_of(mine, 100 + rank)
red = GradAllReduce(flat)
scale = red()
flat = flat * scale
# every rank now holds the mean of the two per-rank gradients
torch.save(dict(loss=loss, flat=flat, nz=nz, shard=mine), os.path.join({out!r}, "rank%d.pt" % rank))
dist.barrier()
"""


def test_gradient_allreduce_world2_gloo(tmp_path):
    """N>1 path on CPU: two gloo ranks shard a batch by whole molecules, run the (oracle) DDM step on
    their shard, all-reduce the flat gradient; the result equals the mean of the per-shard gradients
    computed in one process."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(repo=REPO, golden=GOLDEN, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert torch.equal(r0["flat"], r1["flat"])  # both ranks hold the same reduced buffer
    # recompute per-shard grads in this process and average
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import ncsn_oracle_params, schnet_oracle_params, t
    from oracle import nets
    cfg = dict(hidden_channels=32, num_filters=32, num_interactions=2, num_gaussians=8, cutoff=5.0, node_class=9,
               readout="mean")
    flats = []
    for r in (r0, r1):
        b, nz = r["shard"], r["nz"]
        Pm, P1, P2 = schnet_oracle_params(cfg), ncsn_oracle_params(32, 50), ncsn_oracle_params(32, 50, 0.9)
        loss = nets.do_ddm_schnet(Pm, P1, P2, t(b["x"]), t(b["positions"]), t(b["batch"]), t(b["super_edge_index"]),
                                  t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]),
                                  t(nz["noise_level_2"]), t(nz["dist_noise_2"]), 5.0, 2, 2)
        loss.backward()
        flats.append(torch.cat([p.grad.reshape(-1) for P in (Pm, P1, P2) for k, p in sorted(P.items())
                                if p.requires_grad]))
    want = (flats[0] + flats[1]) / 2
    assert float((r0["flat"] - want).abs().max() / want.abs().max()) < 1e-6


def _loader_molecules(g):
    from geossl_amd.Geom3D.dataloaders import Data
    sizes, off, mols = g["sizes"].tolist(), 0, []
    from oracle import graph
    for n in sizes:
        pos = g["positions"][off:off + n]
        mols.append(Data(x=torch.from_numpy(g["x"][off:off + n]), positions=torch.from_numpy(pos),
                         radius_edge_index=torch.from_numpy(graph.radius_graph_np(pos, 5.0))))
        off += n
    return mols


@pytest.mark.parametrize("option", ["combination", "permutation"])
@pytest.mark.parametrize("ratio", [1, 0.5])
def test_reference_loader_surface_matches_reference(option, ratio):
    """AtomTupleExtractor()(data) as a per-molecule transform + BatchAtomTuple.from_data_list
    (dataloaders_AtomTuple.py:15-37,46-78) against fixture G11 = the unmodified reference's own classes on the same
    molecules; ratio < 1 uses the same np.random.choice stream."""
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, BatchAtomTuple
    g = load_golden("g11_loader")
    np.random.seed(123)
    ext = AtomTupleExtractor(ratio=ratio, option=option)
    mols = [ext(d) for d in _loader_molecules(g)]
    assert mols[0].super_edge_index.shape == (2, 0) and mols[0].super_edge_index.dtype == torch.long  # 1-atom molecule
    bt = BatchAtomTuple.from_data_list(mols)
    tag = "%s_%g" % (option, ratio)
    assert torch.equal(bt.super_edge_index, torch.from_numpy(g["sei/" + tag]))
    assert torch.equal(bt.batch, torch.from_numpy(g["batch/" + tag]))
    assert torch.equal(bt.radius_edge_index, torch.from_numpy(g["rei/" + tag]))
    assert torch.equal(bt.x, torch.from_numpy(g["x"])) and torch.equal(bt.positions, torch.from_numpy(g["positions"]))
    assert bt.num_graphs == int(g["num_graphs/" + tag]) == len(g["sizes"])
    assert all(v.is_contiguous() for v in (bt.x, bt.positions, bt.batch, bt.super_edge_index))


def test_dataloader_atom_tuple_batches_like_the_reference():
    """DataLoaderAtomTuple(dataset, batch_size, shuffle, **kw) (dataloaders_AtomTuple.py:81-88): a torch DataLoader
    whose batches are BatchAtomTuple.from_data_list of consecutive molecules (shuffle off)."""
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, BatchAtomTuple, DataLoaderAtomTuple
    g = load_golden("g11_loader")
    ext = AtomTupleExtractor(ratio=1, option="combination")
    dataset = [ext(d) for d in _loader_molecules(g)]
    loader = DataLoaderAtomTuple(dataset, batch_size=4, shuffle=False, num_workers=0)
    batches = list(loader)
    assert [b.num_graphs for b in batches] == [4, 2]
    ref = BatchAtomTuple.from_data_list(dataset[:4])
    for k in ("x", "positions", "batch", "super_edge_index", "radius_edge_index"):
        assert torch.equal(batches[0][k], ref[k]), k
    # second batch renumbers from 0 (node offsets restart per batch, :57-66)
    assert int(batches[1].batch.min()) == 0 and int(batches[1].super_edge_index.min()) == 0
    assert isinstance(DataLoaderAtomTuple(dataset).sampler, torch.utils.data.RandomSampler)  # shuffle=True default


# ------------------------------------------------------------------------------------ step-graph fingerprints (round 3)
def _cpu_batch(b, canonical="auto"):
    from geossl_amd import pretrain_GeoSSL as pg
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    canon = pg.canonical_option(b["sizes"], b["super_edge_index"]) if canonical == "auto" else canonical
    return pg.Batch(tt(b["x"]), tt(b["positions"]), tt(b["batch"]), tt(b["super_edge_index"]), None, len(b["sizes"]),
                    b["sizes"], canon)


def test_canonical_option_recognises_the_extractors_enumerations():
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    for option in ("combination", "permutation"):
        b = make_batch(6, seed=3, mode="B", option=option)
        assert pg.canonical_option(b["sizes"], b["super_edge_index"]) == option
        sei = b["super_edge_index"].copy()
        sei[:, [0, 1]] = sei[:, [1, 0]]                      # the same tuples in another order
        assert pg.canonical_option(b["sizes"], sei) is None
        assert pg.canonical_option(b["sizes"], b["super_edge_index"][:, :-1]) is None   # a sampled subset
        assert pg.canonical_option(b["sizes"][::-1], b["super_edge_index"]) in (None, option)


def test_structure_fingerprint_is_content_derived():
    """The graph a batch replays is chosen by what the graph binds: equal molecule sizes in equal order ->
    one fingerprint whatever the tensor objects; the same atoms split differently -> another one, although N and S
    agree; index tensors that are not a function of the sizes -> identified by the tensor objects themselves."""
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import make_batch
    b1 = make_batch(2, seed=1, sizes=[10, 20])
    b2 = make_batch(2, seed=2, sizes=[10, 20])
    b3 = make_batch(2, seed=1, sizes=[20, 10])
    assert b1["x"].shape == b3["x"].shape and b1["super_edge_index"].shape == b3["super_edge_index"].shape
    f1, f2, f3 = (pg.structure_fingerprint(_cpu_batch(b)) for b in (b1, b2, b3))
    assert f1 == f2 and f1 != f3 and f1[0] == "sizes"
    # no canonical marker (e.g. tuples sampled with ratio < 1): tensor identity, two collations never share a graph
    n1, n2 = _cpu_batch(b1, canonical=None), _cpu_batch(b1, canonical=None)
    g1, g2 = pg.structure_fingerprint(n1), pg.structure_fingerprint(n2)
    assert g1[0] == "tensors" and g1 != g2 and pg.structure_fingerprint(n1) == g1
    n1.super_edge_index.add_(0)                               # an in-place edit bumps the version: a new structure
    assert pg.structure_fingerprint(n1) != g1
    # PaiNN's edge list depends on the geometry: always by identity
    assert pg.structure_fingerprint(_cpu_batch(b1), "painn")[0] == "tensors"


def test_loader_marks_canonical_batches():
    """AtomTupleExtractor(ratio=1) marks its molecules, from_data_list carries the mark (and the host sizes) into the
    batch; a sampled extractor, or a mix of options, leaves no mark."""
    from geossl_amd.Geom3D.dataloaders import AtomTupleExtractor, BatchAtomTuple
    g = load_golden("g11_loader")
    full = [AtomTupleExtractor(ratio=1, option="combination")(d) for d in _loader_molecules(g)]
    bt = BatchAtomTuple.from_data_list(full)
    assert bt._canonical == "combination" and bt._sizes == g["sizes"].tolist()
    assert "_sei_canonical" not in bt.keys and "_canonical" not in bt.keys
    np.random.seed(0)
    half = [AtomTupleExtractor(ratio=0.5, option="combination")(d) for d in _loader_molecules(g)]
    assert BatchAtomTuple.from_data_list(half)._canonical is None
    mixed = full[:2] + [AtomTupleExtractor(ratio=1, option="permutation")(d) for d in _loader_molecules(g)[2:]]
    assert BatchAtomTuple.from_data_list(mixed)._canonical is None


def test_shard_batch_numpy_keeps_painn_edges_with_their_molecules():
    """Data parallelism for PaiNN (BASELINE config 4/5): the precomputed radius_edge_index is sharded with the molecules
    it belongs to and renumbered from 0, so the shards of two ranks are exactly the collations of their own molecules."""
    from geossl_amd.parallel import shard_batch_numpy
    from geossl_amd.synthetic import make_batch
    from oracle import graph
    full = make_batch(6, seed=9, mode="B")
    off = np.concatenate([[0], np.cumsum(full["sizes"])])
    mols = [(full["x"][off[m]:off[m + 1]], full["positions"][off[m]:off[m + 1]]) for m in range(6)]
    full["radius_edge_index"] = graph.collate_np(mols, radius=5.0)["radius_edge_index"]
    for rank in range(2):
        mine = shard_batch_numpy(full, rank, 2)
        own = graph.collate_np(mols[3 * rank:3 * rank + 3], radius=5.0)
        for k in ("x", "positions", "batch", "super_edge_index", "radius_edge_index"):
            assert np.array_equal(mine[k], own[k]), (rank, k)
        assert mine["sizes"].tolist() == full["sizes"][3 * rank:3 * rank + 3].tolist()


_PAINN_WORKER = r"""
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, {golden!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from geossl_amd.parallel import init_distributed, shard_batch_numpy, GradAllReduce
from geossl_amd.synthetic import make_batch, draw_noise
from helpers import ncsn_oracle_params, t
from oracle import graph, nets
from test_oracle_golden import painn_params
torch.set_num_threads(2)
rank, _, world = init_distributed("gloo")
cfg = dict(n_atom_basis=32, n_interactions=2, n_rbf=8, cutoff=5.0, max_z=9)
full = make_batch(8, seed=6, mode="B")
off = np.concatenate([[0], np.cumsum(full["sizes"])])
mols = [(full["x"][off[m]:off[m + 1]], full["positions"][off[m]:off[m + 1]]) for m in range(8)]
full["radius_edge_index"] = graph.collate_np(mols, radius=5.0)["radius_edge_index"]
mine = shard_batch_numpy(full, rank, world)
own = graph.collate_np(mols[4 * rank:4 * rank + 4], radius=5.0)
for k in ("x", "positions", "batch", "super_edge_index", "radius_edge_index"):
    assert np.array_equal(mine[k], own[k]), k          # the shard IS the collation of this rank's molecules
nz = draw_noise(mine, 100 + rank)
Pm, P1, P2 = painn_params(cfg), ncsn_oracle_params(32, 50), ncsn_oracle_params(32, 50, 0.9)
loss = nets.do_ddm_painn(Pm, P1, P2, t(mine["x"]), t(mine["positions"]), t(mine["batch"]), t(mine["radius_edge_index"]),
    t(mine["super_edge_index"]), t(nz["pos_noise"]), t(nz["noise_level_1"]), t(nz["dist_noise_1"]), t(nz["noise_level_2"]),
    t(nz["dist_noise_2"]), 32, 2, 5.0, 2, "add")
loss.backward()
flat = torch.cat([p.grad.reshape(-1) for P in (Pm, P1, P2) for k, p in sorted(P.items()) if p.requires_grad and p.grad is not None])
local = flat.clone()
scale = GradAllReduce(flat)()
torch.save(dict(local=local, reduced=flat * scale), os.path.join({out!r}, "rank%d.pt" % rank))
dist.barrier()
"""


def test_painn_shards_world2_gloo(tmp_path):
    """BASELINE config 4 with the second backbone, on CPU: two gloo ranks shard a ragged batch WITH its precomputed
    radius_edge_index (each shard equals the collation of the rank's own molecules), run the oracle's PaiNN DDM step
    and all-reduce the flat gradient: both hold the mean of the two local gradients."""
    script = tmp_path / "worker.py"
    script.write_text(_PAINN_WORKER.format(repo=REPO, golden=GOLDEN, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert torch.equal(r0["reduced"], r1["reduced"])
    want = (r0["local"] + r1["local"]) / 2
    assert float((r0["reduced"] - want).abs().max() / want.abs().max()) < 1e-6
    assert not torch.equal(r0["local"], r1["local"])


def test_layer_loop_block_plan():
    """layout.loop_block_plan: one round of at most 512 blocks, the fewest molecules per block that fit it, at most 96
    rows per block; uniform batches only."""
    from geossl_amd.layout import loop_block_plan
    p = loop_block_plan([18] * 2048)                       # the bench batch (two views of 1024 molecules)
    assert p.shape == (512, 4) and p[0].tolist() == [0, 72, 0, 4] and p[-1].tolist() == [36792, 36864, 2044, 2048]
    p = loop_block_plan([18] * 256)                        # bs = 128: one molecule per block (fills 256 of the slots)
    assert p.shape == (256, 4) and p[3].tolist() == [54, 72, 3, 4]
    p = loop_block_plan([18] * 2060)                       # 515 blocks of four would need a second round: five per block
    assert p.shape == (412, 4) and p[-1].tolist() == [18 * 2055, 18 * 2060, 2055, 2060]
    assert (p[:, 1] - p[:, 0]).max() <= 96 and p[:, 0].tolist() == sorted(p[:, 0].tolist())
    assert (p[1:, 0] == p[:-1, 1]).all() and (p[1:, 2] == p[:-1, 3]).all()      # consecutive blocks cover the batch
    assert loop_block_plan([18] * 3000) is None            # six per block would be 108 rows
    assert loop_block_plan([18] * 10 + [17]) is None       # ragged: no plan
    assert loop_block_plan([120] * 4) is None              # a molecule larger than a block
    assert loop_block_plan([]) is None
    p = loop_block_plan([7] * 6000)                        # small molecules: 12 per block = 84 rows, 500 blocks
    assert p.shape == (500, 4) and int((p[:, 1] - p[:, 0]).max()) == 84


# ------------------------------------------------------------------------------------- capacity buckets, host side
@pytest.mark.parametrize("by_targets", [False, True])
def test_bucket_host_plan_matches_the_collated_batch(by_targets, monkeypatch):
    """bucket.host_plan: everything of a batch's index structures that is a function of the molecule sizes, against what
    the collation itself produces (synthetic.make_batch: AtomTupleExtractor's enumeration with node offsets) and the
    oracle's pair-slot order - pointer arrays of the two-view batch, se_ptr, the incidence counts behind inc_ptr, the
    divisor of NCSN.py:210-212, and an aggregation work list that covers every (molecule, part) once, largest first."""
    from geossl_amd import bucket as bk
    from geossl_amd._lib import load
    from geossl_amd.synthetic import make_batch
    lib = load()
    # by_targets: the work list of small launches - one item per atom of every molecule (molecule | atom << 24)
    monkeypatch.setenv("GEOSSL_AGG_TARGETS_MAX", "256" if by_targets else "0")
    for option in ("combination", "permutation"):
        # (trailing single atom: divisor B - 1; 34 / 60 atoms: above the size classes, one work item per atom)
        sizes = np.array([5, 33, 1, 18, 27, 2, 30, 1, 60, 9, 31, 34, 20, 1], dtype=np.int64)
        b = make_batch(len(sizes), seed=3, sizes=sizes, option=option)
        hp = bk.host_plan(sizes, option)
        N, P, S, W = hp["counts"]
        B = len(sizes)
        assert N == b["x"].shape[0] and S == b["super_edge_index"].shape[1] and P == int((sizes * (sizes - 1) // 2).sum())
        off = np.concatenate([[0], np.cumsum(sizes)])
        assert np.array_equal(hp["mol_ptr2"], np.concatenate([off, off[1:] + N]))
        poff = np.concatenate([[0], np.cumsum(sizes * (sizes - 1) // 2)])
        assert np.array_equal(hp["pair_ptr2"], np.concatenate([poff, poff[1:] + P]))
        e2g = b["batch"][b["super_edge_index"][0]]
        assert np.array_equal(np.diff(hp["se_ptr"]), np.bincount(e2g, minlength=B)) and hp["divisor"] == int(e2g.max()) + 1
        deg = np.bincount(b["super_edge_index"].reshape(-1), minlength=N)
        assert np.array_equal(np.diff(hp["inc_ptr"]), deg) and hp["inc_ptr"][0] == 0
        # eight queues of equal length (one per XCD: workgroup b takes entry b / 8 of queue b mod 8), padded with -1
        wk = hp["work"]
        assert wk.size % 8 == 0 and wk.size <= W
        queues = wk.reshape(8, -1)
        real = wk[wk != -1]
        mol, part = real & 0x00FFFFFF, (real.astype(np.int64) >> 24) & 255
        n2 = np.concatenate([sizes, sizes])
        want = sorted((m, k) for m in range(2 * B)
                      for k in range(int(n2[m]) if by_targets else lib.geossl_aggregate_parts(int(n2[m]))))
        assert sorted(zip(mol.tolist(), part.tolist())) == want
        where = {}
        for k in range(8):
            row = queues[k]
            valid = row[row != -1]
            assert np.all(row[len(valid):] == -1)                                   # padding at the end of a queue only
            qm, qp = valid & 0x00FFFFFF, (valid.astype(np.int64) >> 24) & 255
            assert np.all(np.diff(n2[qm]) <= 0)                                     # largest molecules first in a queue
            for m_ in np.unique(qm):
                assert m_ not in where                                              # a molecule's items: one queue ...
                where[m_] = k
                pos = np.nonzero(qm == m_)[0]
                assert np.array_equal(pos, np.arange(pos[0], pos[0] + len(pos)))    # ... consecutive entries ...
                assert np.array_equal(qp[pos], np.arange(len(pos)))                 # ... in part order
        loads = [(queues[k] != -1).sum() for k in range(8)]
        assert max(loads) - min(loads) <= int(max(want, key=lambda t_: t_[1])[1]) + 1   # balanced within one molecule


def test_bucket_capacities_and_eligibility():
    """Capacities cover the batch with slack, are multiples of the kernels' tiles and never shrink; only batches whose
    index tensors are a function of the sizes (host sizes + the extractor's full enumeration, molecules of <= 33 atoms)
    go through a bucket, and equal-sized molecules keep their per-structure graph."""
    import types
    from geossl_amd import bucket as bk
    from geossl_amd import pretrain_GeoSSL as pg
    caps = bk.capacities(18000, 160000, 160000, 2100, B=1024)
    assert all(c >= v * 1.03 for c, v in zip(caps, (18000, 160000, 160000, 2100)))
    assert caps[0] % 32 == 0 and caps[1] % 64 == 0 and caps[2] % 64 == 0
    grown = bk.capacities(17000, 170000, 150000, 2000, B=1024, prev=caps)
    assert all(g >= c for g, c in zip(grown, caps)) and grown[1] >= 170000 * 1.03
    small = bk.capacities(2300, 20000, 20000, 260, B=128)
    assert small[0] >= 2300 * 1.13                                                   # 1.5 / sqrt(128) of slack

    def fake(dtype, *shape):   # what bucket.tensors_ok asks of a tensor, without a GPU
        n = int(np.prod(shape))
        return types.SimpleNamespace(is_cuda=True, dtype=dtype, dim=lambda: len(shape), size=lambda i: shape[i],
                                     numel=lambda: n, is_contiguous=lambda: True, stride=lambda i: 1, _version=0)

    def batch(sizes, canon="combination", x_dtype=torch.long, cut=0, rei=True):
        n = np.asarray(sizes)
        N, P = int(n.sum()), int((n * (n - 1) // 2).sum())
        S = (P if canon != "permutation" else 2 * P) - cut
        return types.SimpleNamespace(_sizes=sizes, _canonical=canon, positions=fake(torch.float32, N, 3), x=fake(x_dtype, N, 2),
                                     batch=fake(torch.long, N), super_edge_index=fake(torch.long, 2, S),
                                     radius_edge_index=fake(torch.long, 2, 40) if rei else None)

    assert bk.eligible(batch([18, 20, 2]), "schnet") and not bk.is_uniform(batch([18, 20, 2]))
    assert bk.is_uniform(batch([18] * 5))
    assert bk.eligible(batch([18, 34]), "schnet") and bk.eligible(batch([18, 255]), "schnet")   # molecules with hydrogens
    assert not bk.eligible(batch([18, 256]), "schnet")                               # beyond the work list / the heads
    assert not bk.eligible(batch([18, 20], canon=None), "schnet")                    # sampled tuples: not a function of the sizes
    assert not bk.eligible(batch([18, 20]), "schnet", normalize=True)
    assert bk.eligible(batch([18, 20]), "painn") and not bk.eligible(batch([18, 20], rei=False), "painn")
    assert not bk.eligible(batch([18, 20]), "dimenet")
    assert not bk.eligible(batch([1, 1, 1]), "schnet")                               # no pair at all
    assert not bk.eligible(batch([18, 20], x_dtype=torch.int32), "schnet")           # the fill copies int64 by byte count
    assert not bk.eligible(batch([18, 20], cut=3), "schnet")                         # super_edge_index cut after the collation
    assert bk.max_n_class(18) == 33 and bk.max_n_class(34) == 64 and bk.max_n_class(20, prev=64) == 64
    assert bk.max_n_class(58) == 128                                                 # (a quarter of head room above 33)
    assert bk.max_n_class(18, model_3d="painn") == 22 and bk.max_n_class(60, model_3d="painn") == 96
    assert bk.max_n_class(44, model_3d="painn") == 44
    assert bk.edge_capacity(36000, 128) >= 36000 * 1.13 and bk.edge_capacity(100, 128, prev=4096) == 4096
    # routing of StepGraphs without a GPU: auto mode buckets ragged batches only when the modules allow it
    sg = pg.StepGraphs(lambda b, n: None, "schnet", modules=None)
    assert sg.bucket_key(batch([18, 20, 2])) is None                                 # (no modules known: no bucket)
    sg2 = pg.StepGraphs(lambda b, n: None, "schnet", mode="structure", modules=None)
    assert sg2.capture_now(batch([18, 20, 2]))                                       # structure mode: capture at first sight
