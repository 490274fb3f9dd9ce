"""CPU tests added in round 6: the bench line the driver parses (compact, strict JSON, bounded size whatever the run
measured)."""
import json
import os
import sys

import pytest

from conftest import REPO

sys.path.insert(0, REPO)


def _dummy_detail(n_secondary=40, prose=2000):
    """A detail object shaped like bench.py's, with paragraph-long strings and awkward numbers in every place one has
    ever appeared."""
    long = "x" * prose
    sec = {"trainer/some/very/long/secondary/name/number=%d/distinct" % i:
           {"value": 123456.789012345 + i, "unit": "molecules/s", "ms_per_step": 1.234567890123, "steps": 480, "warmup": 0,
            "workload": long, "execution": long, "final_loss": float("nan"), "p50_ms": 1.2, "graphs": 1, "captures": 1,
            "roofline": {"kernel": "k", "peak_note": long}} for i in range(n_secondary)}
    sec["broken"] = {"error": "RuntimeError: " + long}
    return {
        "metric": "molecules/s/GPU SchNet+DDM fwd+bwd (QM9-sized, bs=1024); % HBM roofline", "value": 411111.123456789,
        "unit": "molecules/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 2.4912345678, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": long, "api": "trainer", "parallelism": "dp1", "backend": None, "world_size_initialised": 1,
                   "arithmetic": long, "execution": long,
                   "product_bits": {"filter_fwd": 22, "filter_bwd": 22, "atom_row_chains": 22, "weight_gradients": 22,
                                    "ncsn_head_fwd": 24, "ncsn_head_bwd": 22, "accumulate": "fp32"}},
        "roofline": {"kernel": "geossl_cfconv_filter_bwd", "bound": "mfma", "unit": "TFLOP/s", "traffic": 1311658920.0,
                     "traffic_from": long, "avg_launch_ms": 0.81456338763237, "launches_per_step": 1.0, "timing": long,
                     "achieved": 567.2296786908015, "peak": 2500.0, "frac": 0.2268918714763206, "peak_note": long,
                     "hbm_frac": float("inf")},
        "step_roofline": {"measured_MB_per_mol": 8.89, "measured_hbm_frac": 0.44, "measured_from": long,
                          "model_hbm_frac": 1.11, "model_fp32_frac": 1.37, "alg_MB_per_mol": 22.4, "alg_MFLOP_per_mol": 542.4},
        "step_ms_percentiles": {"p10": 2.5477, "p50": 2.5744, "p90": 2.6094},
        "kernel_ms": {"k%d" % i: {"avg_ms": 0.1, "per_step": 1.0} for i in range(10)},
        "final_loss": 55.5, "secondary": sec, "value_24bit": 286000.123456,
        "parity_vs_oracle": {"molecules": 1024, "oracle_loss": 82.8,
                             "22bit": {"loss": 82.8, "loss_rel_err": 1.9e-7, "worst_grad_rel_err": 7.7e-6, "worst_grad": "model.x"},
                             "24bit": {"loss": 82.8, "loss_rel_err": 1.0e-7, "worst_grad_rel_err": 2.0e-6, "worst_grad": "model.y"}},
        "cpu_baseline": {"value": 143.3, "unit": "molecules/s", "cores": 32, "kind": "port", "sample": long,
                         "sample_short": "oracle DDM step"},
    }


def _strict(text):
    def bad(c):
        raise ValueError("non-finite constant %r in the line" % c)
    return json.loads(text, parse_constant=bad)


def test_bench_line_is_compact_strict_json():
    """BENCH_r05.parsed was null because the one line had grown to 24 KB.  The line is now built by
    bench.compact_line from the detail object: under 8 KB (4 KB for a run of today's size) and strict JSON - no NaN /
    Infinity tokens - with every key the driver's contract names, `roofline` and `cpu_baseline`."""
    import bench
    text = bench.compact_line(_dummy_detail(n_secondary=26, prose=3000))
    assert "\n" not in text and len(text) < bench.LINE_TARGET + 1024, len(text)
    line = _strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_24bit", "parity_vs_oracle"):
        assert k in line, k
    assert line["config"]["workload"] and len(line["config"]["workload"]) <= 160 and "model" not in line["config"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["roofline"]["hbm_frac"] is None                      # the inf of the dummy
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    assert all(set(v) <= {"value", "ms_per_step", "steps", "error", "vs"} for v in line["secondary"].values())
    assert line["secondary"]["broken"]["error"].startswith("RuntimeError")
    assert abs(line["value"] - 411111.123456789) < 1.0


def test_bench_line_stays_under_the_limit_whatever_the_run_holds():
    import bench
    text = bench.compact_line(_dummy_detail(n_secondary=400, prose=50000))
    assert len(text) < bench.LINE_LIMIT
    line = _strict(text)
    assert line["secondary_truncated"] is True and line["value"] > 0 and line["roofline"] and line["cpu_baseline"]


def test_bench_emit_prints_the_compact_line_last(capsys, tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    bench.emit(_dummy_detail(n_secondary=3, prose=100))
    out, err = capsys.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_TARGET
    line = _strict(lines[0])
    detail = json.load(open(os.path.join(str(tmp_path), line["detail"])))
    assert detail["config"]["execution"].startswith("x") and err.startswith("bench detail: {")


def test_device_loader_visits_molecules_in_the_order_of_torchs_shuffled_loader():
    """DeviceLoader(shuffle=True) draws its permutation the way torch.utils.data.RandomSampler does (the sampler behind
    DataLoaderAtomTuple(dataset, batch_size, shuffle=True), dataloaders_AtomTuple.py:81-88): under the same global seed
    the same molecules meet in the same batches; the last short batch is kept or dropped like DataLoader's."""
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    from geossl_amd.Geom3D.dataloaders.device_dataset import DeviceLoader

    class Stub:   # (only len() is needed to draw an order)
        def __len__(self):
            return 103

    for drop_last in (False, True):
        torch.manual_seed(123)
        want = [list(b) for b in DataLoader(list(range(103)), batch_size=10, shuffle=True, drop_last=drop_last,
                                            collate_fn=lambda items: items)]
        torch.manual_seed(123)
        ld = DeviceLoader(Stub(), batch_size=10, shuffle=True, drop_last=drop_last)
        order = ld.order()
        got = [order[k * 10:(k + 1) * 10].tolist() for k in range(len(ld))]
        assert got == want and len(ld) == len(want)
    want = [list(b) for b in DataLoader(list(range(103)), batch_size=10, shuffle=True, collate_fn=lambda items: items,
                                        generator=torch.Generator().manual_seed(7))]
    ld = DeviceLoader(Stub(), batch_size=10, shuffle=True, generator=torch.Generator().manual_seed(7))
    order = ld.order()
    assert [order[k * 10:(k + 1) * 10].tolist() for k in range(len(ld))] == want
    assert DeviceLoader(Stub(), batch_size=10, shuffle=False).order().tolist() == list(range(103))


def test_dataset_handle_counts_and_fingerprint_match_the_collated_batch():
    """A DatasetBatch is ids + host sizes; what the step graphs ask of it without touching tensors - atom / super-edge /
    edge counts and the structure fingerprint - equals what the collated batch of the same molecules gives (host logic
    only: a stub in place of the device-resident arrays)."""
    import types
    import numpy as np
    import torch
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.Geom3D.dataloaders.device_dataset import DatasetBatch
    from geossl_amd.synthetic import collate_subset, make_batch
    for option in ("combination", "permutation"):
        pool = make_batch(40, seed=3, mode="B", option=option)
        sizes = np.asarray(pool["sizes"], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(sizes)])
        edge_cnt = sizes * 3
        ds = types.SimpleNamespace(sizes=sizes, off=off, pairs=sizes * (sizes - 1) // 2, option=option, x_cols=2,
                                   device=torch.device("cpu"), edges=object(), edge_cnt=edge_cnt,
                                   edge_off=np.concatenate([[0], np.cumsum(edge_cnt)]))
        ds.__len__ = lambda: 40
        ds = type("Stub", (), dict(vars(ds), __len__=lambda self: 40))()
        ids = np.random.default_rng(1).permutation(40)[:12]
        hb = DatasetBatch(ds, ids)
        raw = collate_subset(pool, ids, option=option)
        bt = pg.Batch.from_numpy(raw, "cpu", prepare=False)
        assert hb.num_graphs == 12 and hb.n_atoms == raw["x"].shape[0] and hb.n_super == raw["super_edge_index"].shape[1]
        assert hb.n_edges == int(edge_cnt[ids].sum()) and list(hb._sizes) == list(raw["sizes"])
        assert hb.fingerprint() == pg.structure_fingerprint(bt, "schnet") == pg.structure_fingerprint(hb, "schnet")
    with pytest.raises(IndexError):
        DatasetBatch(ds, [40])
    with pytest.raises(ValueError):
        DatasetBatch(ds, [])


def test_matrix_pipe_painn_kernels_hold_no_packed_fp32_arithmetic():
    """painn_mma.hip is built without packed fp32 ops (build.py SOURCE_FLAGS): `v_pk_mul_f32 .. op_sel:[0,1]`, which the
    compiler had chosen for the mu-zero form of k_painn_fwd_mma, dropped low results in lanes 48-63 on MI355X with two
    waves per SIMD.  The check reads the code object of the built library (tools/scan_packed_opsel.py)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import scan_packed_opsel as sp
    from geossl_amd import _lib
    if not os.path.exists(sp.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    table = sp.scan(_lib.LIB_PATH)
    assert any("k_schnet" in k or "k_filter" in k for k in table), "the scan sees the library's kernels"
    mma = {k: c for k, c in table.items() if "k_painn_fwd_mma" in k}
    assert not mma, {k: dict(c) for k, c in mma.items()}


def test_weight_gradient_plan_grows_chunks_only_while_every_cu_keeps_a_block():
    """geossl_tn_plan (host code of the library): about 1024 row chunks per launch; chunks grow towards 256 rows while the
    launch still has 256 blocks - the reference's batch size with 18 problems gets 18 chunks of 256 rows per problem, the
    one-problem launches of the tape keep their 72 chunks of 64, the bench size is untouched."""
    import ctypes
    from geossl_amd import _lib
    lib = _lib.load()

    def plan(R, nprob):
        c, b = ctypes.c_int(), ctypes.c_int()
        lib.geossl_tn_plan(R, nprob, ctypes.byref(c), ctypes.byref(b))
        assert c.value % 64 == 0 and (b.value - 1) * c.value < R <= b.value * c.value
        return c.value, b.value

    assert plan(4608, 18) == (256, 18)
    assert plan(4608, 1) == (64, 72)
    assert plan(36864, 18) == (704, 53)
    assert plan(36864, 1) == (128, 288)
    assert plan(100, 1) == (64, 2)


def test_step_loss_backward_takes_the_engine_whenever_the_direct_path_declines():
    """_StepLoss (the loss of do_DDM's graph path): a plain backward() asks the step's engine to set the gradients itself
    and goes through autograd when that declines or when backward() is given any argument; operations on it return plain
    tensors."""
    import torch
    from geossl_amd.pretrain_GeoSSL import _StepLoss

    class Engine:
        def __init__(self, accept):
            self.accept, self.asked = accept, 0

        def direct_backward(self, ticket):
            self.asked += 1
            return self.accept

    def make(engine):
        w = torch.ones(3, requires_grad=True)
        loss = (w * 2.0).sum().as_subclass(_StepLoss)
        loss._geossl_step = (engine, {})
        return w, loss

    eng = Engine(True)
    w, loss = make(eng)
    assert type(loss * 2) is torch.Tensor and type(loss.detach()) is torch.Tensor and loss.item() == 6.0
    loss.backward()
    assert eng.asked == 1 and w.grad is None              # the engine of the step took it: autograd did not run
    eng = Engine(False)
    w, loss = make(eng)
    loss.backward()
    assert eng.asked == 1 and torch.equal(w.grad, torch.full((3,), 2.0))
    eng = Engine(True)
    w, loss = make(eng)
    loss.backward(retain_graph=True)                       # any argument: autograd
    assert eng.asked == 0 and torch.equal(w.grad, torch.full((3,), 2.0))
    eng = Engine(True)
    w, loss = make(eng)
    (g,) = torch.autograd.grad(loss, [w])
    assert eng.asked == 0 and torch.equal(g, torch.full((3,), 2.0))
