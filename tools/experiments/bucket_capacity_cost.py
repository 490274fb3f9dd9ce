"""What a capacity bucket's head room costs per step: the same 16 set-C batches (bs = 128) replayed on a bucket sized by
the first batch, and on one that was first grown by an artificially large batch (everything 1.35 x).
    python tools/experiments/bucket_capacity_cost.py [schnet|painn] [set] [cutoff]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "schnet"
    molset = sys.argv[2] if len(sys.argv) > 2 else "C"
    cutoff = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    from geossl_amd import _lib, ops
    from geossl_amd import pretrain_GeoSSL as pg
    from geossl_amd.synthetic import collate_subset, make_batch
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for grow in (False, True):
        wl = bench.Workload(dev, 0, 1, model=model, mols=128, molset=molset, cutoff=cutoff, n_batches=16, distinct=True)
        if grow:
            pool = make_batch(512, seed=5, mode=molset)
            order = np.argsort(-pool["sizes"])[:128]          # the 128 largest molecules of a pool: a batch nothing outgrows
            bt = pg.Batch.from_numpy(collate_subset(pool, order), dev, prepare=False)
            if model == "painn":
                bt.radius_edge_index = ops.radius_graph(bt.positions, 5.0, bt.batch)
            wl.trainer.step(bt, None)
        for i in range(32):
            wl.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(160):
            wl.step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        g = next(iter(wl.trainer._graphs.values()))
        bkt = g["bucket"]
        print("%s set %s grown=%s: %.4f ms/step, captures %d, caps %s max_n %d E_cap %d" % (
            model, molset, grow, 1e3 * el / 160, wl.trainer.step_graphs.captures, bkt.caps(), bkt.max_n, bkt.E_cap), flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
