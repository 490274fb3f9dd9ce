"""PaiNN DDM step time by interaction-kernel form: python tools/experiments/painn_forms.py <set> <mols>
(env GEOSSL_PAINN_MMA_CAP / GEOSSL_PAINN_VECTOR / GEOSSL_PAINN_PER_ATOM select the form; one process per form)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    molset, mols = sys.argv[1], int(sys.argv[2])
    from geossl_amd import _lib
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    wl = bench.Workload(dev, 0, 1, model="painn", mols=mols, molset=molset, n_batches=8, from_pool=True)
    for i in range(24):
        wl.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 96 if mols <= 256 else 32
    for i in range(n):
        wl.step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("set %s mols %d %s: %.4f ms/step (%d captures)" % (
        molset, mols, {k: v for k, v in os.environ.items() if k.startswith("GEOSSL_PAINN")}, 1e3 * el / n,
        wl.trainer.step_graphs.captures), flush=True)


if __name__ == "__main__":
    main()
