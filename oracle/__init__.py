"""CPU oracle for the GeoSSL SchNet/PaiNN + DDM hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a plain restatement (numpy for the
integer/index work, fp32 torch-CPU for the network arithmetic) of the reference algorithm, each
function citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it, and only as the checker — the product package
`geossl_amd` never imports it and has no CPU fallback.

Pinning: the network arithmetic is pinned against golden vectors produced by importing the
unmodified reference (`/root/reference`) in the build container (`tests/golden/make_golden.py`,
fixtures in `tests/golden/*.npz`).  The third-party boundary (`torch_cluster.radius_graph`,
`torch_scatter.scatter`, `MessagePassing.propagate`) is absent from the reference tree and from the
container, and the reference has no test that pins it: at that boundary the oracle restates the
published semantics from memory — **parity unpinned** there.
"""
