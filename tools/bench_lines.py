"""Run chosen secondary lines of bench.py on their own (same process, one JSON object per line):
    python tools/bench_lines.py trainer/painn trainer/painn/distinct
    python tools/bench_lines.py --list"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    names = sys.argv[1:]
    if not names or names == ["--list"]:
        print("\n".join(bench.SECONDARY_LINES))
        return
    from geossl_amd import _lib
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    for n in names:
        out[n] = bench.run_secondary(n, dev, 0, 1)
        r = out[n]
        print(json.dumps({n: {k: r[k] for k in r if k not in ("workload", "execution")}}), flush=True)
    for a, b in bench.SECONDARY_RATIOS:
        if a in out and b in out and "value" in out[a] and "value" in out[b]:
            print("%s / %s = %.3f" % (a, b, out[a]["value"] / out[b]["value"]))


if __name__ == "__main__":
    main()
