#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (CSV output):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/prof_step.py 4
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/prof_step.py 4
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv \
        > profiles/rNN_hbm_traffic_pmc.json

Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KB (x1024); FETCH_SIZE is doubled on
gfx950 (it reports half of wide coalesced reads); WRITE_SIZE is taken as is.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    if name.startswith("at::native::") or name.startswith("rocprim::"):
        name = name[:110]
    return name.strip()


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = short(row["Kernel_Name"])
        tot[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    """argv: fetch.csv write.csv [steps] [molecules_per_step] [workload id] [git head]
    (the last four are written into the summary: bench.py only reports these figures for the same workload)."""
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    mols = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    workload = sys.argv[5] if len(sys.argv) > 5 else "schnet/ddm-step/mols=1024/set=A/cutoff=5"
    head = sys.argv[6] if len(sys.argv) > 6 else None
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/prof_step.py (eager DDM "
                    "steps, both views). Units KB->bytes (x1024); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 "
                    "reports half of wide coalesced reads); WRITE_SIZE uncorrected.",
           "workload": workload, "steps": steps, "molecules_per_step": mols, "git_head": head,
           "csrc_sha": __import__("geossl_amd.build", fromlist=["source_hash"]).source_hash(),
           "kernels": {}}
    for k in f:
        out["kernels"][k] = {"launches": f[k][1], "fetch_bytes_per_launch": 2.0 * 1024.0 * f[k][0],
                             "write_bytes_per_launch": 1024.0 * w.get(k, (0.0, 0))[0]}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
