#!/usr/bin/env python3
"""Per-kernel SQ counters from one rocprofv3 PMC pass (CSV output), averaged per launch:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
        SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv \
        -d gpurun_out/sq -o sq -- python3 tools/prof_step.py 4
    python tools/pmc_sq.py gpurun_out/sq/sq_counter_collection.csv "<label>" "<git head>" k_filter_bwd k_filter_fwd ...

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 16-bit MFMA) summed over SIMDs, SQ_INSTS_VALU counts wave
instructions.  Derived columns: mfma_busy/wave = MFMA_BUSY / (4 * WAVE_CYCLES) (share of a wave's resident time its
SIMD's matrix pipe is busy, if one wave per SIMD; with two waves per SIMD the pipe's own utilisation is twice that),
wait_inst/wave, wait_any/wave, valu_active/wave."""
import csv
import sys
from collections import defaultdict


def main():
    path, label, head = sys.argv[1], sys.argv[2], sys.argv[3]
    want = sys.argv[4:]
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if want and not any(w in k for w in want):
            continue
        k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
    names = sorted({c for k in tot for c in tot[k]})
    print("# %s @ %s" % (label, head))
    print("# rocprofv3 --pmc %s --kernel-trace (one pass); averages per launch" % " ".join(names))
    for k in sorted(tot, key=lambda k_: -tot[k_].get("SQ_WAVE_CYCLES", 0.0)):
        avg = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        print("%s  (launches %d)" % (k[:110], max(cnt[k].values())))
        for c in names:
            if c in avg:
                print("    %-28s %16.0f" % (c, avg[c]))
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            d = []
            if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
                d.append("mfma_busy/(4*wave_cycles) %.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc)))
            for c, n in (("SQ_WAIT_INST_ANY", "wait_inst/wave"), ("SQ_WAIT_ANY", "wait_any/wave"),
                         ("SQ_ACTIVE_INST_VALU", "valu_active/wave"), ("SQ_ACTIVE_INST_LDS", "lds_active/wave")):
                if c in avg:
                    d.append("%s %.3f" % (n, avg[c] / wc))
            if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_ACTIVE_INST_LDS"):
                d.append("bank_conflict/lds_active %.3f" % (avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_ACTIVE_INST_LDS"]))
            print("    derived: " + ", ".join(d))


if __name__ == "__main__":
    main()
