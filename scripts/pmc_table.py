#!/usr/bin/env python3
"""One-line-per-kernel table from the rocprofv3 --pmc CSVs that scripts/pmc.sh collected.

usage: pmc_table.py <pmc dir> [out.txt]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "bv::" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "").replace("bv::", "")
                agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    m = lambda k, c: (sum(agg[k][c]) / len(agg[k][c])) if agg[k].get(c) else 0.0
    lines = ["PMC counters per kernel (rocprofv3 --pmc, separate passes, BVGPU_OVERLAP=0, C2 workload; mean per dispatch).",
             "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB (TCC_EA0 read / write requests); shown here in MB, NOT corrected:",
             "the guide's x2 correction for FETCH_SIZE is calibrated for wide coalesced 16 B/lane streams, these kernels mostly issue",
             "scattered 4..16-byte accesses (MI355X_MICROARCH.md, HBM section).  Read them as ratios against the algorithmic bytes.",
             "Calibration inside this very run: k_rebase streams 10 000 001 int64 in and out (76.3 MiB each, coalesced 8 B/lane):",
             "WRITE_SIZE reports it exactly, FETCH_SIZE reports half -- the guide's x2 holds for coalesced reads, writes need no correction.",
             "",
             "%-28s %9s %9s %7s %7s %7s %7s %12s %12s" % ("kernel", "FETCH_MB", "WRITE_MB", "L2hit%", "wait%", "issue%", "valu%", "VALU insts", "LDS insts")]
    for k in sorted(agg):
        hit, miss = m(k, "TCC_HIT"), m(k, "TCC_MISS")
        wc = max(m(k, "SQ_WAVE_CYCLES"), 1.0)
        lines.append("%-28s %9.1f %9.1f %7.0f %7.0f %7.0f %7.0f %12.3g %12.3g" % (
            k[:28], m(k, "FETCH_SIZE") / 1024, m(k, "WRITE_SIZE") / 1024, 100 * hit / max(hit + miss, 1), 100 * m(k, "SQ_WAIT_ANY") / wc,
            100 * m(k, "SQ_ACTIVE_INST_ANY") / wc, 100 * m(k, "SQ_ACTIVE_INST_VALU") / wc, m(k, "SQ_INSTS_VALU"), m(k, "SQ_INSTS_LDS")))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
