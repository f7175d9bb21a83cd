#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
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
                if "bv::" not in name and "k_totals" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "")
                agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("    %-32s mean/dispatch %16.1f   dispatches %d" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
