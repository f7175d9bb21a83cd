#!/usr/bin/env python3
"""Sums one PMC counter of a rocprofv3 --pmc csv run over all dispatches, per kernel and in total.  usage: pmc_total.py <dir> <COUNTER>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, want = sys.argv[1], sys.argv[2]
    per = defaultdict(float)
    n = defaultdict(int)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != want:
                    continue
                k = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
                per[k] += float(row["Counter_Value"])
                n[k] += 1
    tot = sum(per.values())
    print("%s %s total %.1f (counter units; WRITE_SIZE / FETCH_SIZE: KB)" % (root, want, tot))
    for k in sorted(per, key=per.get, reverse=True)[:12]:
        print("   %14.1f  %5d dispatches  %s" % (per[k], n[k], k[:90]))


if __name__ == "__main__":
    main()
