#!/usr/bin/env python3
"""Prints the kernels behind the last marker fill of a scripts/op_timeline.py trace.  usage: op_dump.py <results.db> [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
rows = list(db.execute("select name, start, end from %s order by start" % kt[0]))
idx = [i for i, r in enumerate(rows) if "FillFunctor" in r[0] or "fill" in r[0].lower() and "at::" in r[0]]
rows = rows[idx[-1] + 1:] if idx else rows[-80:]
t0 = rows[0][1]
for name, st, en in rows:
    if (en - st) / 1e3 >= minus:
        print("%9.1f %9.1f %8.1f  %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, name.split("(")[0].replace("void ", "").replace("bv::", "")[:60]))
