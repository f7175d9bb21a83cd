#!/usr/bin/env python3
"""Prints the kernel timeline of one scan recorded in a rocprofv3 results .db (--kernel-trace): the last one, or the K-th
from the end (--back K: scripts/ab_time.py ends with three serialised profiling scans, so its last overlapped scan is --back 3).

usage: timeline.py <results.db> [out.txt] [--back K]
"""
import sqlite3
import sys


def main():
    back = 0
    if "--back" in sys.argv:
        i = sys.argv.index("--back")
        back = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kt[0])]
    rows = list(db.execute("select name, start, end, stream_id from %s order by start" % kt[0])) if "stream_id" in cols else [r + (0,) for r in db.execute("select name, start, end from %s order by start" % kt[0])]
    # last scan = from the last k_headers on
    hs = [i for i, r in enumerate(rows) if "k_headers" in r[0]]
    rows = rows[hs[-1 - back]:(hs[-back] if back else len(rows))]
    t0 = rows[0][1]
    out = []
    for name, st, en, sid in rows:
        short = name.split("(")[0].replace("void ", "").replace("bv::", "")
        out.append("%9.1f %9.1f %8.1f  s%-3s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, sid, short[:60]))
    text = "start_us    end_us   dur_us  stream kernel\n" + "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
