#!/usr/bin/env python3
"""Per-dispatch durations of the kernels whose name contains <substr>, from a rocprofv3 results .db (--kernel-trace).
usage: rocprof_calls.py <results.db> <substr> [last N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in tabs else None
    if not view:
        print("no 'kernels' view; tables:", tabs)
        return
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
    rows = list(db.execute("select name, start, end from %s where name like ? order by start" % view, ("%" + sub + "%",)))
    for name, s, e in rows[-last:]:
        print("%-60s %10.1f us" % (name[:60], (e - s) / 1e3))


if __name__ == "__main__":
    main()
