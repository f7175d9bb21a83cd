#!/usr/bin/env python3
"""Dumps the per-kernel stats of a rocprofv3 results .db (--kernel-trace --stats) as a small text table.

usage: rocprof_summary.py <results.db> [out.txt]   (kernel names are truncated to keep the file readable)
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-72s %8s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) <= 72 else name[:69] + "..."
        lines.append("%-72s %8d %14.3f %14.3f %8.3f" % (short, calls, tot, avg, pct))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
