#!/usr/bin/env python
"""Turns a rocprofv3 rocpd sqlite database (…_results.db) into the per-kernel summary table
committed under profiles/ (same columns as `rocprofv3 --stats`: calls, total, avg, min, max, %)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["# source: %s" % db, "%-96s %7s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for r in rows:
        lines.append("%-96s %7d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:96], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    lines.append("%-96s %7s %14d" % ("TOTAL", "", tot))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
