#!/usr/bin/env python
"""Turns a rocprofv3 (--kernel-trace --stats) rocpd SQLite result into the text summary that is
committed under profiles/: per kernel AND grid size -- calls, average / min / max duration, share.
Usage: python profiles/summarize_rocpd.py <results.db> [title]"""
import sqlite3
import sys


def main(path, title=""):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, grid_x, workgroup_x, lds_size, vgpr_count, count(*), avg(duration), min(duration), "
        "max(duration), sum(duration) from kernels group by name, grid_x order by sum(duration) desc").fetchall()
    total = sum(r[-1] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary %s" % title)
    print("# source: %s ; durations in microseconds" % path.split("/")[-1])
    print("%-78s %9s %5s %7s %5s %7s %9s %9s %9s %6s" % ("kernel", "grid", "wg", "lds", "vgpr", "calls", "avg_us",
                                                          "min_us", "max_us", "pct"))
    for name, gx, wx, lds, vgpr, n, avg, mn, mx, tot in rows:
        short = name if len(name) <= 78 else name[:75] + "..."
        print("%-78s %9d %5d %7d %5d %7d %9.2f %9.2f %9.2f %6.2f" % (short, gx, wx, lds, vgpr, n, avg / 1e3, mn / 1e3,
                                                                       mx / 1e3, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
