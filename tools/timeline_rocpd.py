#!/usr/bin/env python
"""Kernel timeline of the LAST forward replay in a rocprofv3 --kernel-trace rocpd database (tools/forward_graph.py): every
kernel in launch order with its duration and the idle gap before it, then totals (busy, gaps, span).
Usage: python tools/timeline_rocpd.py <results.db> [first-kernel-substring, default einit_fwd]"""
import sqlite3
import sys


def main(path, first="einit_fwd"):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        raise SystemExit("need at least two passes in the trace")
    lo, hi = starts[-2], starts[-1]          # the last complete pass
    seq = rows[lo:hi]
    busy = sum(e - s for _, s, e in seq)
    span = seq[-1][2] - seq[0][1]
    print("# last complete pass: %d kernels, span %.1f us, kernels busy %.1f us, gaps %.1f us" % (
        len(seq), span / 1e3, busy / 1e3, (span - busy) / 1e3))
    print("%-70s %9s %9s" % ("kernel", "dur_us", "gap_us"))
    prev_end = None
    agg = {}
    for name, s, e in seq:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        short = name if len(name) <= 70 else name[:67] + "..."
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap
        prev_end = e
    for k, (n, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-70s %9.1f %9.1f   x%d (avg %.2f us, avg gap before %.2f us)" % (k, dur, gap, n, dur / n, gap / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "einit_fwd")
