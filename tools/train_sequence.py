#!/usr/bin/env python
"""Kernel sequence of the LAST training step in a rocprofv3 --kernel-trace rocpd database: launch order, durations, and the
busy / span totals (development aid).  Usage: python tools/train_sequence.py <results.db> [max_lines]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "einit_fwd" in r[0]]
# a training step runs einit_fwd once (forward); take the last complete step
lo, hi = starts[-2], starts[-1]
seq = rows[lo:hi]
busy = sum(e - s for _, s, e in seq)
span = seq[-1][2] - seq[0][1]
print("# last complete step: %d kernels, span %.1f us, busy %.1f us, gaps %.1f us" % (len(seq), span / 1e3, busy / 1e3, (span - busy) / 1e3))
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("tspgnn::", "")[:60]
out, prev, cnt, dur = [], None, 0, 0.0
for name, s, e in seq:
    n = short(name)
    out.append("%-60s %8.2f" % (n, (e - s) / 1e3))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mid = len(out) // 2
print("\n".join(out[:40]))
print("...")
print("\n".join(out[mid + 40:mid + 40 + limit]))
