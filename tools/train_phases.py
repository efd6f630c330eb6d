#!/usr/bin/env python
"""The last complete training step of a rocprofv3 --kernel-trace rocpd database, cut into phases (head: packs and the
pre-loop; forward loop; between the loops: vote head, loss, their backward; backward loop; tail: weight gradients, folds,
optimiser): kernels, busy time and span of each.  Usage: python tools/train_phases.py <results.db> [list]"""
import re
import sqlite3
import sys
import collections

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "einit_fwd" in r[0]]
seq = rows[starts[-2]:starts[-1]]
# the step begins with its packs, which precede einit_fwd: shift the cut back to the first kernel after the previous step's adam
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n)).replace("tspgnn::", "")[:48]
names = [short(r[0]) for r in seq]
def first(pat, lo=0):
    return next(i for i in range(lo, len(names)) if pat in names[i])
def last(pat):
    return max(i for i in range(len(names)) if pat in names[i])
f0, f1 = first("lnlstm_mlp_fwd"), last("lnlstm_mlp_fwd")
b0, b1 = first("lnlstm_bwd"), last("mlp_bwd_h2")
cuts = [("pre-loop (einit, first messages)", 0, f0), ("forward loop", f0, f1 + 1), ("vote head, loss, their backward", f1 + 1, b0),
        ("backward loop", b0, b1 + 1), ("tail: weight gradients, folds, optimiser, next step's packs", b1 + 1, len(seq))]
print("# step: %d kernels, span %.1f us, busy %.1f us" % (len(seq), (seq[-1][2] - seq[0][1]) / 1e3, sum(e - s for _, s, e in seq) / 1e3))
for label, a, b in cuts:
    part = seq[a:b]
    if not part:
        continue
    busy = sum(e - s for _, s, e in part) / 1e3
    span = (part[-1][2] - part[0][1]) / 1e3
    print("%-62s %4d kernels  busy %8.1f us  span %8.1f us" % (label, len(part), busy, span))
    if len(sys.argv) > 2:
        agg = collections.OrderedDict()
        for (n, s, e) in part:
            k = short(n)
            agg.setdefault(k, [0, 0.0])
            agg[k][0] += 1
            agg[k][1] += (e - s) / 1e3
        for k, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("      %-48s %4d  %8.1f" % (k, cnt, us))
