#!/usr/bin/env python
"""Development aid: per-phase cycle sums of one workgroup of the fused cell launch's edge task, from an instrumented build.
The instrumentation lives OUTSIDE the shipped kernel source, as tools/h2_trace.patch:
    patch -o /tmp/dense_h2_traced.hip tsp-gnn_amd/csrc/dense_h2.hip tools/h2_trace.patch
    SRC=/tmp/dense_h2_traced.hip tools/build_variant.sh trace -DH2_TRACE=1
    TSPGNN_LIB=tools/variants/trace.so python tools/h2_trace.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402
from tspgnn import _lib  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
sizes, d, T, storage = WORKLOADS[name]
EV, W, C, route_exists, n_vertices, n_edges = tspgnn.synthetic_batch(sizes, seed=1234)
model = tspgnn.build_network(d)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
        model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
b = sess.prepare(feed)
for _ in range(3):
    sess.forward_device(b)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
fn = _lib.lib.tspgnn_debug_h2_trace
fn.restype = ctypes.c_int
rc = fn(buf)
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ticket", "endpoints+gathers", "K GEMM", "gates+stores", "MLP+store", "p5", "p6", "tiles"]
print("rc", rc, "phases:", names)
tot = [0] * 8
for w in range(16):
    row = [buf[w * 8 + i] for i in range(8)]
    if row[7] == 0:
        continue
    n = row[7]
    print("wave %2d tiles %d: " % (w, n) + "  ".join("%s %6.0f" % (names[i][:10], row[i] / n) for i in range(7) if row[i]) + "   sum/tile %.0f" % (sum(row[:7]) / n))
    for i in range(8):
        tot[i] += row[i]
if tot[7]:
    print("mean per tile: " + "  ".join("%s %.0f (%.0f%%)" % (names[i], tot[i] / tot[7], 100.0 * tot[i] / sum(tot[:7])) for i in range(7) if tot[i]))
