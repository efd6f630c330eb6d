"""Where lnlstm_bwd_h2 spends a C2 backward step: per-wavefront sums of s_memrealtime ticks (100 MHz) per phase, from a
library built with tools/bwd_trace.patch (git apply; make -C tsp-gnn_amd/csrc).  python tools/bwd_trace.py [graphs=128]
phases: 0 staging K, K^T, slabs (per launch)  1 Zx gather arrives  2 h loads + recompute z = h Kh  3 tile backward (LayerNorms,
gates, slab sums)  4 dz, dc stores issued  5 dh = dz Kh^T + stores  6 -  7 slab fold"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402
from tspgnn import _lib  # noqa: E402
from oracle import params as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = 32
t = tspgnn.synthetic_batch([40] * B, seed=0)
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer())
model.store.load(P.init_params(64, seed=1, perturb=True))
EV, W, C, r, nv, ne = t
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
        model["n_vertices"]: nv, model["n_edges"]: ne}
b = sess.prepare(feed)
for _ in range(2):
    sess.loss_and_grads(b)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
fn = _lib.lib.tspgnn_debug_bwd_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf.ctypes.data, 1) == 0
sess.loss_and_grads(b)
torch.cuda.synchronize()
assert fn(buf.ctypes.data, 0) == 0
full = buf.reshape(256, 8, 16).astype(np.float64)
print("shader clock during the launch: %.0f MHz (s_memtime ticks per s_memrealtime tick x 100 MHz)" % (100.0 * full[:, :, 8].sum() / max(full[:, :, 9].sum(), 1)))
tr = buf.reshape(256, 8, 16)[:, :, :8].astype(np.float64) * 0.01 / T      # us per launch and wavefront
edge = tr[:, :, 5].sum(1) > 0            # workgroups of the edge task (the fused data gradient runs there only)
for name, sel in (("edge-cell workgroups", edge), ("vertex-cell workgroups", ~edge & (tr.sum((1, 2)) > 0))):
    x = tr[sel].reshape(-1, 8)
    x = x[x.sum(1) > 0]
    if len(x) == 0:
        continue
    print("%s: %d workgroups, %d wavefronts" % (name, int(sel.sum()), len(x)))
    print("  phase      " + " ".join("%7d" % i for i in range(8)) + "    total   (us per launch and wavefront)")
    for label, v in (("mean", x.mean(0)), ("p10", np.percentile(x, 10, axis=0)), ("p90", np.percentile(x, 90, axis=0)),
                     ("max", x.max(0))):
        print("  %-9s  " % label + " ".join("%7.2f" % a for a in v) + "  %7.2f" % v.sum())
M = EV.shape[0] if hasattr(EV, "shape") else 780 * B
print("edge tiles per wavefront: %.2f" % ((780 * B / 16) / max(1, int(edge.sum()) * 8)))
tot = tr.sum(2)                      # [workgroup, wavefront] us per launch
wg = tot.max(1)
order = np.argsort(-wg)
print("slowest workgroups (index: slowest wavefront's us per launch):", ", ".join("%d: %.1f" % (i, wg[i]) for i in order[:12]))
print("workgroups 0-15:", " ".join("%.0f" % x for x in wg[:16]))
print("workgroups 240-255:", " ".join("%.0f" % x for x in wg[240:]))
print("median workgroup %.1f, p90 %.1f, max %.1f" % (np.median(wg), np.percentile(wg, 90), wg.max()))
