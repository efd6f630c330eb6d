"""Where the one-launch loop (tspgnn_mp_loop_h2) spends its time: per-wavefront sums of s_memrealtime ticks (100 MHz) per
phase, averaged per step.  TSPGNN_LOOP_TRACE=1 python tools/loop_trace.py [graphs=128] [n=40] [T=32]
edge wavefronts:   0 wait for the group's message tiles   1 row-sum share (+ drain, arrive)   2 wait for projected tiles
                   3 acquire fence   4 resident tiles (cells + message MLP)   5 drain + arrive
                   inside 4, summed over the tiles: 8 gather + f stage  9 i,j stage  10 o stage  11 split + MLP layer 1
                   12 MLP layers 2.. (the marks 8.. make phase 4 = the time after the last mark: loop overhead)  13 stores
vertex wavefronts: 0 wait for aggregates + operand fetch   1 K staging wait + barrier   2 cells   3 barrier + MLP staging
                   4 message MLP + projection   5 drain + arrive   6 barrier + K staging issue"""
import os
import sys

os.environ["TSPGNN_LOOP_TRACE"] = "1"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402
from oracle import params as P  # noqa: E402
from tspgnn import loop_plan as LP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
t = tspgnn.synthetic_batch([n] * B, seed=0)
params = P.init_params(64, seed=1, perturb=True)
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer())
model.store.load(params)
EV, W, C, r, nv, ne = t
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
        model["n_vertices"]: nv, model["n_edges"]: ne}
b = sess.prepare(feed)
for _ in range(3):
    sess.forward_device(b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sess.forward_device(b)
e1.record()
torch.cuda.synchronize()
print("eager forward with trace: %.3f ms" % e0.elapsed_time(e1))
tr = model["gnn"].loop_trace.cpu().numpy().astype(np.float64) * 0.01 / T     # us per step
plan, G, grid = b.adj.loop_plan[:3]
p = plan.cpu().numpy().reshape(grid, LP.WAVES, LP.DESC)
print("\n".join(__doc__.split("\n")[2:]))
for role, name in ((1, "edge"), (2, "vertex")):
    sel = (p[:, :, 0] == role) & (p[:, :, 1] > 0)
    x = tr[sel]
    if len(x) == 0:
        continue
    print("%s wavefronts: %d, tiles per wavefront %s" % (name, len(x), np.bincount(p[:, :, 1][sel]).tolist()))
    print("  phase      " + " ".join("%7d" % i for i in range(x.shape[1])) + "    total")
    for label, v in (("mean", x.mean(0)), ("p10", np.percentile(x, 10, axis=0)), ("p90", np.percentile(x, 90, axis=0)),
                     ("max", x.max(0))):
        print("  %-9s  " % label + " ".join("%7.2f" % a for a in v) + "  %7.2f" % v.sum())
    if role == 1:
        for k in (3, 4):
            xs = tr[sel & (p[:, :, 1] == k)]
            if len(xs):
                tiles_us = xs[:, [4, 8, 9, 10, 11, 12, 13]].sum(1).mean()
                print("  %d-tile wavefronts: resident tiles %.2f us per step = %.2f per tile" % (k, tiles_us, tiles_us / k))
