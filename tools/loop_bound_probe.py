#!/usr/bin/env python
"""Upper bound on what a resident T-step loop built from the STEPWISE cell kernel can reach: the cell +
message launch of one step repeated R times INSIDE one launch -- weights staged once, the LDS ticket running through R passes
over the workgroup's tile range, the lock-step vertex task looping beside it, no synchronisation between passes (wrong
numbers, right time) -- against the same R steps as separate launches (row-sum + cell).
Needs the variant library: cp tsp-gnn_amd/csrc/dense_h2.hip tools/variants/dense_h2_bound.hip; patch tools/variants/dense_h2_bound.hip tools/h2_bound.patch;
SRC=$PWD/tools/variants/dense_h2_bound.hip tools/build_variant.sh bound
Usage: TSPGNN_LIB=tools/variants/bound.so TSPGNN_LOOP=0 python tools/loop_bound_probe.py [graphs=128] [R=32]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
os.environ["TSPGNN_LOOP"] = "0"
import tspgnn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
EV, W, C, r, nv, ne = tspgnn.synthetic_batch([40] * B, seed=1234)


def timed(T, reps):
    os.environ["TSPGNN_BOUND_REPS"] = str(reps)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
            model["n_vertices"]: nv, model["n_edges"]: ne}
    b = sess.prepare(feed)
    replay = sess.capture_forward(b)
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    return best * 1e6


a = timed(2, 1)
b_ = timed(2, R)
c = timed(R + 1, 1)
full = timed(32, 1)
out = {"graphs": B, "R": R, "T2_us": round(a, 1), "T2_reps_us": round(b_, 1), "Tstep_us": round(c, 1),
       "in_launch_us_per_step": round((b_ - a) / (R - 1), 2), "stepwise_us_per_step": round((c - a) / (R - 1), 2),
       "forward_T32_ms": round(full / 1e3, 4),
       "bound_T32_ms_rowsum_hidden": round((full - 31 * ((c - a) / (R - 1)) + 31 * (b_ - a) / (R - 1)) / 1e3, 4)}
print(json.dumps(out))
