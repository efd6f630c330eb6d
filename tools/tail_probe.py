#!/usr/bin/env python
"""How does the forward scale with the number of graphs around C2's 128?  The edge task hands 16-row tiles to 229 workgroups x 12
wavefronts: at 128 graphs of n = 40 that is 27.25 tiles per workgroup = 2.27 per wavefront, i.e. a thin third round.  If the
step time is a staircase in the tile count the tail is worth attacking; if it is a line it is not.
Usage: python tools/tail_probe.py [T=32]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
out = []
for B in [int(x) for x in os.environ.get("GRAPHS", "88,96,104,112,116,120,124,128,132,136,144,152,160,176").split(",")]:
    EV, W, C, r, nv, ne = tspgnn.synthetic_batch([40] * B, seed=1234)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
            model["n_vertices"]: nv, model["n_edges"]: ne}
    b = sess.prepare(feed)
    replay = sess.capture_forward(b)
    for _ in range(10):
        replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    tiles = (EV.shape[0] + 15) // 16
    out.append({"graphs": B, "edge_tiles": tiles, "tiles_per_wg_229": round(tiles / 229.0, 2), "ms": round(1e3 * best, 4),
                "us_per_step": round(1e6 * best / T, 2), "ns_per_tile_step": round(1e9 * best / T / tiles, 2)})
    print(out[-1], flush=True)
print(json.dumps(out))
