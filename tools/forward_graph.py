#!/usr/bin/env python
"""Profiling target: the HIP-graph replay of one forward pass of a bench.py workload, N times, and nothing else -- for a
kernel timeline (tools/timeline_rocpd.py) of what one replay consists of.
Usage: python tools/forward_graph.py [c2|c4|c5] [replays]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sizes, d, T, storage = WORKLOADS[name]
EV, W, C, route_exists, n_vertices, n_edges = tspgnn.synthetic_batch(sizes, seed=1234)
model = tspgnn.build_network(d, float_dtype=torch.bfloat16 if storage == "bf16" else torch.float32)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
        model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
b = sess.prepare(feed)
replay = sess.capture_forward(b)
for _ in range(replays):
    out = replay()
torch.cuda.synchronize()
print(name, "loss", float(out["stats"][0].item()))
