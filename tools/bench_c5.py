#!/usr/bin/env python
"""BASELINE config 5, one GPU's shard: n=200, 32 graphs (batch 256 over 8 GPUs), embed=128, T=64 --
forward in the bf16-storage mode (build_network(d, float_dtype=torch.bfloat16)) next to the fp32 mode.
Development aid (gpurun); the headline metric stays bench.py's C2 line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402

n, B, d, T = int(os.environ.get("N", 200)), int(os.environ.get("B", 32)), int(os.environ.get("D", 128)), int(os.environ.get("T", 64))
batch = tspgnn.synthetic_batch([n] * B, seed=1234)
EV, W, C, route_exists, n_vertices, n_edges = batch
M, N = EV.shape
out = {"workload": "c5 shard: %d graphs n=%d, d=%d, T=%d" % (B, n, d, T), "N": N, "M": M}
for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    model = tspgnn.build_network(d, float_dtype=dtype)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer(seed=0))
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    replay = sess.capture_forward(b)
    for _ in range(2):
        o = replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        o = replay()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    out[tag] = {"ms_per_forward": round(ms, 3), "mp_steps_per_s": round(T / (ms * 1e-3), 1),
                "us_per_mp_step": round(1e3 * ms / T, 1), "loss": float(o["stats"][0].item()),
                "predictions_head": [round(float(x), 5) for x in o["predictions"][:4]]}
    from tspgnn import _lib
    _lib.TIMELINE = []
    sess.forward_device(b)
    torch.cuda.synchronize()
    per = {}
    for name, e0, e1 in _lib.TIMELINE:
        per.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    _lib.TIMELINE = None
    out[tag]["kernels_us"] = {k: round(sum(v) / len(v), 1) for k, v in per.items() if len(v) >= T}
    del replay, sess, model
    torch.cuda.empty_cache()
print(json.dumps(out))
