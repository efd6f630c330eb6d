#!/usr/bin/env python
"""Experiment (round 4): throughput of the forward with TWO independent batches in flight -- one captured graph per batch, each
replayed on its own stream -- against the same two graphs replayed alternately on one stream.  Consecutive forward
passes of a serving loop are independent, so the second pass's launches can fill the first one's launch boundaries, tails
and its (issue-idle, bandwidth-bound) row-sum launches.
Usage: python tools/two_in_flight.py [c2|c4] [replays per stream]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sizes, d, T, storage = WORKLOADS[name]
model = tspgnn.build_network(d, float_dtype=torch.bfloat16 if storage == "bf16" else torch.float32)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
batches, replays = [], []
for seed in (1234, 4321):
    EV, W, C, route_exists, n_vertices, n_edges = tspgnn.synthetic_batch(sizes, seed=seed)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
            model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    batches.append(b)
    replays.append(sess.capture_forward(b))
ref = [sess.forward_device(b)["predictions"].clone() for b in batches]
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one_stream(k):
    for _ in range(k):
        for r in replays:
            r()


def two_streams(k):
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
    for _ in range(k):
        for s, r in zip(streams, replays):
            with torch.cuda.stream(s):
                r()
    for s in streams:
        main.wait_stream(s)


def timed(fn, k):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * k)


res = {"workload": name, "one_stream_ms_per_pass": [], "two_streams_ms_per_pass": []}
for _ in range(4):
    res["one_stream_ms_per_pass"].append(round(1e3 * timed(one_stream, n), 4))
    res["two_streams_ms_per_pass"].append(round(1e3 * timed(two_streams, n), 4))
outs = [r()["predictions"] for r in replays]
two_streams(2)
torch.cuda.synchronize()
res["predictions_bit_identical_to_eager"] = bool(all(torch.equal(a, o) for a, o in zip(ref, outs)))
print(json.dumps(res))
