#!/usr/bin/env python
"""End-to-end forward-only serving rate at C2 (host instances -> packed batch -> upload -> forward), i.e. the
PCIe- and packer-inclusive number next to bench.py's resident-batch headline.  Fresh instances every batch:
BatchPrefetcher packs and uploads batch i+1 on a worker thread / side stream while the GPU runs batch i."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402

B, n, T, nb = 128, 40, 32, int(os.environ.get("BATCHES", 60))
rng = np.random.RandomState(0)
pool = [tspgnn.random_instance(n, rng) for _ in range(4 * B)]


def batches():
    for i in range(nb):
        k = (i * 37) % (3 * B)
        yield tspgnn.InstanceLoader.create_batch(pool[k:k + B], dev=0.02)


pool_batch = tspgnn.InstanceLoader.create_batch(pool[:B], dev=0.02)
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
out = None
for b in tspgnn.BatchPrefetcher(sess, batches(), T):      # warm-up pass (allocator, weight packs)
    out = sess.forward_device(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
preds = []
for b in tspgnn.BatchPrefetcher(sess, batches(), T):
    preds.append(sess.forward_device(b)["predictions"])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in batches():
    pass
pack = (time.perf_counter() - t1) / nb
# same-shaped batches through ONE captured graph: each prefetched batch is copied into the graph's resident buffers
static = sess.prepare({model["EV"]: pool_batch[0], model["W"]: pool_batch[1], model["C"]: pool_batch[2],
                       model["time_steps"]: T, model["route_exists"]: pool_batch[3], model["n_vertices"]: pool_batch[4],
                       model["n_edges"]: pool_batch[5]})
replay = sess.capture_forward(static)
torch.cuda.synchronize()
t3 = time.perf_counter()
for b in tspgnn.BatchPrefetcher(sess, batches(), T):
    static.copy_from(b)
    preds.append(replay()["predictions"].clone())
torch.cuda.synchronize()
graphed = (time.perf_counter() - t3) / nb
# the same without the worker thread: pack, upload and launch from one thread
m = model
torch.cuda.synchronize()
t2 = time.perf_counter()
for t in batches():
    EV, W, C, r, nv, ne = t
    b = sess.prepare({m["EV"]: EV, m["W"]: W, m["C"]: C, m["time_steps"]: T, m["route_exists"]: r, m["n_vertices"]: nv,
                      m["n_edges"]: ne})
    preds.append(sess.forward_device(b)["predictions"])
torch.cuda.synchronize()
inline = (time.perf_counter() - t2) / nb
print(json.dumps({"workload": "c2 serving: fresh instances every batch, pack + upload + forward (eager launches)",
                  "batches": nb, "ms_per_batch_end_to_end": round(1e3 * dt / nb, 3),
                  "mp_steps_per_s_end_to_end": round(nb * T / dt, 1), "host_pack_ms_per_batch": round(1e3 * pack, 3),
                  "ms_per_batch_single_thread": round(1e3 * inline, 3),
                  "ms_per_batch_graph_replay": round(1e3 * graphed, 3),
                  "mp_steps_per_s_graph_replay": round(T / graphed, 1)}))
