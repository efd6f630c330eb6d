#!/usr/bin/env python
"""Where does the fresh-batch path's time go?  C2: (a) the prefetcher alone (pack + upload, nothing consumes the GPU), (b) copy_from +
replay of one resident batch, (c) both, as bench.py's serve leg runs them; with 1 / 2 / 3 packer workers."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
sys.path.insert(0, ROOT)
from bench import WORKLOADS
sizes, d, T, storage = WORKLOADS[name]
nb = 40 if name == "c2" else 12
model = tspgnn.build_network(d); sess = tspgnn.Session(model); sess.run(tspgnn.global_variables_initializer(seed=0))
rng = np.random.RandomState(5)
uniq = sorted(set(int(n) for n in sizes))
pool = {n: [tspgnn.random_instance(n, rng) for _ in range(64 if len(uniq) > 1 else 3 * len(sizes))] for n in uniq}
def instances(k):
    for i in range(k):
        yield [pool[int(n)][(i * 37 + j) % len(pool[int(n)])] for j, n in enumerate(sizes)]
pack = lambda inst: tspgnn.InstanceLoader.create_batch(inst, dev=0.02)
EV, W, C, r, nv, ne = pack(next(instances(1)))
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r, model["n_vertices"]: nv, model["n_edges"]: ne}
dev_batch = sess.prepare(feed); replay = sess.capture_forward(dev_batch)
for _ in range(5): replay()
torch.cuda.synchronize()
res = {"workload": name}
t0 = time.perf_counter()
for _ in range(nb): replay()
torch.cuda.synchronize(); res["replay_only_ms"] = round(1e3 * (time.perf_counter() - t0) / nb, 4)
other = sess.prepare(feed)
t0 = time.perf_counter()
for _ in range(nb): dev_batch.copy_from(other); replay()["predictions"].clone()
torch.cuda.synchronize(); res["copy_from_plus_replay_ms"] = round(1e3 * (time.perf_counter() - t0) / nb, 4)
t0 = time.perf_counter()
for inst in instances(8): pack(inst)
res["pack_ms_one_thread"] = round(1e3 * (time.perf_counter() - t0) / 8, 3)
t0 = time.perf_counter()
def feed_of(t):
    EV, W, C, r, nv, ne = t
    return {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r, model["n_vertices"]: nv, model["n_edges"]: ne}
for inst in instances(8): sess.prepare(feed_of(pack(inst)))
torch.cuda.synchronize(); res["pack_plus_prepare_ms_one_thread"] = round(1e3 * (time.perf_counter() - t0) / 8, 3)
for pinned in (False, True):
  for workers in (1, 2):
    tag = "w%d%s" % (workers, "_pinned" if pinned else "")
    for _ in tspgnn.BatchPrefetcher(sess, instances(6), T, workers=workers, pack=pack, pinned=pinned): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in tspgnn.BatchPrefetcher(sess, instances(nb), T, workers=workers, pack=pack, pinned=pinned): pass
    torch.cuda.synchronize(); res["prefetcher_alone_ms_" + tag] = round(1e3 * (time.perf_counter() - t0) / nb, 4)
    for rep in range(2):
        t0 = time.perf_counter()
        for b in tspgnn.BatchPrefetcher(sess, instances(nb), T, workers=workers, pack=pack, pinned=pinned):
            dev_batch.copy_from(b); replay()["predictions"].clone()
        torch.cuda.synchronize(); res["serve_ms_%s_run%d" % (tag, rep)] = round(1e3 * (time.perf_counter() - t0) / nb, 4)
print(json.dumps(res))
