"""Where the staged fresh-batch path (parallel.BatchStager) spends its time at C2: the native staging call alone, the
Python wrapper around it, the worker alone (stage + upload, consumer does nothing), the full loop.  python tools/stager_breakdown.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402

B, n, T, NB = 128, 40, 32, 48
rng = np.random.RandomState(99)
pool = [tspgnn.random_instance(n, rng) for _ in range(3 * B)]


def fresh(nb):
    for i in range(nb):
        yield [pool[(i * 37 + j) % len(pool)] for j in range(B)]


model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer())
stager = tspgnn.BatchStager(sess, next(fresh(1)), T)
replay = sess.capture_forward(stager.batch)
for _ in range(10):
    replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(NB):
    replay()
torch.cuda.synchronize()
print("replay only            %.3f ms per batch" % (1e3 * (time.perf_counter() - t0) / NB))
insts = list(fresh(8))
t0 = time.perf_counter()
for inst in insts:
    stager._stage(inst, 0)
print("_stage (python+native) %.3f ms per batch" % (1e3 * (time.perf_counter() - t0) / len(insts)))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for inst in insts:
    stager._stage(inst, 0)
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(8); print(s.getvalue()[:1500])
t0 = time.perf_counter()
for _ in stager.feed(fresh(NB)):
    pass
torch.cuda.synchronize()
print("worker alone (feed, no replay) %.3f ms per batch" % (1e3 * (time.perf_counter() - t0) / NB))
for _ in stager.feed(fresh(4)):
    replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in stager.feed(fresh(NB)):
    replay()
torch.cuda.synchronize()
print("full loop (feed + replay)      %.3f ms per batch" % (1e3 * (time.perf_counter() - t0) / NB))
t0 = time.perf_counter()
waited, it = 0.0, stager.feed(fresh(NB))
while True:
    w0 = time.perf_counter()
    try:
        next(it)
    except StopIteration:
        break
    waited += time.perf_counter() - w0
    replay()
torch.cuda.synchronize()
print("full loop again %.3f ms per batch, of which the consumer waited for the worker %.3f ms per batch"
      % (1e3 * (time.perf_counter() - t0) / NB, 1e3 * waited / NB))
