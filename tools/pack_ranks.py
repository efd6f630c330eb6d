#!/usr/bin/env python
"""What decides >= 6x at 8 GPUs is not xGMI but the HOST: every rank packs its own shard.  This script runs R
packer ranks concurrently on this box -- each a process of its own, as under torch.distributed.run -- and reports the
per-rank time of the native packer (create_batch + the CSR build Session.prepare does, no GPU work) against one rank alone.
Usage: python tools/pack_ranks.py [ranks=8] [batches=20]      (prints one JSON line; tests/test_packer.py imports it)"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))


def _rank(args):
    rank, n_batches, start_at, B, n = args
    import numpy as np
    import tspgnn
    rng = np.random.RandomState(1000 + rank)
    pool = [tspgnn.random_instance(n, rng) for _ in range(2 * B)]
    tspgnn.InstanceLoader.create_batch(pool[:B], dev=0.02)      # warm-up (page in the library, numpy pools)
    while time.time() < start_at:                                # every rank starts packing at the same moment
        time.sleep(0.001)
    t0 = time.perf_counter()
    edges = 0
    for i in range(n_batches):
        k = (i * 37) % B
        EV, W, C, r, nv, ne = tspgnn.InstanceLoader.create_batch(pool[k:k + B], dev=0.02)
        EV.csr_by_vertex()                                       # what Session.prepare adds on the host
        edges += EV.shape[0]
    return (time.perf_counter() - t0) / n_batches, edges // n_batches


def measure(ranks=8, n_batches=20, B=128, n=40):
    ctx = mp.get_context("spawn")
    out = {}
    for r in (1, ranks):
        with ctx.Pool(r) as pool:
            start_at = time.time() + 6.0 + 0.5 * r                # (imports + pool generation take a few seconds per process)
            res = pool.map(_rank, [(k, n_batches, start_at, B, n) for k in range(r)])
        out[r] = [t for t, _ in res]
        edges = res[0][1]
    solo = out[1][0]
    worst = max(out[ranks])
    return {"what": "native packer (create_batch + CSR by vertex), C2 shard per rank (%d graphs of n=%d, %d edges)" % (B, n, edges),
            "host_cores": os.cpu_count(), "ranks": ranks, "batches_per_rank": n_batches,
            "ms_per_batch_one_rank": round(1e3 * solo, 3),
            "ms_per_batch_per_rank_concurrent": [round(1e3 * t, 3) for t in out[ranks]],
            "slowdown_worst_rank": round(worst / solo, 3)}


if __name__ == "__main__":
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    print(json.dumps(measure(r, nb)))
