"""Data-parallel path: instances sharded over ranks, ONE all-reduce of the flat
gradient bucket, weighted by B_r/B.  Runs on CPU with the gloo backend, world_size 2 (and 3 with
unequal shards); the per-rank gradients come from the oracle, the all-reduce / weighting logic is
the product's Session.allreduce_grads.  RCCL itself only exists on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pack


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _shard(g, lo, hi):
    eo = np.concatenate([[0], np.cumsum(g["n_edges"])]); vo = np.concatenate([[0], np.cumsum(g["n_vertices"])])
    return {"ev_uv": g["ev_uv"][eo[lo]:eo[hi]] - vo[lo], "W": g["W"][eo[lo]:eo[hi]], "C": g["C"][eo[lo]:eo[hi]],
            "route_exists": g["route_exists"][lo:hi], "n_vertices": g["n_vertices"][lo:hi], "n_edges": g["n_edges"][lo:hi]}


def _worker(rank, world, port, bounds, q):
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tspgnn
    from oracle import params as P
    from oracle import torch_oracle as TO
    torch.set_num_threads(1)
    d, T = 32, 2
    g = load_pack("ragged_B6", 0)
    params = P.init_params(d, seed=3, perturb=True)
    lo, hi = bounds[rank], bounds[rank + 1]
    _, local = TO.loss_and_grads(params, _shard(g, lo, hi), T)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model, device="cpu")          # plumbing only: no kernel is launched on CPU
    store = model.store
    store.zero_grad()
    for name in store.names():
        # the oracle's gradient includes the L2 term, which the product adds AFTER the all-reduce
        store.grad_view(name).copy_(torch.from_numpy(local[name] - TO.L2NORM_SCALING * params[name]).float())
    w = sess.allreduce_grads(hi - lo)
    if rank == 0:
        q.put((w, store.grad_dict()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bounds", [[0, 2, 6], [0, 4, 6], [0, 2, 4, 6]])
def test_sharded_gradient_equals_global_batch_gradient(bounds):
    from oracle import params as P
    from oracle import torch_oracle as TO
    world = len(bounds) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    w0, got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_pack("ragged_B6", 0)
    params = P.init_params(32, seed=3, perturb=True)
    _, ref = TO.loss_and_grads(params, _shard(g, 0, 6), 2)
    assert abs(w0 - (bounds[1] - bounds[0]) / 6.0) < 1e-12
    for k in ref:
        want = ref[k] - TO.L2NORM_SCALING * params[k]
        scale = max(np.abs(want).max(), 1e-12)
        assert np.abs(got[k] - want).max() / scale < 5e-6, k


def test_allreduce_is_noop_without_process_group():
    import tspgnn
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model, device="cpu")
    model.store.zero_grad()
    model.store.grad.fill_(2.0)
    assert sess.allreduce_grads(8) == 1.0 and float(model.store.grad[0]) == 2.0
