"""Data-parallel path: instances sharded over ranks, ONE all-reduce of the bucket
[flat gradient | batch size, statistics], weighted by B_r/B.  Runs on CPU with the gloo backend, world_size 2
(and 3 with unequal shards); the per-rank gradients and statistics come from the oracle, the bucket packing,
the collective and the division by the reduced batch size are the product's Session.allreduce_grads
(tensor ops only, so they run here).  The same logic with the HIP backward producing the gradients is covered
on the GPU box by tests/test_gpu_dp.py; RCCL itself only exists there."""
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
    out, local = TO.loss_and_grads(params, _shard(g, lo, hi), T)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model, device="cpu")          # plumbing only: no kernel is launched on CPU
    store = model.store
    store.zero_grad()
    for name in store.names():
        # the oracle's gradient includes the L2 term, which the product adds AFTER the all-reduce
        store.grad_view(name).copy_(torch.from_numpy(local[name] - TO.L2NORM_SCALING * params[name]).float())
    stats = torch.tensor([float(out[k]) for k in ("loss", "acc", "TP", "FP", "TN", "FN")], dtype=torch.float32)
    assert sess.world_size == world
    sess.allreduce_grads(hi - lo, stats)
    theta0 = store.theta.clone()
    if rank != 0:
        store.theta.add_(1.0)          # a replica that drifted: the broadcast restores rank 0's variables
    else:
        # rank 0 alone restored a checkpoint with optimiser slots (util.load_weights creates them there only): every
        # rank must still issue the same collectives, and end up with rank 0's moments and step count
        sess._ensure_adam()
        sess._adam["m"].fill_(3.0); sess._adam["v"].fill_(5.0); sess._adam["t"].fill_(7); sess._adam["step"] = 7
    sess.broadcast_variables(0)
    slots_ok = (float(sess._adam["m"].min()) == 3.0 and float(sess._adam["v"].max()) == 5.0
                and int(sess._adam["t"]) == 7 and sess._adam["step"] == 7)
    # an assignment of the variables after the first sync (store.load on rank 0) triggers a new broadcast
    sess._sync_replicas_once()         # nothing changed: the ranks agree on that (one 1-int all-reduce), no broadcast
    store.load({"V_init": np.full((1, d), float(rank + 2), dtype=np.float32)})
    sess._sync_replicas_once()
    resynced = float(store.view("V_init").min()) == 2.0 and float(store.view("V_init").max()) == 2.0
    # ... and so does an assignment on rank 0 ALONE (a restore there only): its peers' counters did not move, the
    # decision to broadcast is taken together -- before round 4 rank 0 broadcast while the others went on: a hang
    if rank == 0:
        store.load({"V_init": np.full((1, d), 9.0, dtype=np.float32)})
    sess._sync_replicas_once()
    resynced = resynced and float(store.view("V_init").min()) == 9.0 and float(store.view("V_init").max()) == 9.0
    store.load({"V_init": theta0[store.offset("V_init"):store.offset("V_init") + d].numpy()})
    sess._sync_replicas_once()
    means = sess.allreduce_host_sums(np.array([float(hi - lo), 1.0]))
    if rank == world - 1:
        q.put((store.grad_dict(), stats.numpy().copy(),
               bool(torch.equal(store.theta, theta0)) and slots_ok and resynced, means))
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
    got, stats, same_theta, means = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_pack("ragged_B6", 0)
    params = P.init_params(32, seed=3, perturb=True)
    out, ref = TO.loss_and_grads(params, _shard(g, 0, 6), 2)
    assert same_theta and means.tolist() == [6.0, float(world)]
    # statistics of the GLOBAL batch (model.py:150-157): B-weighted means of loss / acc, sums of TP..FN
    for i, k in enumerate(("loss", "acc", "TP", "FP", "TN", "FN")):
        assert abs(stats[i] - float(out[k].detach())) < 2e-6, (k, stats[i])
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
    stats = torch.arange(6, dtype=torch.float32)
    sess.allreduce_grads(8, stats)
    assert sess.world_size == 1 and float(model.store.grad[0]) == 2.0 and stats.tolist() == [0, 1, 2, 3, 4, 5]


def test_cpu_session_is_plumbing_only():
    """no CPU compute path -- a fetch on a device='cpu' session raises instead of running anything."""
    import tspgnn
    from conftest import load_pack as lp
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model, device="cpu")
    sess.run(tspgnn.global_variables_initializer())
    g = lp("n5_B2", 0)
    feed = {model["EV"]: tspgnn.SparseEV(g["ev_uv"], int(g["n_vertices"].sum())), model["W"]: g["W"], model["C"]: g["C"],
            model["time_steps"]: 1, model["route_exists"]: g["route_exists"], model["n_vertices"]: g["n_vertices"],
            model["n_edges"]: g["n_edges"]}
    with pytest.raises(RuntimeError, match="plumbing only"):
        sess.run(model["predictions"], feed_dict=feed)
    with pytest.raises(RuntimeError, match="plumbing only"):
        sess.run(model["train_step"], feed_dict=feed)
