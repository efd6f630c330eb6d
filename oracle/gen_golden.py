#!/usr/bin/env python
"""Generates tests/golden/*.npz.  TEST INFRASTRUCTURE; runs only in the build container, where
/root/reference exists (the reference's Python never travels to the GPU box).

1. ``pack_*.npz`` -- outputs of the REFERENCE's own ``InstanceLoader.create_batch``
   (/root/reference/instance_loader.py:29-80, imported here, numpy-only) on seeded synthetic
   instances, stored next to the instances that produced them.  They pin the batch layout the
   hot path consumes: edge order, the two-ones-per-row EV pattern, W, the C closing-edge quirk,
   labels, count arrays.  EV is stored as the (row, col) coordinates of its non-zeros (and
   densely for the tiniest case).
2. ``oracle_*.npz`` -- float64 outputs of oracle/torch_oracle.py for seeded weights on two of
   those batches: regression anchors for the oracle itself (it has no reference vectors to
   pin against -- "parity unpinned", see oracle/__init__.py).

Usage:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")


def make_instance(n, rng, connectivity):
    """Seeded stand-in for dataset.create_graph (dataset.py:52-116) without Concorde:
    upper-triangular 0/1 adjacency with a planted Hamiltonian cycle, Euclidean weights, and that
    cycle as the route."""
    pts = rng.rand(n, 2)
    Mw = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
    Ma = np.triu((rng.rand(n, n) < connectivity).astype(int), 1)
    perm = [int(x) for x in rng.permutation(n)]
    for i, j in zip(perm, perm[1:] + perm[:1]):
        Ma[min(i, j), max(i, j)] = 1
    return Ma, Mw, perm


CASES = {
    # name: (sizes, connectivity, dev, target_cost, keep_dense)
    "n5_B2": ([5, 5], 1.0, 0.02, None, True),
    "n20_B32": ([20] * 32, 1.0, 0.02, None, False),
    "ragged_B6": ([3, 7, 20, 3, 7, 20], 1.0, 0.05, None, False),
    "sparse_B4": ([12, 9, 12, 9], 0.4, 0.02, None, False),
    "target_B4": ([6, 8, 6, 8], 1.0, 0.02, 0.3712, False),
}


def gen_pack():
    from instance_loader import InstanceLoader  # the reference's module

    for seed in (0, 1, 2):
        for name, (sizes, conn, dev, target, keep_dense) in CASES.items():
            rng = np.random.RandomState(1000 * seed + len(name))
            base = [make_instance(n, rng, conn) for n in sizes[::2]]
            # the reference yields every instance twice (instance_loader.py:21-23)
            instances = [inst for inst in base for _ in (0, 1)][:len(sizes)]
            EV, W, C, route_exists, n_vertices, n_edges = InstanceLoader.create_batch(
                instances, dev=dev, target_cost=target)
            r, c = np.nonzero(EV)
            assert np.all(EV[r, c] == 1)
            data = {
                "n_instances": np.int64(len(instances)),
                "dev": np.float64(dev),
                "target_cost": np.float64(np.nan if target is None else target),
                "ev_shape": np.array(EV.shape, dtype=np.int64),
                "ev_rows": r.astype(np.int32), "ev_cols": c.astype(np.int32),
                "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges,
            }
            for i, (Ma, Mw, route) in enumerate(instances):
                data["Ma_%d" % i] = Ma.astype(np.int8)
                data["Mw_%d" % i] = Mw
                data["route_%d" % i] = np.array(route, dtype=np.int64)
            if keep_dense:
                data["EV_dense"] = EV
            np.savez_compressed(os.path.join(OUT, "pack_%s_seed%d.npz" % (name, seed)), **data)
            print("pack", name, seed, EV.shape)


def gen_oracle():
    import torch
    from oracle import params as P
    from oracle import torch_oracle as TO

    for name, d, T in (("n5_B2", 32, 3), ("ragged_B6", 64, 4)):
        z = np.load(os.path.join(OUT, "pack_%s_seed0.npz" % name))
        M = int(z["ev_shape"][0])
        batch = {"ev_uv": z["ev_cols"].reshape(M, 2), "W": z["W"], "C": z["C"], "route_exists": z["route_exists"],
                 "n_vertices": z["n_vertices"], "n_edges": z["n_edges"]}
        params = P.init_params(d, seed=7, perturb=True)
        out, grads = TO.loss_and_grads(params, batch, T, dtype=torch.float64)
        data = {
            "d": np.int64(d), "T": np.int64(T), "param_seed": np.int64(7),
            "predictions": out["predictions"].detach().numpy(), "logits": out["logits"].detach().numpy(),
            "loss": np.float64(out["loss"].item()), "acc": np.float64(out["acc"].item()),
            "Vh": out["last_states"]["V"][0].detach().numpy(), "Eh_sum": out["last_states"]["E"][0].sum(0).detach().numpy(),
            "Ec_sum": out["last_states"]["E"][1].sum(0).detach().numpy(),
            "grad_norms": np.array([np.sqrt((g ** 2).sum()) for g in grads.values()]),
        }
        np.savez_compressed(os.path.join(OUT, "oracle_%s_d%d_T%d.npz" % (name, d, T)), **data)
        print("oracle", name, d, T, "loss", data["loss"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_pack()
    gen_oracle()
