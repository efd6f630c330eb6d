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

3. ``anchor_*.npz`` -- float64 oracle outputs at the FULL size of BASELINE.json's configs 0, 1 and 3 (C1: n=20,
   B=32, T=8; C2: n=40, B=128, T=32; C4: ragged n in 20..80, B=512, T=2) for ``init_params(64, seed=0)`` on
   ``tspgnn.synthetic_batch`` inputs: predictions, logits, loss, and for each of E.h, E.c, V.h, V.c the column sums
   and 512 evenly spaced rows -- so that the GPU suite checks full-size parity in seconds without running the
   oracle there (tests/test_gpu_anchors.py).  Input fingerprints are stored next to them: the test first checks
   that it regenerated the same batch and the same weights.
4. ``graph_*.npz`` -- ``.graph`` instance files (text) together with what the REFERENCE's own ``read_graph``
   (/root/reference/instance_loader.py:95-127, imported here) parses out of them: pins the native reader.

5. ``anchor_grad_*.npz`` -- float64 GRADIENTS at full size (C2 at T = 2, C1 at T = 8): per variable its norm, largest
   entry, 64 sampled entries and the loss of the fp32 autograd restatement (oracle/anchors.py GRAD_ANCHORS; "grads").
6. ``trained_d64.npz`` -- the oracle's own weights after 2 000 Adam steps at lr 1e-3 on synthetic batches: a network
   whose statistics are not those of the initialisers, for parity far from init ("trained").

Usage:  python oracle/gen_golden.py [pack] [oracle] [anchors] [graph] [bf16] [grads] [trained]   (default: the first five)
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


from oracle.anchors import (ANCHORS, BF16_ANCHORS, GRAD_ANCHORS, TRAINED, anchor_inputs, anchor_rows,  # noqa: E402
                            bf16_anchor_inputs, grad_anchor_inputs, grad_sample_index)


def gen_anchors(only=None):
    import time
    import torch
    from oracle import torch_oracle as TO

    torch.set_num_threads(os.cpu_count() or 1)
    for name in ANCHORS:
        if only is not None and name not in only:
            continue
        batch, params, T, finger = anchor_inputs(name)
        EV, W, C, route_exists, n_vertices, n_edges = batch
        ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
        t0 = time.time()
        with torch.no_grad():
            ref = TO.forward(TO.to_torch(params, torch.float64), ob, T)
        data = {"T": np.int64(T), "d": np.int64(64), "fingerprint": finger,
                "predictions": ref["predictions"].numpy(), "logits": ref["logits"].numpy(),
                "loss": np.float64(ref["loss"].item()), "acc": np.float64(ref["acc"].item())}
        for var in ("E", "V"):
            for k, part in enumerate(("h", "c")):
                a = ref["last_states"][var][k].numpy()
                rows = anchor_rows(a.shape[0])
                data["%s%s_rows" % (var, part)] = a[rows]
                data["%s%s_colsum" % (var, part)] = a.sum(0)
                data["%s%s_absmax" % (var, part)] = np.float64(np.abs(a).max())
        np.savez_compressed(os.path.join(OUT, "anchor_%s.npz" % name), **data)
        print("anchor", name, "T", T, "M", EV.shape[0], "loss %.9f" % data["loss"], "%.1f s" % (time.time() - t0))


def gen_grad_anchors(only=None):
    """anchor_grad_*.npz: float64 gradients at full size, what the op-for-op float32 restatement loses per variable, and
    the per-variable spread of the float64 gradient under one-ulp perturbations of the weights (oracle/anchors.py)."""
    import time
    import torch
    from oracle import torch_oracle as TO
    from oracle.anchors import GRAD_PERTURBATIONS, ulp_perturbed

    torch.set_num_threads(os.cpu_count() or 1)
    for name in GRAD_ANCHORS:
        if only is not None and name not in only:
            continue
        batch, params, T, finger = grad_anchor_inputs(name)
        EV, W, C, route_exists, n_vertices, n_edges = batch
        ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
        t0 = time.time()
        out, g = TO.loss_and_grads(params, ob, T, dtype=torch.float64)
        print("  float64 gradient %.1f s" % (time.time() - t0), flush=True)
        _, g32 = TO.loss_and_grads(params, ob, T, dtype=torch.float32)
        data = {"T": np.int64(T), "d": np.int64(64), "fingerprint": finger, "loss": np.float64(out["loss"].item())}
        gmax = max(float(np.abs(v).max()) for v in g.values())
        data["grad_absmax"] = np.float64(gmax)
        for k, v in g.items():
            flat = np.asarray(v, dtype=np.float64).reshape(-1)
            idx = grad_sample_index(k, flat.size)
            data["norm:" + k] = np.float64(np.sqrt((flat ** 2).sum()))
            data["absmax:" + k] = np.float64(np.abs(flat).max())
            data["sample:" + k] = flat[idx]
            f32 = np.asarray(g32[k], dtype=np.float64).reshape(-1)
            data["err32:" + k] = np.float64(np.abs(f32 - flat).max())
            data["err32_norm:" + k] = np.float64(abs(float(np.sqrt((f32 ** 2).sum())) - float(np.sqrt((flat ** 2).sum()))))
            data["ulp_spread:" + k] = np.float64(0.0)
            data["ulp_spread_norm:" + k] = np.float64(0.0)
        for draw in range(GRAD_PERTURBATIONS):
            _, gp = TO.loss_and_grads(ulp_perturbed(params, draw), ob, T, dtype=torch.float64)
            for k, v in g.items():
                a, b = np.asarray(v, dtype=np.float64).reshape(-1), np.asarray(gp[k], dtype=np.float64).reshape(-1)
                idx = grad_sample_index(k, a.size)
                data["ulp_spread:" + k] = np.float64(max(float(data["ulp_spread:" + k]), float(np.abs(a[idx] - b[idx]).max())))
                data["ulp_spread_norm:" + k] = np.float64(max(float(data["ulp_spread_norm:" + k]),
                                                              abs(float(np.sqrt((a ** 2).sum())) - float(np.sqrt((b ** 2).sum())))))
            print("  perturbation %d %.1f s" % (draw, time.time() - t0), flush=True)
        data["ulp_perturbations"] = np.int64(GRAD_PERTURBATIONS)
        np.savez_compressed(os.path.join(OUT, "anchor_grad_%s.npz" % name), **data)
        print("grad anchor", name, "T", T, "M", EV.shape[0], "loss %.9f" % data["loss"], "|g|max %.3e" % gmax,
              "%.1f s" % (time.time() - t0), flush=True)


def gen_trained():
    """trained_d64.npz: the oracle's weights after TRAINED['steps'] Adam steps (float64) -- a network far from its
    initialisers for the parity tests (tests/test_gpu_trained.py)."""
    import time
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    import tspgnn
    from oracle import params as P
    from oracle import torch_oracle as TO

    torch.set_num_threads(os.cpu_count() or 1)
    cfg = TRAINED
    batches = []
    for s in range(cfg["n_batches"]):
        EV, W, C, route_exists, n_vertices, n_edges = tspgnn.synthetic_batch(cfg["sizes"], seed=100 + s)
        batches.append({"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices,
                        "n_edges": n_edges})
    params = {k: np.asarray(v, dtype=np.float64) for k, v in P.init_params(cfg["d"], seed=42).items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v_ = {k: np.zeros_like(v) for k, v in params.items()}
    t0 = time.time()
    for step in range(1, cfg["steps"] + 1):
        out, g = TO.loss_and_grads(params, batches[step % len(batches)], cfg["T"], dtype=torch.float64)
        g, gn = TO.clip_by_global_norm(g)
        params, m, v_ = TO.adam_step(params, g, m, v_, step, lr=cfg["lr"])
        if step % 200 == 0 or step == 1:
            print("trained: step %d loss %.6f |g| %.3e  %.0f s" % (step, out["loss"].item(), gn, time.time() - t0), flush=True)
    init = P.init_params(cfg["d"], seed=42)
    moved = {k: float(np.abs(params[k] - init[k]).max()) for k in params}
    print("largest movement per variable: max %.3f, LayerNorm gains now in [%.3f, %.3f]"
          % (max(moved.values()), min(params[k].min() for k in params if k.endswith("gamma")),
             max(params[k].max() for k in params if k.endswith("gamma"))))
    np.savez_compressed(os.path.join(OUT, "trained_d64.npz"), **{k: v.astype(np.float32) for k, v in params.items()})


def gen_bf16_anchors(only=None):
    """anchor_bf16_*.npz: the bf16-storage oracle (torch_oracle.forward(bf16=True)) and the plain float64 oracle at the
    depth and width of BASELINE config 5 (tests/test_gpu_anchors.py::test_bf16_storage_at_config5_depth)."""
    import time
    import torch
    from oracle import torch_oracle as TO

    torch.set_num_threads(os.cpu_count() or 1)
    for name in BF16_ANCHORS:
        if only and name not in only:
            continue
        batch, params, d, T, finger = bf16_anchor_inputs(name)
        EV, W, C, route_exists, n_vertices, n_edges = batch
        ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
        t0 = time.time()
        data = {"T": np.int64(T), "d": np.int64(d), "fingerprint": finger}
        with torch.no_grad():
            tp = TO.to_torch(params, torch.float64)
            for tag, ref in (("bf16", TO.forward(tp, ob, T, bf16=True)), ("f64", TO.forward(tp, ob, T))):
                data[tag + "_predictions"] = ref["predictions"].numpy()
                data[tag + "_loss"] = np.float64(ref["loss"].item())
                for var in ("E", "V"):
                    for k, part in enumerate(("h", "c")):
                        a = ref["last_states"][var][k].numpy()
                        data["%s_%s%s_rows" % (tag, var, part)] = a[anchor_rows(a.shape[0])].astype(np.float32)
                        data["%s_%s%s_absmax" % (tag, var, part)] = np.float64(np.abs(a).max())
        np.savez_compressed(os.path.join(OUT, "anchor_bf16_%s.npz" % name), **data)
        print("bf16 anchor", name, "d", d, "T", T, "M", EV.shape[0], "loss %.9f" % data["bf16_loss"], "%.0f s" % (time.time() - t0))


def gen_graph():
    """.graph texts parsed by the reference's read_graph.  The texts are written by this repository's writer
    (tspgnn.write_graph: float weights, and dataset.py's integer-binned variant) plus one laid out by hand with the
    format's freedoms: blank-separated header variants, a sparse graph, trailing blanks."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    import tspgnn
    from instance_loader import read_graph  # the reference's parser

    rng = np.random.RandomState(42)
    texts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, n, conn, int_w in (("full_n7", 7, 1.0, False), ("sparse_n12", 12, 0.35, False), ("binned_n6", 6, 1.0, True)):
            Ma, Mw, route = make_instance(n, rng, conn)
            path = os.path.join(tmp, name + ".graph")
            tspgnn.write_graph(Ma, Mw, path, route=route, int_weights=int_w)
            texts[name] = open(path).read()
        texts["hand_n4"] = ("NAME : hand\nTYPE : TSP\nCOMMENT: laid out by hand\nDIMENSION: 4\nEDGE_DATA_FORMAT: EDGE_LIST\n"
                            "EDGE_WEIGHT_TYPE: EXPLICIT\nEDGE_WEIGHT_FORMAT: FULL_MATRIX \nEDGE_DATA_SECTION:\n0 1\n0 3\n1 2\n"
                            "2 3\n-1\nEDGE_WEIGHT_SECTION:\n0 0.5 0 0.25 \n0 0 1.5 0 \n0 0 0 2 \n0 0 0 0 \nTOUR_SECTION:\n"
                            "0 1 2 3 \nEOF\n")
        for name, text in texts.items():
            path = os.path.join(tmp, name + ".graph")
            with open(path, "w") as f:
                f.write(text)
            Ma, Mw, route = read_graph(path)
            np.savez_compressed(os.path.join(OUT, "graph_%s.npz" % name), text=np.array(text), Ma=Ma, Mw=Mw,
                                route=np.array(route, dtype=np.int64))
            print("graph", name, Ma.shape, int(Ma.sum()), "edges, tour", route)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    what = sys.argv[1:] or ["pack", "oracle", "anchors", "graph", "bf16"]
    if "pack" in what:
        gen_pack()
    if "oracle" in what:
        gen_oracle()
    if "anchors" in what:
        gen_anchors()
    if "graph" in what:
        gen_graph()
    if "bf16" in what:
        gen_bf16_anchors()
    if "grads" in what:      # (not in the default list: minutes of float64 autograd at C2 size)
        gen_grad_anchors()
    if "trained" in what:    # (not in the default list: 2 000 oracle training steps)
        gen_trained()
    for w in what:   # "bf16:<name>" / "anchors:<name>": one anchor only (the others take minutes and do not change)
        if w.startswith("bf16:"):
            gen_bf16_anchors(only=w[5:].split(","))
        if w.startswith("anchors:"):
            gen_anchors(only=w[8:].split(","))
        if w.startswith("grads:"):
            gen_grad_anchors(only=w[6:].split(","))
