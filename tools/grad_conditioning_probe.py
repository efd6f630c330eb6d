#!/usr/bin/env python
"""How ill-conditioned are the full-size gradients the anchors compare?  (TEST INFRASTRUCTURE: uses the oracle.)

The C2 / T = 2 gradient anchor leaves the f16x2 training path at 8.4e-4 on a few bias-like variables where the fp32
autograd restatement loses 3.5e-5.  This probe repeats the comparison on a NEIGHBOURING problem: every weight
matrix rounded to the 22 significand bits two fp16 pieces carry (a relative change <= 1e-7), oracle and HIP path on the
same rounded weights -- so the f16x2 weight packing is exact there.

    python tools/grad_conditioning_probe.py gen    # build container (CPU, float64 oracle, ~15 s): tools/tmp_w22.npz
    python tools/grad_conditioning_probe.py run    # GPU box: errors of f16x2 and bf16x3 against that reference
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
from oracle.anchors import grad_anchor_inputs, grad_sample_index  # noqa: E402

TMP = os.path.join(ROOT, "tools", "tmp_w22.npz")


def r22(w):
    w = np.asarray(w, dtype=np.float32)
    x = (w * np.float32(64.0)).astype(np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return ((hi + lo) / np.float32(64.0)).astype(np.float32)


def gen():
    from oracle import torch_oracle as TO
    torch.set_num_threads(os.cpu_count() or 1)
    batch, params, T, _ = grad_anchor_inputs("c2")
    p22 = {k: (r22(v) if np.asarray(v).ndim == 2 else np.asarray(v, dtype=np.float32)) for k, v in params.items()}
    print("largest relative change of a weight: %.2e" % max(float(np.abs(p22[k] - params[k]).max() / max(np.abs(params[k]).max(), 1e-30)) for k in params))
    EV, W, C, route_exists, n_vertices, n_edges = batch
    ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
    out, g = TO.loss_and_grads(p22, ob, T, dtype=torch.float64)
    data = {"grad_absmax": np.float64(max(float(np.abs(v).max()) for v in g.values()))}
    for k, v in g.items():
        flat = np.asarray(v, dtype=np.float64).reshape(-1)
        data["absmax:" + k] = np.float64(np.abs(flat).max())
        data["sample:" + k] = flat[grad_sample_index(k, flat.size)]
        data["param:" + k] = p22[k]
    np.savez_compressed(TMP, **data)
    print("wrote", TMP)


def run():
    import tspgnn
    z = np.load(TMP)
    z0 = np.load(os.path.join(ROOT, "tests/golden/anchor_grad_c2.npz"))
    batch, params, T, _ = grad_anchor_inputs("c2")
    EV, W, C, r, nv, ne = batch
    for label, ref, weights in (("the anchor's weights", z0, params), ("weights rounded to 22 bits", z, {k: z["param:" + k] for k in params})):
        for gemm in ("f16x2", "bf16x3"):
            model = tspgnn.build_network(64)
            model["gnn"].gemm = gemm
            sess = tspgnn.Session(model)
            sess.run(tspgnn.global_variables_initializer())
            model.store.load(weights)
            feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
                    model["n_vertices"]: nv, model["n_edges"]: ne}
            sess.loss_and_grads(feed)
            torch.cuda.synchronize()
            g = model.store.grad_dict()
            gscale = float(ref["grad_absmax"])
            rows = []
            for k in g:
                idx = grad_sample_index(k, params[k].size)
                want = ref["sample:" + k] - 1e-10 * np.asarray(weights[k], dtype=np.float64).reshape(-1)[idx]
                got = np.asarray(g[k], dtype=np.float64).reshape(-1)[idx]
                scale = max(float(ref["absmax:" + k]), 1e-3 * gscale)
                rows.append((float(np.abs(got - want).max()) / scale, k))
            rows.sort(reverse=True)
            print("== C2, T = 2, %s, %s: worst variables" % (label, gemm))
            for e, k in rows[:4]:
                print("   %.2e  %s" % (e, k))


if __name__ == "__main__":
    {"gen": gen, "run": run}[sys.argv[1]]()
