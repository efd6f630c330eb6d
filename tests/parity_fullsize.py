#!/usr/bin/env python
"""One-off full-size parity checks: the HIP forward (both GEMM arithmetics) against the float64 index-form oracle on the
host at the headline configuration (C2: n=40, B=128, d=64, T=32; the float64 oracle takes ~14 min) or, with WORKLOAD=c4,
at BASELINE configs[3] (ragged n in 20..80, B=512; choose a small T).  GRADS=1 adds the gradients of one training
step (float64 autograd on the host).  Too slow for the test suite."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it uses the oracle)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
sys.path.insert(0, ROOT)
import tspgnn  # noqa: E402
from oracle import params as P  # noqa: E402
from oracle import torch_oracle as TO  # noqa: E402

d, T = 64, int(os.environ.get("T", 32))
workload = os.environ.get("WORKLOAD", "c2")
sizes = [40] * 128 if workload == "c2" else np.random.RandomState(0).randint(20, 81, size=512)
batch = tspgnn.synthetic_batch(sizes, seed=1234)
EV, W, C, r, nv, ne = batch
params = P.init_params(d, seed=0)
ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": r, "n_vertices": nv, "n_edges": ne}
t0 = time.time()
torch.set_num_threads(os.cpu_count() or 1)
ref = TO.forward(TO.to_torch(params, torch.float64), ob, T)
t_ref = time.time() - t0
out = {"workload": "%s, T=%d, N=%d, M=%d" % (workload, T, EV.shape[1], EV.shape[0]), "oracle_seconds": round(t_ref, 1)}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


for gemm in ("bf16x3", "f32"):
    model = tspgnn.build_network(d)
    model["gnn"].gemm = gemm
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
            model["n_vertices"]: nv, model["n_edges"]: ne}
    pred, last, loss = sess.run([model["predictions"], model["last_states"], model["loss"]], feed_dict=feed)
    out[gemm] = {"predictions": rel(pred, ref["predictions"].numpy()), "E.h": rel(last["E"].h, ref["last_states"]["E"][0].numpy()),
                 "E.c": rel(last["E"].c, ref["last_states"]["E"][1].numpy()), "V.h": rel(last["V"].h, ref["last_states"]["V"][0].numpy()),
                 "loss_abs": abs(float(loss) - ref["loss"].item())}
    if gemm == "bf16x3" and os.environ.get("GRADS"):
        t0 = time.time()
        _, ref_g = TO.loss_and_grads(params, ob, T, dtype=torch.float64)
        sess.loss_and_grads(feed)
        torch.cuda.synchronize()
        g = model.store.grad_dict()
        l2 = {k: TO.L2NORM_SCALING * params[k] for k in params}
        gscale = max(np.abs(ref_g[k] - l2[k]).max() for k in ref_g)
        _, f32_g = TO.loss_and_grads(params, ob, T, dtype=torch.float32)      # what an fp32 autograd run loses itself
        per = {}
        for k in ref_g:
            rg = ref_g[k] - l2[k]
            scale = max(np.abs(rg).max(), 1e-3 * gscale)
            per[k] = (float(np.abs(g[k] - rg).max() / scale), float(np.abs(f32_g[k] - ref_g[k]).max() / scale))
        top = sorted(per.items(), key=lambda kv: -kv[1][0])[:4]
        out["grads"] = {"worst_per_variable_rel": top[0][1][0], "oracle_seconds": round(time.time() - t0, 1),
                        "top": [{"var": k, "hip": round(a, 9), "fp32_oracle": round(b, 9)} for k, (a, b) in top],
                        "median_hip": float(np.median([a for a, _ in per.values()])),
                        "median_fp32_oracle": float(np.median([b for _, b in per.values()]))}
print(json.dumps(out))
