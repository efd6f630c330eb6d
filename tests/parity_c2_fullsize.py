#!/usr/bin/env python
"""One-off full-size parity check at the headline configuration (C2: n=40, B=128, d=64, T=32): the HIP forward (both GEMM
arithmetics) against the float64 index-form oracle on the host.  Too slow for the test suite (the float64 oracle takes ~14 min on the host)."""
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
batch = tspgnn.synthetic_batch([40] * 128, seed=1234)
EV, W, C, r, nv, ne = batch
params = P.init_params(d, seed=0)
ob = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": r, "n_vertices": nv, "n_edges": ne}
t0 = time.time()
torch.set_num_threads(os.cpu_count() or 1)
ref = TO.forward(TO.to_torch(params, torch.float64), ob, T)
t_ref = time.time() - t0
out = {"workload": "c2, T=%d" % T, "oracle_seconds": round(t_ref, 1)}


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
print(json.dumps(out))
