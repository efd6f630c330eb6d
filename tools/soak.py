#!/usr/bin/env python
"""Soak (development aid, gpurun): N replays of the captured C2 forward graph must be bit-identical, then K captured
training steps must keep the loss finite and move the weights; finally the forward is replayed again and compared with
an eager forward on the trained weights."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402

n_fwd, n_train = int(os.environ.get("NFWD", 3000)), int(os.environ.get("NTRAIN", 300))
d, T = 64, 32
# GRAPHS=192 puts the forward on tspgnn_mp_resident_h2 (the default selector's window); TSPGNN_LOOP_KIND forces a form
batch = tspgnn.synthetic_batch([40] * int(os.environ.get("GRAPHS", 128)), seed=1234)
EV, W, C, r, nv, ne = batch
model = tspgnn.build_network(d)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer(seed=0))
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
        model["n_vertices"]: nv, model["n_edges"]: ne}
b = sess.prepare(feed)
print("one-launch form:", None if b.adj.loop_plan is None else b.adj.loop_plan[3])
replay = sess.capture_forward(b)
first = replay()
ref_pred = first["predictions"].clone()
ref_h = first["last_states"]["E"].h.clone()
bad = 0
for i in range(n_fwd):
    o = replay()
    if i % 50 == 0:
        bad += int(not torch.equal(o["predictions"], ref_pred)) + int(not torch.equal(o["last_states"]["E"].h, ref_h))
torch.cuda.synchronize()
print("forward replays: %d, mismatching checks: %d" % (n_fwd, bad))
step = sess.capture_train_step(b)
theta0 = model.store.theta.clone()
losses = []
for i in range(n_train):
    o = step()
    if i % 20 == 0:
        losses.append(float(o["stats"][0].item()))
torch.cuda.synchronize()
moved = float((model.store.theta - theta0).abs().max())
print("train steps: %d, losses %s ... finite: %s, max |dtheta| %.3e" % (n_train, ["%.5f" % x for x in losses[:3] + losses[-2:]],
                                                                    bool(np.all(np.isfinite(losses))), moved))
eager = sess.forward_device(b)["predictions"].clone()
again = sess.capture_forward(b)()["predictions"]
print("after training: captured forward == eager forward:", bool(torch.equal(eager, again)))
assert bad == 0 and np.all(np.isfinite(losses)) and moved > 0 and torch.equal(eager, again)
print("soak OK")
