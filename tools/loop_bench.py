"""Forward pass (replayed HIP graph) with the one-launch loop (tspgnn_mp_loop_h2) and with the stepwise launches, same box,
alternating; checks bit equality first.  python tools/loop_bench.py [graphs=128] [n=40] [T=32] [reps=5]
LOOP_BENCH_MODES=both|loop|steps; TSPGNN_LOOP_MAX_TILES=4 lets the loop take batches of 4 tiles per wavefront (C2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402
from oracle import params as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
modes = {"both": (True, False), "loop": (True,), "steps": (False,)}[os.environ.get("LOOP_BENCH_MODES", "both")]
t = tspgnn.synthetic_batch([n] * B, seed=0)
params = P.init_params(64, seed=1, perturb=True)
replays, outs = {}, {}
for loop in modes:
    model = tspgnn.build_network(64)
    model["gnn"].persistent_loop = loop
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, r, nv, ne = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
            model["n_vertices"]: nv, model["n_edges"]: ne}
    b = sess.prepare(feed)
    print("loop" if loop else "steps", "plan:", None if b.adj.loop_plan is None else b.adj.loop_plan[1:], flush=True)
    rp = sess.capture_forward(b)     # (the closure keeps the batch alive: the graph reads its buffers on every replay)
    out = rp()
    torch.cuda.synchronize()
    assert not sess.range_exceeded()
    outs[loop] = {"pred": out["predictions"].clone(), "Eh": out["last_states"]["E"].h.clone(),
                  "Vc": out["last_states"]["V"].c.clone()}
    replays[loop] = rp
for k in (outs[True] if len(modes) == 2 else ()):
    same = torch.equal(outs[True][k], outs[False][k])
    print("bit-equal %s: %s  (max abs diff %.3e)" % (k, same, float((outs[True][k] - outs[False][k]).abs().max())), flush=True)
for rep in range(reps):
    for loop in modes:
        rp = replays[loop]
        for _ in range(3):
            rp()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rp()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-5s %.4f ms per forward = %.2f us per step" % ("loop" if loop else "steps", ms, ms * 1e3 / T), flush=True)
