import os, sys
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tsp-gnn_amd")
import tspgnn
from oracle.anchors import grad_anchor_inputs, grad_sample_index
for name in os.environ.get("ANCHORS", "c2,c1").split(","):
    z = np.load(os.path.join(ROOT, "tests/golden/anchor_grad_%s.npz" % name))
    batch, params, T, finger = grad_anchor_inputs(name)
    EV, W, C, r, nv, ne = batch
    for gemm in os.environ.get("GEMMS", "f16x2,f32").split(","):
        model = tspgnn.build_network(64); model["gnn"].gemm = gemm
        sess = tspgnn.Session(model); sess.run(tspgnn.global_variables_initializer()); model.store.load(params)
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r, model["n_vertices"]: nv, model["n_edges"]: ne}
        out = sess.loss_and_grads(feed); torch.cuda.synchronize()
        g = model.store.grad_dict(); gscale = float(z["grad_absmax"])
        rows = []
        for k in g:
            idx = grad_sample_index(k, params[k].size)
            ref = z["sample:" + k] - 1e-10 * np.asarray(params[k], dtype=np.float64).reshape(-1)[idx]
            got = np.asarray(g[k], dtype=np.float64).reshape(-1)[idx]
            scale = max(float(z["absmax:" + k]), 1e-3 * gscale)
            rows.append((float(np.abs(got - ref).max()) / scale, float(z["err32:" + k]) / scale,
                         float(z["ulp_spread:" + k]) / scale if "ulp_spread:" + k in z.files else float("nan"), k))
        rows.sort(reverse=True)
        print("==", name, gemm, "T", T)
        for e, e32, sp, k in rows[:int(os.environ.get("TOP", "12"))]:
            print("  err %.2e  fp32-restatement %.2e  one-ulp spread %.2e  err / max(2 e32, 2 spread, 1e-5) = %.2f  %s"
                  % (e, e32, sp, e / max(2 * e32, 2 * sp, 1e-5), k))
        print("  worst err / bar over all variables: %.2f" % max(e / max(2 * e32, 2 * sp, 1e-5) for e, e32, sp, _ in rows))
