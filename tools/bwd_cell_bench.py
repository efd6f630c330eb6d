"""The training step's cell-backward launch (tspgnn_lnlstm_bwd_multi_h2: edge cell in gather-init mode with the fused data
gradient + pushed vertex cell) alone, C2 shapes, back to back between two HIP events.  Operands: a real forward's states.
python tools/bwd_cell_bench.py [graphs=128]      (TSPGNN_LIB=tools/variants/X.so for A/B runs)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402
from tspgnn import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
EV, W, C, r, nv, ne = tspgnn.synthetic_batch([40] * B, seed=0)
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer())
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 4, model["route_exists"]: r,
        model["n_vertices"]: nv, model["n_edges"]: ne}
b = sess.prepare(feed)
sess.loss_and_grads(b)     # packs, caches
gnn = model["gnn"]
dev = gnn.store.theta.device
f32 = dict(dtype=torch.float32, device=dev)
M, N, d = EV.shape[0], EV.shape[1], 64
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g, **f32)
adj = b.adj if hasattr(b, "adj") else None
mats = {"EV": adj}
ce, cv = gnn._RNN_cells["E"], gnn._RNN_cells["V"]
h_e, c_e, dh_e, dc_e = rnd(M, d).abs() * 0.3, rnd(M, d), rnd(M, d) * 1e-4, rnd(M, d) * 1e-4
zx = rnd(((N + 15) // 16) * 16, 4 * d) * 64.0
dz_e, ndc_e, ndh_e = torch.empty(M, 4 * d, **f32), torch.empty(M, d, **f32), torch.empty(M, d, **f32)
ws_e = _lib.workspace("tspgnn_lnlstm_bwd_workspace_floats", d, device=dev).zero_()
te = ce.gather_backward_task(adj, zx, h_e, c_e, dh_e, dc_e, dz_e, ndc_e, ws_e, dh_in=ndh_e, defer=True, arith="h2")
mlp = gnn._msg_MLPs[gnn.loop["V"][0]["msg"]]
kp, zb = cv.pushed_bias_pack(mlp, arith="h2")
deg = adj.row_degrees(bool(gnn.loop["V"][0].get("transpose?", False)))
x_v, h_v, c_v, dh_v, dc_v = rnd(N, cv.dx), rnd(N, d).abs() * 0.3, rnd(N, d), rnd(N, d) * 1e-4, rnd(N, d) * 1e-4
dz_v, ndc_v = torch.empty(N, 4 * d, **f32), torch.empty(N, d, **f32)
ws_v = _lib.workspace("tspgnn_lnlstm_bwd_workspace_floats", d, device=dev).zero_()
tv = cv.pushed_backward_task(x_v, h_v, c_v, dh_v, dc_v, dz_v, ndc_v, ws_v, kp, zb, deg, defer=True)
order = [tv, te] if list(gnn.var)[0] == "V" else [te, tv]
tasks = _lib.task_array(order)


def run():
    _lib.call_multi("tspgnn_lnlstm_bwd_multi_h2", tasks, d)


for _ in range(20):
    run()
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("lnlstm_bwd_multi_h2 (E gather + fused dh, V pushed), %d graphs: %.2f us per launch" % (B, e0.elapsed_time(e1) * 5.0))
print("checksum dz_e %.6e dh %.6e" % (float(dz_e.double().abs().sum()), float(ndh_e.double().abs().sum())))
