#!/usr/bin/env python
"""Kernel-level timing at the C2 sizes (M=99840 edge rows, N=5120 vertex rows, d=64): each
entry point in a back-to-back loop between two HIP events.  Development aid (gpurun)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
import tspgnn  # noqa: E402
from tspgnn import _lib  # noqa: E402

dev = torch.device("cuda:0")
d = int(os.environ.get("D", "64"))
nsz = int(os.environ.get("NSZ", "40"))
B = int(os.environ.get("B", "128"))
EV, *_ = tspgnn.synthetic_batch([nsz] * B, seed=1234)
M, N = EV.shape
adj = tspgnn.DeviceAdjacency.from_sparse_ev(EV, dev)


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def pack(W):
    out = torch.empty_like(W)
    _lib.call("tspgnn_pack_weights_f32", _lib.ptr(W), _lib.ptr(out), W.shape[0], W.shape[1], 0, None)
    return out


res = {}
for rows, tag in ((M, "E"), (N, "V")):
    X = torch.randn(rows, d, device=dev)
    H = torch.randn(rows, d, device=dev)
    C = torch.randn(rows, d, device=dev)
    Ho, Co, Y = torch.empty_like(H), torch.empty_like(C), torch.empty_like(X)
    wb = torch.cat([torch.cat([pack(torch.randn(d, d, device=dev) / 8).view(-1), torch.randn(d, device=dev)]) for _ in range(4)])
    K = pack(torch.randn(2 * d, 4 * d, device=dev) / 11)
    ln = torch.cat([torch.ones(d, device=dev), torch.zeros(d, device=dev)] * 5)
    t = timeit(lambda: _lib.call("tspgnn_mlp_fwd_f32", _lib.ptr(X), _lib.ptr(wb), _lib.ptr(Y), None, 0, rows, d, 4, 7, None))
    fl = rows * 4 * 2 * d * d
    res["mlp4_" + tag] = (t, fl / t / 1e6)
    t = timeit(lambda: _lib.call("tspgnn_lnlstm_fwd_f32", _lib.ptr(X), d, _lib.ptr(H), _lib.ptr(C), _lib.ptr(K), _lib.ptr(ln),
                                 _lib.ptr(Ho), _lib.ptr(Co), rows, d, None))
    fl = rows * 2 * 2 * d * 4 * d
    res["lnlstm_" + tag] = (t, fl / t / 1e6)
# folded E update: z = Zx[u] + Zx[v] + h Kh  (and the same GEMM without the gather, dx = 0)
H = torch.randn(M, d, device=dev); C = torch.randn(M, d, device=dev); Ho, Co = torch.empty_like(H), torch.empty_like(C)
Kh = pack(torch.randn(d, 4 * d, device=dev) / 8)
Zx = torch.randn(N, 4 * d, device=dev)
ln = torch.cat([torch.ones(d, device=dev), torch.zeros(d, device=dev)] * 5)
t = timeit(lambda: _lib.call("tspgnn_lnlstm_gather_fwd_f32", _lib.ptr(adj.uv), _lib.ptr(Zx), _lib.ptr(H), _lib.ptr(C), _lib.ptr(Kh),
                             _lib.ptr(ln), _lib.ptr(Ho), _lib.ptr(Co), M, N, d, None))
res["lstm_gather_E"] = (t, M * 2 * d * 4 * d / t / 1e6)
t = timeit(lambda: _lib.call("tspgnn_lnlstm_fwd_f32", None, 0, _lib.ptr(H), _lib.ptr(C), _lib.ptr(Kh), _lib.ptr(ln),
                             _lib.ptr(Ho), _lib.ptr(Co), M, d, None))
res["lstm_dx0_E"] = (t, M * 2 * d * 4 * d / t / 1e6)
Xv = torch.randn(N, d, device=dev); Ze = torch.randn(M, d, device=dev)
Ye = torch.empty(M, d, device=dev); Yv = torch.empty(N, d, device=dev)
rowptr, eid, _ = adj.csr_t
gb = N * d * 4 + 2 * M * 4 + M * d * 4
rb = M * d * 4 + (2 * M + N + 1) * 4 + N * d * 4
t = timeit(lambda: _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(adj.uv), _lib.ptr(Xv), _lib.ptr(Ye), M, N, d, None))
res["gather2"] = (t, gb / t / 1e3)
t = timeit(lambda: _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(rowptr), _lib.ptr(eid), _lib.ptr(Ze), _lib.ptr(Yv), N, M, d, None))
res["rowsum"] = (t, rb / t / 1e3)

# bf16x3 variants (multi entry points; single task and the step's real task pairs)
def pack3(W):
    out = torch.empty(3 * W.numel() * 2, dtype=torch.uint8, device=dev)
    _lib.call("tspgnn_pack_weights_x3", _lib.ptr(W), _lib.ptr(out), W.shape[0], W.shape[1], None)
    return out


if d in (32, 64):
    def mlp3_blocks(n):
        return torch.cat([torch.cat([pack3(torch.randn(d, d, device=dev) / 8), torch.randn(d, device=dev).view(torch.uint8)])
                          for _ in range(n)])
    Xe = torch.randn(M, d, device=dev); Ye3 = torch.empty_like(Xe)
    Xv3 = torch.randn(N, d, device=dev); Yv3 = torch.empty_like(Xv3); Zv3 = torch.empty(N, 4 * d, device=dev)
    wbe, wbv = mlp3_blocks(4), mlp3_blocks(4)
    proj = pack3(torch.randn(d, 4 * d, device=dev) / 8)
    for nl in (4, 3):
        te = _lib.task_array([_lib.MlpTask(_lib.ptr(Xe), _lib.ptr(wbe), _lib.ptr(Ye3), None, 0, M, nl, 7, None, None)])
        t = timeit(lambda: _lib.call_multi("tspgnn_mlp_fwd_multi_x3", te, d))
        res["x3_mlp%d_E" % nl] = (t, M * nl * 2 * d * d / t / 1e6)
    tev = _lib.task_array([_lib.MlpTask(_lib.ptr(Xe), _lib.ptr(wbe), _lib.ptr(Ye3), None, 0, M, 3, 7, None, None),
                           _lib.MlpTask(_lib.ptr(Xv3), _lib.ptr(wbv), _lib.ptr(Yv3), None, 0, N, 4, 7, _lib.ptr(proj), _lib.ptr(Zv3))])
    t = timeit(lambda: _lib.call_multi("tspgnn_mlp_fwd_multi_x3", tev, d))
    res["x3_mlp_step"] = (t, (M * 3 + N * 8) * 2 * d * d / t / 1e6)
    Kh3 = pack3(torch.randn(d, 4 * d, device=dev) / 8)
    Kv3 = pack3(torch.randn(2 * d, 4 * d, device=dev) / 11)
    Hv = torch.randn(N, d, device=dev); Cv = torch.randn(N, d, device=dev); Hvo, Cvo = torch.empty_like(Hv), torch.empty_like(Cv)
    zb = torch.randn(4 * d, device=dev); deg = torch.rand(N, device=dev)
    tg = _lib.task_array([_lib.LstmTask(None, 0, _lib.ptr(H), _lib.ptr(C), _lib.ptr(Kh3), _lib.ptr(ln), _lib.ptr(Ho), _lib.ptr(Co), M,
                                        _lib.ptr(adj.uv), _lib.ptr(Zx), None, None)])
    t = timeit(lambda: _lib.call_multi("tspgnn_lnlstm_fwd_multi_x3", tg, d))
    res["x3_lstm_gather_E"] = (t, M * 2 * d * 4 * d / t / 1e6)
    tv = _lib.task_array([_lib.LstmTask(_lib.ptr(Xv3), d, _lib.ptr(Hv), _lib.ptr(Cv), _lib.ptr(Kv3), _lib.ptr(ln), _lib.ptr(Hvo),
                                        _lib.ptr(Cvo), N, None, None, _lib.ptr(zb), _lib.ptr(deg))])
    t = timeit(lambda: _lib.call_multi("tspgnn_lnlstm_fwd_multi_x3", tv, d))
    res["x3_lstm_V"] = (t, N * 2 * 2 * d * 4 * d / t / 1e6)
    tgv = _lib.task_array([tg[0], tv[0]])
    t = timeit(lambda: _lib.call_multi("tspgnn_lnlstm_fwd_multi_x3", tgv, d))
    res["x3_lstm_step"] = (t, (M + 2 * N) * 2 * d * 4 * d / t / 1e6)
    # fused cell + next step's message MLP (tspgnn_lnlstm_mlp_fwd_multi_x3): edge task, vertex task, both
    Ae = torch.empty(M, d, device=dev); Yv = torch.empty(N, d, device=dev)
    ce = _lib.CellMlpTask(tg[0], _lib.ptr(wbe), 3, 7, _lib.ptr(Ae), None, None)
    cv = _lib.CellMlpTask(tv[0], _lib.ptr(wbv), 4, 7, None, _lib.ptr(proj), _lib.ptr(Zv3))
    for tag, ts in (("E", [ce]), ("V", [cv]), ("step", [ce, cv])):
        arr = _lib.task_array(ts)
        t = timeit(lambda: _lib.call_multi("tspgnn_lnlstm_mlp_fwd_multi_x3", arr, d))
        res["x3_fused_" + tag] = (t, 0.0)
for k, (t, r) in res.items():
    unit = "GB/s" if k in ("gather2", "rowsum") else "TFLOP/s"
    print("%-12s %8.2f us  %8.1f %s" % (k, t, r, unit))
