"""Column sums over many rows of what the LN-LSTM backward kernels emit (dz: the operand of every weight / bias gradient
upstream; ln_grad: the LayerNorm parameter gradients) against float64 autograd, f32 vs f16x2 kernels, same inputs.
Round 5: at full C2 size the f16x2 training step's bias / LayerNorm-shift gradients are 5x further from float64 than the
fp32-MFMA backward's; this probe asks whether one cell-backward launch already shows it.  python tools/cell_bwd_colsum_probe.py [rows]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import torch_oracle as TO  # noqa: E402
from tspgnn import _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 99840
d = dx = 64
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
x, h, c = f32(rng.randn(rows, dx) * 3), f32(np.maximum(rng.randn(rows, d), 0)), f32(rng.randn(rows, d))
K = f32(rng.randn(dx + d, 4 * d) / np.sqrt(dx + d))
ln = f32(np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]))
dh_o, dc_o = f32(1e-4 * rng.randn(rows, d) * np.exp(rng.randn(rows, 1))), f32(1e-4 * rng.randn(rows, d) * np.exp(rng.randn(rows, 1)))
tx, th, tc, tK = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, h, c, K))
tln = torch.tensor(ln, dtype=torch.float64, requires_grad=True)
names = ("input", "transform", "forget", "output", "state")
params = {"TSP/Q_cell/layer_norm_basic_lstm_cell/kernel": tK}
for i, g in enumerate(names):
    params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/gamma" % g] = tln[i, 0]
    params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/beta" % g] = tln[i, 1]
torch.set_num_threads(os.cpu_count())
nh, nc = TO.lnlstm_cell(tx, th, tc, params, "Q")
z = torch.cat([tx, th], 1) @ tK
z.retain_grad()
loss = (nh * torch.tensor(dh_o, dtype=torch.float64)).sum() + (nc * torch.tensor(dc_o, dtype=torch.float64)).sum()
gln, gK = torch.autograd.grad(loss, [tln, tK])
ref_ln = gln.numpy().reshape(-1)
# colsum(dz) = gradient of a bias added to z = ones^T dz; from dK = [x|h]^T dz we cannot get it: recompute via autograd on a bias
tb = torch.zeros(4 * d, dtype=torch.float64, requires_grad=True)


def cell_with_bias():
    zz = torch.cat([tx, th], 1) @ tK + tb
    i, j, f, o = torch.split(zz, d, dim=1)
    lnf = lambda v, k: TO.layer_norm(v, tln[k, 0], tln[k, 1])
    i, j, f, o = lnf(i, 0), lnf(j, 1), lnf(f, 2), lnf(o, 3)
    g = torch.relu(j)
    ncc = tc * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * g
    ncc = TO.layer_norm(ncc, tln[4, 0], tln[4, 1])
    return torch.relu(ncc) * torch.sigmoid(o), ncc


nh2, nc2 = cell_with_bias()
assert float((nh2 - nh).abs().max()) < 1e-12
ref_b = torch.autograd.grad((nh2 * torch.tensor(dh_o, dtype=torch.float64)).sum() + (nc2 * torch.tensor(dc_o, dtype=torch.float64)).sum(), [tb])[0].numpy()


def to(a, dt=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)


def run(arith):
    src = to(K)
    if arith == "f32":
        Kp = torch.empty_like(src)
        _lib.call("tspgnn_pack_weights_f32", _lib.ptr(src), _lib.ptr(Kp), dx + d, 4 * d, 0, None)
    else:
        Kp = torch.empty(4 * (dx + d) * 4 * d, dtype=torch.uint8, device=dev)
        _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src), _lib.ptr(Kp), dx + d, 4 * d, None, None)
    dz, dc_in = torch.empty((rows, 4 * d), device=dev), torch.empty((rows, d), device=dev)
    ln_grad = torch.zeros(10 * d, device=dev)
    n = int(_lib.lib.tspgnn_lnlstm_bwd_workspace_floats(d))
    wsl = torch.empty(max(n, 1), device=dev)
    keep = [to(x), to(h), to(c), to(ln), to(dh_o), to(dc_o)]
    task = _lib.LstmBwdTask(_lib.ptr(keep[0]), dx, _lib.ptr(keep[1]), _lib.ptr(keep[2]), _lib.ptr(Kp), _lib.ptr(keep[3]),
                            _lib.ptr(keep[4]), _lib.ptr(keep[5]), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad), _lib.ptr(wsl),
                            rows, None, None, None, None, 0, None, None)
    _lib.call_multi("tspgnn_lnlstm_bwd_multi_" + arith, [task], d)
    torch.cuda.synchronize()
    return dz.double().sum(0).cpu().numpy(), ln_grad.double().cpu().numpy(), dz.cpu().numpy()


out = {a: run(a) for a in ("f32", "h2")}
for a in ("f32", "h2"):
    cs, lg, dzv = out[a]
    e_b = np.abs(cs - ref_b).max() / np.abs(ref_b).max()
    e_ln = np.abs(lg - ref_ln).max() / np.abs(ref_ln).max()
    print("%-3s  colsum(dz) rel err %.2e   ln_grad rel err %.2e   (rows %d; |colsum| max %.3e, sum|dz| per column ~%.3e)"
          % (a, e_b, e_ln, rows, np.abs(ref_b).max(), np.abs(dzv).sum(0).mean()))
print("per-row |dz_h2 - dz_f32| / rowmax: max %.2e  mean %.2e" % (
    float((np.abs(out["h2"][2] - out["f32"][2]) / np.abs(out["f32"][2]).max(1, keepdims=True)).max()),
    float((np.abs(out["h2"][2] - out["f32"][2]) / np.abs(out["f32"][2]).max(1, keepdims=True)).mean())))
