"""Backward HIP kernels vs torch-autograd (float64, CPU) on the oracle's restatement of each op."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import h2_zx_pack, rel_err
from oracle import torch_oracle as TO
from tspgnn import _lib

pytestmark = pytest.mark.gpu

_KEEP = []
TOL = 5e-6


def dev(a, device, dtype=np.float32):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)
    _KEEP.append(t)
    return t


def empty(shape, device, fill=None):
    t = torch.empty(shape, dtype=torch.float32, device=device) if fill is None else \
        torch.full(shape, fill, dtype=torch.float32, device=device)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def packed(W, device, transposed=0):
    """pack(W) or, with transposed=1, pack(W^T) from the [out,in]... i.e. W given as stored."""
    src = dev(W, device)
    out = empty(src.shape, device)
    kr, nc = (W.shape[1], W.shape[0]) if transposed else W.shape
    _lib.call("tspgnn_pack_weights_f32", _lib.ptr(src), _lib.ptr(out), kr, nc, transposed, None)
    return out


def ws(name, *args, device):
    n = getattr(_lib.lib, name)(*args)
    return empty((max(int(n), 1),), device)


@pytest.mark.parametrize("kin,n1,n2,rows,acc", [(256, 64, 64, 1000, 0), (128, 32, 32, 77, 1), (512, 128, 128, 300, 1),
                                                (64, 0, 64, 50, 0), (256, 48, 16, 33, 0)])
def test_linear(cuda_device, kin, n1, n2, rows, acc):
    rng = np.random.RandomState(kin + rows)
    X = rng.randn(rows, kin).astype(np.float32)
    W = (rng.randn(kin, n1 + n2) / np.sqrt(kin)).astype(np.float32)
    Y2_0 = rng.randn(rows, n2).astype(np.float32)
    Y1 = empty((rows, max(n1, 1)), cuda_device)
    Y2 = dev(Y2_0, cuda_device)
    _lib.call("tspgnn_linear_f32", _lib.ptr(dev(X, cuda_device)), kin, _lib.ptr(packed(W, cuda_device)),
              _lib.ptr(Y1) if n1 else None, n1, _lib.ptr(Y2), n2, acc, rows, None)
    torch.cuda.synchronize()
    ref = X.astype(np.float64) @ W.astype(np.float64)
    if n1:
        assert rel_err(Y1.cpu().numpy()[:, :n1], ref[:, :n1]) < TOL
    assert rel_err(Y2.cpu().numpy(), ref[:, n1:] + (Y2_0 if acc else 0)) < TOL


def test_pack_transposed(cuda_device):
    W = np.random.RandomState(0).randn(128, 256).astype(np.float32)        # stored [ncols=128? no: stored as W]
    a = packed(np.ascontiguousarray(W.T), cuda_device).cpu().numpy()         # pack(W^T) from an explicit transpose
    b = packed(W, cuda_device, transposed=1).cpu().numpy()                   # pack(W^T) straight from W
    assert np.array_equal(a.reshape(-1), b.reshape(-1))


@pytest.mark.parametrize("d,dx,rows,null_grads", [(64, 64, 333, False), (32, 32, 50, False), (64, 64, 16, True),
                                                   (128, 128, 120, False), (32, 96, 100, False)])
def test_lnlstm_backward_chain(cuda_device, d, dx, rows, null_grads):
    rng = np.random.RandomState(d + rows)
    x = rng.randn(rows, dx); h = rng.randn(rows, d); c = rng.randn(rows, d)
    K = rng.randn(dx + d, 4 * d) / np.sqrt(dx + d)
    ln = np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)])
    dh_o = rng.randn(rows, d); dc_o = rng.randn(rows, d)
    f32 = lambda a: a.astype(np.float32)
    x, h, c, K, ln, dh_o, dc_o = map(f32, (x, h, c, K, ln, dh_o, dc_o))
    # --- oracle: autograd through torch_oracle.lnlstm_cell
    tx, th, tc, tK = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, h, c, K))
    tln = torch.tensor(ln, dtype=torch.float64, requires_grad=True)
    names = ("input", "transform", "forget", "output", "state")
    params = {"TSP/Q_cell/layer_norm_basic_lstm_cell/kernel": tK}
    for i, g in enumerate(names):
        params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/gamma" % g] = tln[i, 0]
        params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/beta" % g] = tln[i, 1]
    nh, nc = TO.lnlstm_cell(tx, th, tc, params, "Q")
    if null_grads:
        loss = (nh * torch.tensor(dh_o, dtype=torch.float64)).sum()
    else:
        loss = (nh * torch.tensor(dh_o, dtype=torch.float64)).sum() + (nc * torch.tensor(dc_o, dtype=torch.float64)).sum()
    gx, gh, gc, gK, gln = torch.autograd.grad(loss, [tx, th, tc, tK, tln])
    # --- HIP: lnlstm_bwd -> dz, dc; linear(dz, K^T) -> dx, dh; wgrad([x|h], dz) -> dK
    dz = empty((rows, 4 * d), cuda_device); dc_in = empty((rows, d), cuda_device)
    ln_grad = empty((10 * d,), cuda_device, 0.0)
    wsl = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
    xd, hd, cd = dev(x, cuda_device), dev(h, cuda_device), dev(c, cuda_device)
    _lib.call("tspgnn_lnlstm_bwd_f32", _lib.ptr(xd), dx, _lib.ptr(hd), _lib.ptr(cd), _lib.ptr(packed(K, cuda_device)),
              _lib.ptr(dev(ln, cuda_device)), _lib.ptr(dev(dh_o, cuda_device)),
              None if null_grads else _lib.ptr(dev(dc_o, cuda_device)), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad),
              _lib.ptr(wsl), rows, d, None)
    dxo = empty((rows, dx), cuda_device); dho = empty((rows, d), cuda_device)
    _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * d, _lib.ptr(packed(K, cuda_device, transposed=1)), _lib.ptr(dxo), dx,
              _lib.ptr(dho), d, 0, rows, None)
    dK = empty((dx + d, 4 * d), cuda_device, 0.0)
    w1 = ws("tspgnn_wgrad_workspace_floats", rows, dx, 4 * d, device=cuda_device)
    w2 = ws("tspgnn_wgrad_workspace_floats", rows, d, 4 * d, device=cuda_device)
    _lib.call("tspgnn_wgrad_f32", _lib.ptr(xd), _lib.ptr(dz), rows, dx, 4 * d, _lib.ptr(dK[:dx]), None, _lib.ptr(w1), None)
    _lib.call("tspgnn_wgrad_f32", _lib.ptr(hd), _lib.ptr(dz), rows, d, 4 * d, _lib.ptr(dK[dx:]), None, _lib.ptr(w2), None)
    torch.cuda.synchronize()
    assert rel_err(dc_in.cpu().numpy(), gc.numpy()) < TOL
    assert rel_err(dxo.cpu().numpy(), gx.numpy()) < TOL
    assert rel_err(dho.cpu().numpy(), gh.numpy()) < TOL
    assert rel_err(dK.cpu().numpy(), gK.numpy()) < TOL
    assert rel_err(ln_grad.cpu().numpy().reshape(5, 2, d), gln.numpy()) < TOL


@pytest.mark.parametrize("rows", [1, 333, 9000])
def test_lnlstm_gather_backward_fused_dh(cuda_device, rows):
    """tspgnn_lstm_bwd_task.KT / .dxh (d = 64, gather-init mode): dh = dz Kh^T formed in the cell launch equals the
    separate tspgnn_linear_f32 on the stored dz, and the other outputs do not change."""
    d, N = 64, 257
    rng = np.random.RandomState(rows)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    uv = np.stack([rng.randint(0, N, rows), rng.randint(0, N, rows)], 1).astype(np.int32)
    Zx, h, c = f32(rng.randn(N, 4 * d)), f32(rng.randn(rows, d)), f32(rng.randn(rows, d))
    Kh = f32(rng.randn(d, 4 * d) / np.sqrt(d))
    ln = f32(np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]))
    dh_o, dc_o = f32(rng.randn(rows, d)), f32(rng.randn(rows, d))
    args = dict(uv=dev(uv, cuda_device, np.int32), Zx=dev(Zx, cuda_device), h=dev(h, cuda_device), c=dev(c, cuda_device),
                K=packed(Kh, cuda_device), KT=packed(Kh, cuda_device, transposed=1), ln=dev(ln, cuda_device),
                dh=dev(dh_o, cuda_device), dc=dev(dc_o, cuda_device))
    outs = []
    for fused in (False, True):
        dz, dc_in, dh_in = empty((rows, 4 * d), cuda_device), empty((rows, d), cuda_device), empty((rows, d), cuda_device)
        ln_grad = empty((10 * d,), cuda_device, 0.0)
        wsl = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
        task = _lib.LstmBwdTask(None, 0, _lib.ptr(args["h"]), _lib.ptr(args["c"]), _lib.ptr(args["K"]), _lib.ptr(args["ln"]),
                                _lib.ptr(args["dh"]), _lib.ptr(args["dc"]), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad),
                                _lib.ptr(wsl), rows, _lib.ptr(args["uv"]), _lib.ptr(args["Zx"]),
                                _lib.ptr(args["KT"]) if fused else None, _lib.ptr(dh_in) if fused else None)
        _lib.call_multi("tspgnn_lnlstm_bwd_multi_f32", [task], d)
        if not fused:
            _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * d, _lib.ptr(args["KT"]), None, 0, _lib.ptr(dh_in), d, 0, rows, None)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (dz, dc_in, dh_in, ln_grad)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)      # same arithmetic in the same order, from registers instead of from HBM


def packed_h2(W, device):
    """tspgnn_pack_weights_h2 (two fp16 pieces of 2^s W) as a byte tensor."""
    src = dev(W, device)
    out = torch.empty(4 * W.size, dtype=torch.uint8, device=device)
    _KEEP.append(out)
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src), _lib.ptr(out), W.shape[0], W.shape[1], None, None)
    return out


@pytest.mark.parametrize("d,dx,rows,grad_scale", [(64, 64, 333, 1.0), (32, 32, 50, 1.0), (64, 64, 1000, 1e-6), (32, 32, 100, 1e3),
                                                   (64, 0, 77, 1.0)])
def test_lnlstm_backward_h2_vs_autograd(cuda_device, d, dx, rows, grad_scale):
    """tspgnn_lnlstm_bwd_multi_h2 (both GEMMs on the fp16 matrix cores): dz, dc, the LayerNorm gradients, and the
    downstream [dx | dh] = dz K^T and dK = [x|h]^T dz against float64 autograd -- also with incoming gradients of
    magnitude 1e-6 and 1e3 (nothing in the kernel may depend on the scale of the gradient)."""
    rng = np.random.RandomState(d + rows)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    x, h, c = f32(rng.randn(rows, dx)), f32(rng.randn(rows, d)), f32(rng.randn(rows, d))
    K = f32(rng.randn(dx + d, 4 * d) / np.sqrt(dx + d))
    ln = f32(np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]))
    dh_o, dc_o = f32(grad_scale * rng.randn(rows, d)), f32(grad_scale * rng.randn(rows, d))
    tx, th, tc, tK = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, h, c, K))
    tln = torch.tensor(ln, dtype=torch.float64, requires_grad=True)
    params = {"TSP/Q_cell/layer_norm_basic_lstm_cell/kernel": tK}
    for i, gname in enumerate(("input", "transform", "forget", "output", "state")):
        params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/gamma" % gname] = tln[i, 0]
        params["TSP/Q_cell/layer_norm_basic_lstm_cell/%s/beta" % gname] = tln[i, 1]
    nh, nc = TO.lnlstm_cell(tx, th, tc, params, "Q")
    loss = (nh * torch.tensor(dh_o, dtype=torch.float64)).sum() + (nc * torch.tensor(dc_o, dtype=torch.float64)).sum()
    gx, gh, gc, gK, gln = torch.autograd.grad(loss, [tx, th, tc, tK, tln])
    dz, dc_in = empty((rows, 4 * d), cuda_device), empty((rows, d), cuda_device)
    ln_grad = empty((10 * d,), cuda_device, 0.0)
    wsl = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
    xd, hd, cd = (dev(x, cuda_device) if dx else None), dev(h, cuda_device), dev(c, cuda_device)
    task = _lib.LstmBwdTask(_lib.ptr(xd), dx, _lib.ptr(hd), _lib.ptr(cd), _lib.ptr(packed_h2(K, cuda_device)),
                            _lib.ptr(dev(ln, cuda_device)), _lib.ptr(dev(dh_o, cuda_device)), _lib.ptr(dev(dc_o, cuda_device)),
                            _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad), _lib.ptr(wsl), rows, None, None, None, None, 0)
    _lib.call_multi("tspgnn_lnlstm_bwd_multi_h2", [task], d)
    dxo = empty((rows, max(dx, 1)), cuda_device); dho = empty((rows, d), cuda_device)
    _lib.call("tspgnn_linear_f32", _lib.ptr(dz), 4 * d, _lib.ptr(packed(K, cuda_device, transposed=1)),
              _lib.ptr(dxo) if dx else None, dx, _lib.ptr(dho), d, 0, rows, None)
    dK = empty((dx + d, 4 * d), cuda_device, 0.0)
    w2 = ws("tspgnn_wgrad_workspace_floats", rows, d, 4 * d, device=cuda_device)
    if dx:
        w1 = ws("tspgnn_wgrad_workspace_floats", rows, dx, 4 * d, device=cuda_device)
        _lib.call("tspgnn_wgrad_f32", _lib.ptr(xd), _lib.ptr(dz), rows, dx, 4 * d, _lib.ptr(dK[:dx]), None, _lib.ptr(w1), None)
    _lib.call("tspgnn_wgrad_f32", _lib.ptr(hd), _lib.ptr(dz), rows, d, 4 * d, _lib.ptr(dK[dx:]), None, _lib.ptr(w2), None)
    torch.cuda.synchronize()
    assert rel_err(dc_in.cpu().numpy(), gc.numpy()) < TOL
    if dx:
        assert rel_err(dxo.cpu().numpy(), gx.numpy()) < TOL
    assert rel_err(dho.cpu().numpy(), gh.numpy()) < TOL
    assert rel_err(dK.cpu().numpy(), gK.numpy()) < TOL
    assert rel_err(ln_grad.cpu().numpy().reshape(5, 2, d), gln.numpy()) < TOL


@pytest.mark.parametrize("rows", [1, 333, 9000])
def test_lnlstm_gather_backward_h2_fused_dh(cuda_device, rows):
    """Gather-init mode of tspgnn_lnlstm_bwd_multi_h2 with the fused data gradient dh = dz Kh^T: against the fp32-MFMA
    kernel on the same inputs (Zx scaled by 2^s for the f16x2 side), with the rows' incoming gradients spread over
    twelve orders of magnitude -- the per-row power-of-two normalisation must keep every row at fp32-class accuracy."""
    d, N = 64, 257
    rng = np.random.RandomState(rows)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    sc = float(_lib.lib.tspgnn_h2_weight_scale())
    uv = np.stack([rng.randint(0, N, rows), rng.randint(0, N, rows)], 1).astype(np.int32)
    Zx, h, c = f32(rng.randn(N, 4 * d)), f32(rng.randn(rows, d)), f32(rng.randn(rows, d))
    Kh = f32(rng.randn(d, 4 * d) / np.sqrt(d))
    ln = f32(np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]))
    row_scale = 10.0 ** rng.uniform(-9, 3, size=(rows, 1))
    dh_o, dc_o = f32(row_scale * rng.randn(rows, d)), f32(row_scale * rng.randn(rows, d))
    common = dict(uv=dev(uv, cuda_device, np.int32), h=dev(h, cuda_device), c=dev(c, cuda_device), ln=dev(ln, cuda_device),
                  dh=dev(dh_o, cuda_device), dc=dev(dc_o, cuda_device))
    outs = {}
    for arith in ("f32", "h2"):
        if arith == "f32":
            K, KT, zx = packed(Kh, cuda_device), packed(Kh, cuda_device, transposed=1), dev(Zx, cuda_device)
        else:
            K, KT, zx = packed_h2(Kh, cuda_device), packed_h2(np.ascontiguousarray(Kh.T), cuda_device), dev(h2_zx_pack(Zx, sc), cuda_device)
        dz, dc_in, dh_in = empty((rows, 4 * d), cuda_device), empty((rows, d), cuda_device), empty((rows, d), cuda_device)
        ln_grad = empty((10 * d,), cuda_device, 0.0)
        wsl = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
        task = _lib.LstmBwdTask(None, 0, _lib.ptr(common["h"]), _lib.ptr(common["c"]), _lib.ptr(K), _lib.ptr(common["ln"]),
                                _lib.ptr(common["dh"]), _lib.ptr(common["dc"]), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad),
                                _lib.ptr(wsl), rows, _lib.ptr(common["uv"]), _lib.ptr(zx), _lib.ptr(KT), _lib.ptr(dh_in), 0)
        _lib.call_multi("tspgnn_lnlstm_bwd_multi_" + arith, [task], d)
        torch.cuda.synchronize()
        outs[arith] = [t.cpu().numpy().astype(np.float64) for t in (dz, dc_in, dh_in)]
    for a, b, name in zip(outs["h2"], outs["f32"], ("dz", "dc", "dh")):
        # row by row, relative to the row's own scale
        scale = np.maximum(np.abs(b).max(axis=1, keepdims=True), 1e-300)
        assert (np.abs(a - b) / scale).max() < 2e-5, name


@pytest.mark.parametrize("d,L,mask,rows", [(64, 4, 0b0111, 500), (64, 3, 0b111, 333), (32, 4, 0b0111, 40), (32, 2, 0b01, 17),
                                           (128, 2, 0b11, 100), (64, 1, 0, 64)])
def test_mlp_backward_and_wgrad(cuda_device, d, L, mask, rows):
    rng = np.random.RandomState(d * L + rows)
    X = rng.randn(rows, d).astype(np.float32)
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    bs = [(0.1 * rng.randn(d)).astype(np.float32) for _ in range(L)]
    dY = rng.randn(rows, d).astype(np.float32)
    dX0 = rng.randn(rows, d).astype(np.float32)
    # oracle
    tX = torch.tensor(X, dtype=torch.float64, requires_grad=True)
    tW = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in Ws]
    tb = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    a = tX
    for l in range(L):
        a = a @ tW[l] + tb[l]
        if (mask >> l) & 1:
            a = torch.relu(a)
    grads = torch.autograd.grad((a * torch.tensor(dY, dtype=torch.float64)).sum(), [tX] + tW + tb)
    gX, gW, gb = grads[0], grads[1:1 + L], grads[1 + L:]
    # HIP forward (saves activations) then backward
    wb = torch.cat([torch.cat([packed(w, cuda_device).view(-1), dev(b, cuda_device)]) for w, b in zip(Ws, bs)])
    wt = torch.cat([packed(w, cuda_device, transposed=1).view(-1) for w in Ws])
    _KEEP.extend([wb, wt])
    Xd = dev(X, cuda_device)
    Y = empty((rows, d), cuda_device); acts = empty((max(L - 1, 1), rows, d), cuda_device)
    _lib.call("tspgnn_mlp_fwd_f32", _lib.ptr(Xd), _lib.ptr(wb), _lib.ptr(Y), _lib.ptr(acts), 0, rows, d, L, mask, None)
    dpre = empty((L, rows, d), cuda_device); dX = dev(dX0, cuda_device)
    _lib.call("tspgnn_mlp_bwd_f32", _lib.ptr(dev(dY, cuda_device)), _lib.ptr(wt), _lib.ptr(acts), 0, _lib.ptr(Y), _lib.ptr(dpre), 0,
              _lib.ptr(dX), 1, rows, d, L, mask, None)
    torch.cuda.synchronize()
    assert rel_err(dX.cpu().numpy(), gX.numpy() + dX0) < TOL
    wsz = ws("tspgnn_wgrad_workspace_floats", rows, d, d, device=cuda_device)
    for l in range(L):
        inp = Xd if l == 0 else acts[l - 1]
        dW = empty((d, d), cuda_device, 0.0); db = empty((d,), cuda_device, 0.0)
        _lib.call("tspgnn_wgrad_f32", _lib.ptr(inp), _lib.ptr(dpre[l]), rows, d, d, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(wsz), None)
        torch.cuda.synchronize()
        assert rel_err(dW.cpu().numpy(), gW[l].numpy()) < TOL, l
        assert rel_err(db.cpu().numpy(), gb[l].numpy()) < TOL, l


def test_wgrad_accumulates_and_large_rows(cuda_device):
    rng = np.random.RandomState(5)
    rows, kin, nout = 70001, 64, 256
    X = rng.randn(rows, kin).astype(np.float32); dY = rng.randn(rows, nout).astype(np.float32)
    dW0 = rng.randn(kin, nout).astype(np.float32)
    dW = dev(dW0, cuda_device)
    w = ws("tspgnn_wgrad_workspace_floats", rows, kin, nout, device=cuda_device)
    _lib.call("tspgnn_wgrad_f32", _lib.ptr(dev(X, cuda_device)), _lib.ptr(dev(dY, cuda_device)), rows, kin, nout, _lib.ptr(dW), None,
              _lib.ptr(w), None)
    torch.cuda.synchronize()
    ref = X.astype(np.float64).T @ dY.astype(np.float64) + dW0
    assert rel_err(dW.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("bf16_x", [False, True])
def test_wgrad_wide_outputs_span_several_workgroups(cuda_device, bf16_x):
    """kin x nout = 128 x 512 (the d = 128 cell): sixteen 64 x 64 output blocks per row chunk, i.e. four workgroups per
    chunk, which the kernel keeps on one XCD (blockIdx remapped) -- every block of every chunk must still be formed
    exactly once; with the bias row, fp32 and bf16 X (tspgnn_wgrad_f32 / tspgnn_wgrad_bf16x_f32)."""
    rng = np.random.RandomState(11)
    rows, kin, nout = 70001, 128, 512
    X = rng.randn(rows, kin).astype(np.float32)
    dY = rng.randn(rows, nout).astype(np.float32)
    if bf16_x:
        xt = dev(X, cuda_device).to(torch.bfloat16)
        X = xt.to(torch.float32).cpu().numpy()
    else:
        xt = dev(X, cuda_device)
    dW = empty((kin, nout), cuda_device, 0.0)
    db = empty((nout,), cuda_device, 0.0)
    w = ws("tspgnn_wgrad_workspace_floats", rows, kin, nout, device=cuda_device)
    _lib.call("tspgnn_wgrad_bf16x_f32" if bf16_x else "tspgnn_wgrad_f32", _lib.ptr(xt), _lib.ptr(dev(dY, cuda_device)), rows, kin,
              nout, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(w), None)
    torch.cuda.synchronize()
    assert rel_err(dW.cpu().numpy(), X.astype(np.float64).T @ dY.astype(np.float64)) < TOL
    assert rel_err(db.cpu().numpy(), dY.astype(np.float64).sum(0)) < TOL


@pytest.mark.parametrize("rows,kin,nout", [(5094400, 128, 128), (3194880, 64, 256)])
def test_wgrad_bf16x_two_piece_dY_at_training_row_counts(cuda_device, rows, kin, nout):
    """tspgnn_wgrad_bf16x_f32 takes dY to two bf16 pieces (16 significand bits) for every weight gradient of
    the bf16-storage mode, and those sums run over millions of rows -- 8 steps of a config-5 shard's 636 800 edge rows,
    32 steps of C2's 99 840 -- with heavy cancellation (dY has mean ~0, X >= 0 like a tape of relu activations).  Against
    the float64 product of the SAME operands, norm-wise as everywhere (conftest.rel_err): the bar is the fp32-class 1e-5 --
    the dropped third piece is 2^-17 relative per term and averages out over sqrt(rows); measured 4e-7 / 6e-7."""
    g = torch.Generator(device=cuda_device).manual_seed(3)
    X = torch.relu(torch.randn(rows, kin, device=cuda_device, generator=g)).to(torch.bfloat16)
    dY = torch.randn(rows, nout, device=cuda_device, generator=g) * (1.0 + torch.rand(rows, 1, device=cuda_device, generator=g) * 50.0)
    dW = empty((kin, nout), cuda_device, 0.0)
    db = empty((nout,), cuda_device, 0.0)
    w = ws("tspgnn_wgrad_workspace_floats", rows, kin, nout, device=cuda_device)
    _lib.call("tspgnn_wgrad_bf16x_f32", _lib.ptr(X), _lib.ptr(dY), rows, kin, nout, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(w), None)
    torch.cuda.synchronize()
    ref = torch.zeros(kin, nout, dtype=torch.float64, device=cuda_device)
    refb = torch.zeros(nout, dtype=torch.float64, device=cuda_device)
    for r0 in range(0, rows, 1 << 18):        # float64 reference of the same operands, in slices
        xs, ys = X[r0:r0 + (1 << 18)].to(torch.float64), dY[r0:r0 + (1 << 18)].to(torch.float64)
        ref += xs.T @ ys
        refb += ys.sum(0)
    e_w, e_b = rel_err(dW.cpu().numpy(), ref.cpu().numpy()), rel_err(db.cpu().numpy(), refb.cpu().numpy())
    print("wgrad_bf16x rows=%d %dx%d: dW %.2e db %.2e" % (rows, kin, nout, e_w, e_b))
    assert e_w < TOL and e_b < TOL


def test_vote_head_backward_pieces(cuda_device):
    rng = np.random.RandomState(9)
    n_edges = np.array([3, 780, 21, 190]); B = len(n_edges)
    seg = np.concatenate([[0], np.cumsum(n_edges)]).astype(np.int32); M = int(seg[-1]); d = 64
    logits = rng.randn(B).astype(np.float32); labels = np.array([0, 1, 1, 0], dtype=np.float32)
    dvote = empty((M,), cuda_device)
    _lib.call("tspgnn_vote_grad_f32", _lib.ptr(dev(logits, cuda_device)), _lib.ptr(dev(labels, cuda_device)),
              _lib.ptr(dev(seg, cuda_device, np.int32)), _lib.ptr(dvote), B, None)
    torch.cuda.synchronize()
    sig = 1 / (1 + np.exp(-logits.astype(np.float64)))
    ref = np.concatenate([np.full(n, (sig[p] - labels[p]) / (B * n)) for p, n in enumerate(n_edges)])
    assert rel_err(dvote.cpu().numpy(), ref) < 1e-6
    X = rng.randn(M, d).astype(np.float32); w = rng.randn(d).astype(np.float32)
    dX = empty((M, d), cuda_device)
    _lib.call("tspgnn_rowdot_bwd_f32", _lib.ptr(dvote), _lib.ptr(dev(w, cuda_device)), _lib.ptr(dX), M, d, None)
    dw = empty((d,), cuda_device, 0.0); dbias = empty((1,), cuda_device, 0.0)
    wsz = ws("tspgnn_wcolsum_workspace_floats", M, d, device=cuda_device)
    _lib.call("tspgnn_wcolsum_f32", _lib.ptr(dev(X, cuda_device)), _lib.ptr(dvote), M, d, 1.0, _lib.ptr(dw), _lib.ptr(dbias),
              _lib.ptr(wsz), None)
    col = empty((d,), cuda_device, 1.0)
    _lib.call("tspgnn_wcolsum_f32", _lib.ptr(dev(X, cuda_device)), None, M, d, 0.125, _lib.ptr(col), None, _lib.ptr(wsz), None)
    torch.cuda.synchronize()
    assert rel_err(dX.cpu().numpy(), ref[:, None] * w[None].astype(np.float64)) < 1e-6
    assert rel_err(dw.cpu().numpy(), (ref[:, None] * X).sum(0)) < TOL
    assert abs(dbias.item() - ref.sum()) < 1e-6
    assert rel_err(col.cpu().numpy(), 1.0 + 0.125 * X.astype(np.float64).sum(0)) < TOL


@pytest.mark.parametrize("d", [32, 64, 128])
def test_einit_backward(cuda_device, d):
    rng = np.random.RandomState(d)
    M = 333
    WC = rng.rand(M, 2).astype(np.float32)
    dims = [2, d // 8, d // 4, d // 2, d]
    Ws = [rng.randn(a, b).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])]
    bs = [(0.3 * rng.randn(b)).astype(np.float32) for b in dims[1:]]
    dE0 = rng.randn(M, d).astype(np.float32)
    tW = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in Ws]
    tb = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    a = torch.tensor(WC, dtype=torch.float64)
    for l in range(4):
        a = a @ tW[l] + tb[l]
        if l < 3:
            a = torch.relu(a)
    grads = torch.autograd.grad((a * torch.tensor(dE0, dtype=torch.float64)).sum(), [v for pair in zip(tW, tb) for v in pair])
    ref = np.concatenate([g.numpy().reshape(-1) for g in grads])
    wb = np.concatenate([v.reshape(-1) for pair in zip(Ws, bs) for v in pair])
    dwb = empty((wb.size,), cuda_device, 0.0)
    wsz = ws("tspgnn_einit_bwd_workspace_floats", M, d, device=cuda_device)
    _lib.call("tspgnn_einit_bwd_f32", _lib.ptr(dev(WC, cuda_device)), _lib.ptr(dev(wb, cuda_device)), _lib.ptr(dev(dE0, cuda_device)),
              _lib.ptr(dwb), _lib.ptr(wsz), M, d, None)
    torch.cuda.synchronize()
    assert rel_err(dwb.cpu().numpy(), ref) < TOL


def test_adam_clip_step(cuda_device):
    rng = np.random.RandomState(1)
    n = 115529
    theta = rng.randn(n).astype(np.float32); g = (0.01 * rng.randn(n)).astype(np.float32)
    m = (0.001 * rng.randn(n)).astype(np.float32); v = (1e-4 * rng.rand(n)).astype(np.float32)
    step = 7
    lr_t = TO.LEARNING_RATE * np.sqrt(1 - TO.ADAM_B2 ** step) / (1 - TO.ADAM_B1 ** step)
    td, gd, md, vd = (dev(a, cuda_device) for a in (theta, g, m, v))
    gn = empty((1,), cuda_device); wsz = ws("tspgnn_adam_workspace_floats", device=cuda_device)
    _lib.call("tspgnn_adam_clip_step_f32", _lib.ptr(td), _lib.ptr(gd), _lib.ptr(md), _lib.ptr(vd), n, TO.L2NORM_SCALING, TO.CLIP_NORM,
              float(lr_t), TO.ADAM_B1, TO.ADAM_B2, TO.ADAM_EPS, _lib.ptr(gn), _lib.ptr(wsz), None, None, None)
    torch.cuda.synchronize()
    g64 = g.astype(np.float64) + TO.L2NORM_SCALING * theta
    clipped, gnorm = TO.clip_by_global_norm({"a": g64})
    p, m2, v2 = TO.adam_step({"a": theta.astype(np.float64)}, clipped, {"a": m.astype(np.float64)}, {"a": v.astype(np.float64)}, step)
    assert abs(gn.item() - gnorm) < 1e-5 * gnorm and gnorm > TO.CLIP_NORM     # the clip is active in this case
    assert rel_err(md.cpu().numpy(), m2["a"]) < 1e-6 and rel_err(vd.cpu().numpy(), v2["a"]) < 1e-6
    assert np.abs(td.cpu().numpy() - p["a"]).max() < 1e-6 * np.abs(p["a"]).max()
    # device-side step counter: same update when the counter reaches the same step
    td2, gd2, md2, vd2 = (dev(a, cuda_device) for a in (theta, g, m, v))
    cnt = torch.full((1,), step - 1, dtype=torch.int32, device=cuda_device); _KEEP.append(cnt)
    _lib.call("tspgnn_adam_clip_step_f32", _lib.ptr(td2), _lib.ptr(gd2), _lib.ptr(md2), _lib.ptr(vd2), n, TO.L2NORM_SCALING, TO.CLIP_NORM,
              TO.LEARNING_RATE, TO.ADAM_B1, TO.ADAM_B2, TO.ADAM_EPS, _lib.ptr(gn), _lib.ptr(wsz), _lib.ptr(cnt), None, None)
    torch.cuda.synchronize()
    assert int(cnt.item()) == step
    assert np.abs(td2.cpu().numpy() - p["a"]).max() < 2e-6 * np.abs(p["a"]).max()
    # The UPDATE itself, on exact inputs: with |theta| ~ 1e-3 one ulp of theta (1e-10) is far below the Adam step
    # (<= lr = 2e-5), so theta' - theta exposes the kernel's arithmetic: every element's movement within 1e-5 of the
    # largest movement (the model-level training tests cannot be this tight: there the gradients differ in rounding).
    small = (1e-3 * theta).astype(np.float32)
    td3, gd3, md3, vd3 = (dev(a, cuda_device) for a in (small, g, m, v))
    _lib.call("tspgnn_adam_clip_step_f32", _lib.ptr(td3), _lib.ptr(gd3), _lib.ptr(md3), _lib.ptr(vd3), n, TO.L2NORM_SCALING, TO.CLIP_NORM,
              float(lr_t), TO.ADAM_B1, TO.ADAM_B2, TO.ADAM_EPS, _lib.ptr(gn), _lib.ptr(wsz), None, None, None)
    torch.cuda.synchronize()
    g64 = g.astype(np.float64) + TO.L2NORM_SCALING * small
    clipped, _ = TO.clip_by_global_norm({"a": g64})
    p3, _, _ = TO.adam_step({"a": small.astype(np.float64)}, clipped, {"a": m.astype(np.float64)}, {"a": v.astype(np.float64)}, step)
    moved_ref = p3["a"] - small.astype(np.float64)
    moved = td3.cpu().numpy().astype(np.float64) - small.astype(np.float64)
    assert np.abs(moved - moved_ref).max() < 1e-5 * np.abs(moved_ref).max()


@pytest.mark.parametrize("d,rows,n_src", [(64, 700, 41), (32, 33, 9)])
def test_mlp_backward_gather_init_equals_explicit_gather(cuda_device, d, rows, n_src):
    """tspgnn_mlp_bwd_task.uv: the chain starts from dY[u] + dY[v] -- bit-identical to gather2_sum followed by the
    plain chain (same additions in the same order)."""
    rng = np.random.RandomState(rows)
    L, mask = 3, 0b011
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    wt = torch.cat([packed(w, cuda_device, transposed=1).view(-1) for w in Ws])
    uv = np.stack([rng.randint(0, n_src, rows), rng.randint(0, n_src, rows)], 1).astype(np.int32)
    src = dev(rng.randn(n_src, d).astype(np.float32), cuda_device)
    acts = dev(rng.randn(L - 1, rows, d).astype(np.float32), cuda_device)
    uvd = dev(uv, cuda_device, np.int32)
    gathered = empty((rows, d), cuda_device)
    _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(uvd), _lib.ptr(src), _lib.ptr(gathered), rows, n_src, d, None)
    outs = []
    for dY, uvp in ((gathered, None), (src, uvd)):
        dpre = empty((L, rows, d), cuda_device)
        dX = empty((rows, d), cuda_device)
        task = _lib.MlpBwdTask(_lib.ptr(dY), _lib.ptr(wt), _lib.ptr(acts), 0, None, _lib.ptr(dpre), 0, _lib.ptr(dX), 0, rows, L, mask,
                               _lib.ptr(uvp))
        _lib.call_multi("tspgnn_mlp_bwd_multi_f32", [task], d)
        torch.cuda.synchronize()
        outs.append((dpre.cpu().numpy(), dX.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_adam_skip_flag_leaves_the_variables_alone(cuda_device):
    """skip_flag != 0 (the f16x2 range flag of the step's forward, include/tspgnn.h): theta, m, v and the device step
    counter stay bit for bit what they were, the global norm is still reported; flag == 0: the ordinary step."""
    rng = np.random.RandomState(9)
    n = 5000
    theta = rng.randn(n).astype(np.float32); g = (0.01 * rng.randn(n)).astype(np.float32)
    m = (0.001 * rng.randn(n)).astype(np.float32); v = (1e-4 * rng.rand(n)).astype(np.float32)
    wsz = ws("tspgnn_adam_workspace_floats", device=cuda_device)
    res = {}
    for flagged in (1, 0):
        td, gd, md, vd = (dev(a, cuda_device) for a in (theta, g, m, v))
        gn = empty((1,), cuda_device)
        cnt = torch.full((1,), 4, dtype=torch.int32, device=cuda_device); _KEEP.append(cnt)
        flag = torch.full((1,), flagged, dtype=torch.int32, device=cuda_device); _KEEP.append(flag)
        _lib.call("tspgnn_adam_clip_step_f32", _lib.ptr(td), _lib.ptr(gd), _lib.ptr(md), _lib.ptr(vd), n, TO.L2NORM_SCALING,
                  TO.CLIP_NORM, TO.LEARNING_RATE, TO.ADAM_B1, TO.ADAM_B2, TO.ADAM_EPS, _lib.ptr(gn), _lib.ptr(wsz), _lib.ptr(cnt),
                  _lib.ptr(flag), None)
        torch.cuda.synchronize()
        res[flagged] = (td.cpu().numpy(), md.cpu().numpy(), vd.cpu().numpy(), int(cnt.item()), float(gn.item()))
    assert np.array_equal(res[1][0], theta) and np.array_equal(res[1][1], m) and np.array_equal(res[1][2], v) and res[1][3] == 4
    assert res[0][3] == 5 and not np.array_equal(res[0][0], theta)
    assert res[1][4] == res[0][4] > 0


def _h2_blocks(Ws, bs, device, transposed=False):
    """The f16x2 kernels' weight blocks: forward {pack_h2(W), 2^s b} per layer, or pack_h2(W^T) back to back."""
    d = Ws[0].shape[0]
    scale = _lib.lib.tspgnn_h2_weight_scale()
    per = 4 * d * d + (0 if transposed else 4 * d)
    out = torch.empty(len(Ws) * per, dtype=torch.uint8, device=device)
    for j, (w, b) in enumerate(zip(Ws, bs)):
        src = dev(w.T if transposed else w, device)
        _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src), _lib.ptr(out[j * per:j * per + 4 * d * d]), d, d, None, None)
        if not transposed:
            out[j * per + 4 * d * d:(j + 1) * per].copy_(dev(scale * b, device).view(torch.uint8))
    _KEEP.append(out)
    return out


@pytest.mark.parametrize("L,mask", [(3, 0b111), (2, 0b11), (1, 0b1), (3, 0b011), (1, 0)])
@pytest.mark.parametrize("rows,n_src", [(1, 0), (33, 7), (700, 41), (1000, 0), (40000, 333)])
def test_mlp_backward_recompute(cuda_device, L, mask, rows, n_src):
    """tspgnn_mlp_bwd_rc_h2 (hidden activations recomputed, data gradient on the fp16 matrix cores, weight gradients formed in
    the launch) against float64 arithmetic on the chain with the forward's own relu masks: dX (accumulated), and dW / db
    from two launches into one partial buffer, folded into a gradient slice that already holds values."""
    d = 64
    rng = np.random.RandomState(100 * L + rows + mask)
    X = rng.randn(rows, d).astype(np.float32)
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    bs = [(0.1 * rng.randn(d)).astype(np.float32) for _ in range(L)]
    if n_src:
        uv = np.stack([rng.randint(0, n_src, rows), rng.randint(0, n_src, rows)], 1).astype(np.int32)
        # gradients of very different magnitude from row to row, as in a real backward pass
        src = (rng.randn(n_src, d) * 10.0 ** rng.uniform(-7, -2, (n_src, 1))).astype(np.float32)
        dY = src[uv[:, 0]] + src[uv[:, 1]]
    else:
        uv, src = None, None
        dY = (rng.randn(rows, d) * 10.0 ** rng.uniform(-7, -2, (rows, 1))).astype(np.float32)
    dX0 = (0.5 * rng.randn(rows, d) * np.abs(dY).max(1, keepdims=True)).astype(np.float32)   # (of each row's own scale)
    # the f16x2 forward: messages and hidden activations
    wb = _h2_blocks(Ws, bs, cuda_device)
    wt = _h2_blocks(Ws, bs, cuda_device, transposed=True)
    Xd = dev(X, cuda_device)
    Y = empty((rows, d), cuda_device)
    acts = empty((max(L - 1, 1), rows, d), cuda_device)
    task = _lib.MlpTask(_lib.ptr(Xd), _lib.ptr(wb), _lib.ptr(Y), _lib.ptr(acts) if L > 1 else None, rows * d, rows, L, mask, None,
                        None, None)
    _lib.call_multi("tspgnn_mlp_fwd_multi_h2", [task], d)
    torch.cuda.synchronize()
    A = [X.astype(np.float64)] + [acts[l].cpu().numpy().astype(np.float64) for l in range(L - 1)] + [Y.cpu().numpy().astype(np.float64)]
    G = dY.astype(np.float64)
    want_dpre = [None] * L
    for l in range(L - 1, -1, -1):
        if (mask >> l) & 1:
            G = G * (A[l + 1] > 0)
        want_dpre[l] = G
        G = G @ Ws[l].astype(np.float64).T
    # HIP: two launches into one partial buffer
    dYd = dev(src if n_src else dY, cuda_device)
    uvd = dev(uv, cuda_device, np.int32) if n_src else None
    n_part = int(_lib.lib.tspgnn_mlp_bwd_rc_partial_floats(d, L))
    part = empty((n_part,), cuda_device, 0.0)
    outs, dXs = [], []
    for rep in range(2):
        dX = dev(dX0, cuda_device)
        t = _lib.MlpBwdRcTask(_lib.ptr(Xd), _lib.ptr(wb), _lib.ptr(wt), _lib.ptr(Y), _lib.ptr(dYd), _lib.ptr(uvd), _lib.ptr(dX), 1,
                              rows, L, mask, None, 0, None, 0, _lib.ptr(part))
        _lib.call("tspgnn_mlp_bwd_rc_h2", ctypes.cast(ctypes.pointer(t), ctypes.c_void_p), d, None)
        torch.cuda.synchronize()
        outs.append(part.clone())
        dXs.append(dX)
    assert torch.equal(dXs[0], dXs[1])
    assert torch.equal(outs[1], 2 * outs[0])      # (deterministic: the second launch adds exactly what the first one did)
    got = dXs[0].cpu().numpy().astype(np.float64) - dX0
    # per row: the rows' scales differ by five decades, and each must come out to fp32-class accuracy
    scale = np.maximum(np.abs(G).max(1, keepdims=True), 1e-30)
    assert (np.abs(got - G) / scale).max() < 4 * TOL
    # the form without in-launch weight gradients was removed (round 6): the entry point says so
    t = _lib.MlpBwdRcTask(_lib.ptr(Xd), _lib.ptr(wb), _lib.ptr(wt), _lib.ptr(Y), _lib.ptr(dYd), _lib.ptr(uvd), _lib.ptr(dXs[0]), 1,
                          rows, L, mask, None, 0, None, 0, None)
    with pytest.raises(_lib.TspgnnError):
        _lib.call("tspgnn_mlp_bwd_rc_h2", ctypes.cast(ctypes.pointer(t), ctypes.c_void_p), d, None)
    g0 = (1e-3 * rng.randn(L * (d * d + d))).astype(np.float32) * np.float32(np.abs(want_dpre[0]).max())
    grad = dev(g0, cuda_device)
    _lib.call("tspgnn_mlp_bwd_rc_finish_f32", _lib.ptr(part), _lib.ptr(grad), d, L, None)
    torch.cuda.synchronize()
    got = grad.cpu().numpy().astype(np.float64) - g0
    for l in range(L):
        o = l * (d * d + d)
        assert rel_err(got[o:o + d * d].reshape(d, d), 2 * (A[l].T @ want_dpre[l])) < TOL, ("dW", l)
        assert rel_err(got[o + d * d:o + d * d + d], 2 * want_dpre[l].sum(0)) < TOL, ("db", l)


@pytest.mark.parametrize("d,bf16", [(64, False), (128, True), (128, False), (64, True)])
def test_mlp_backward_taped_h2(cuda_device, d, bf16):
    """tspgnn_mlp_bwd_multi_h2: the taped backward with its data gradient on the fp16 matrix cores -- two MLPs of different
    depth in one launch (one of them in gather-init mode), fp32 and bf16 tapes, widths 64 and 128, against float64
    arithmetic with the tape's own masks; the pre-activation gradients it hands on and dX, row by row (the rows' scales
    differ by five decades)."""
    rng = np.random.RandomState(77 + d)
    if d == 64:
        specs = [dict(L=3, mask=0b111, rows=5000, n_src=120, acc=1), dict(L=4, mask=0b0111, rows=333, n_src=0, acc=0)]
    else:
        specs = [dict(L=2, mask=0b11, rows=3000, n_src=90, acc=1), dict(L=2, mask=0b01, rows=77, n_src=0, acc=0)]
    tasks, checks = [], []
    for sp in specs:
        L, mask, rows, n_src = sp["L"], sp["mask"], sp["rows"], sp["n_src"]
        Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
        acts = rng.randn(max(L - 1, 1), rows, d).astype(np.float32)
        Y = rng.randn(rows, d).astype(np.float32)
        if bf16:
            acts = torch.from_numpy(acts).to(torch.bfloat16)
            Y = torch.from_numpy(Y).to(torch.bfloat16)
            acts_np, Y_np = acts.to(torch.float32).numpy(), Y.to(torch.float32).numpy()
            acts_d, Y_d = acts.to(cuda_device), Y.to(cuda_device)
            _KEEP.extend([acts_d, Y_d])
        else:
            acts_np, Y_np = acts, Y
            acts_d, Y_d = dev(acts, cuda_device), dev(Y, cuda_device)
        if n_src:
            uv = np.stack([rng.randint(0, n_src, rows), rng.randint(0, n_src, rows)], 1).astype(np.int32)
            src = (rng.randn(n_src, d) * 10.0 ** rng.uniform(-7, -2, (n_src, 1))).astype(np.float32)
            dY = src[uv[:, 0]] + src[uv[:, 1]]
        else:
            uv, src = None, None
            dY = (rng.randn(rows, d) * 10.0 ** rng.uniform(-7, -2, (rows, 1))).astype(np.float32)
        dX0 = (0.5 * rng.randn(rows, d) * np.abs(dY).max(1, keepdims=True)).astype(np.float32)
        G = dY.astype(np.float64)
        want = [None] * L
        for l in range(L - 1, -1, -1):
            if (mask >> l) & 1:
                G = G * ((Y_np if l == L - 1 else acts_np[l]) > 0)
            want[l] = G
            G = G @ Ws[l].astype(np.float64).T
        wt = _h2_blocks(Ws, [None] * L, cuda_device, transposed=True)
        dpre = empty((L, rows, d), cuda_device, 7.0)
        dX = dev(dX0, cuda_device)
        tasks.append(_lib.MlpBwdTask(_lib.ptr(dev(src if n_src else dY, cuda_device)), _lib.ptr(wt), _lib.ptr(acts_d),
                                     rows * d, _lib.ptr(Y_d), _lib.ptr(dpre), rows * d, _lib.ptr(dX), sp["acc"], rows,
                                     L, mask, _lib.ptr(dev(uv, cuda_device, np.int32)) if n_src else None, 1 if bf16 else 0))
        checks.append((dX, dX0 * sp["acc"], G, dpre, want))
    _lib.call_multi("tspgnn_mlp_bwd_multi_h2", tasks, d)
    torch.cuda.synchronize()
    for dX, base, G, dpre, want in checks:
        got = dX.cpu().numpy().astype(np.float64) - base
        scale = np.maximum(np.abs(G).max(1, keepdims=True), 1e-30)
        assert (np.abs(got - G) / scale).max() < 4 * TOL
        for l, ref in enumerate(want):
            scale = np.maximum(np.abs(ref).max(1, keepdims=True), 1e-30)
            assert (np.abs(dpre[l].cpu().numpy() - ref) / scale).max() < TOL, ("dpre", l)


def test_mlp_backward_h2_rows_of_subnormal_and_zero_gradients(cuda_device):
    """A gradient row whose largest entry is an fp32 subnormal (or zero) must come out as (about) zero, not as NaN: the row
    is normalised by a power of two before its fp16 split, and 2^-e has to stay finite (h2_row_exponent)."""
    d, rows, L, mask = 64, 48, 2, 0b01
    rng = np.random.RandomState(3)
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    acts = rng.randn(1, rows, d).astype(np.float32)
    Y = rng.randn(rows, d).astype(np.float32)
    dY = rng.randn(rows, d).astype(np.float32)
    dY[5] = 0.0
    dY[6] = 1e-42          # fp32 subnormals
    dY[7] = -3e-39
    wt = _h2_blocks(Ws, [None] * L, cuda_device, transposed=True)
    dpre = empty((L, rows, d), cuda_device, 7.0)
    dX = empty((rows, d), cuda_device, 7.0)
    task = _lib.MlpBwdTask(_lib.ptr(dev(dY, cuda_device)), _lib.ptr(wt), _lib.ptr(dev(acts, cuda_device)), rows * d,
                           _lib.ptr(dev(Y, cuda_device)), _lib.ptr(dpre), rows * d, _lib.ptr(dX), 0, rows, L, mask, None, 0)
    _lib.call_multi("tspgnn_mlp_bwd_multi_h2", [task], d)
    torch.cuda.synchronize()
    out = dX.cpu().numpy()
    assert np.isfinite(out).all() and np.isfinite(dpre.cpu().numpy()).all()
    assert np.abs(out[5:8]).max() < 1e-30
    G = dY.astype(np.float64) @ Ws[1].astype(np.float64).T
    G = (G * (acts[0] > 0)) @ Ws[0].astype(np.float64).T
    assert rel_err(out[8:], G[8:]) < 4 * TOL


@pytest.mark.parametrize("rows", [1, 333, 5120])
def test_lnlstm_backward_h2_streamed_data_gradient(cuda_device, rows):
    """tspgnn_lstm_bwd_task.KTg: [dxg | dxh] = dz K^T formed in the launch (second phase of the task's workgroups, K^T in
    K's place) -- beside a second task without it in the same launch, bias-init mode as the pushed vertex cell runs it:
    dz, dc bit-identical to the launch without KTg, the data gradient against float64 arithmetic on the dz the kernel
    wrote, row by row (incoming gradients spread over twelve orders of magnitude)."""
    d = dx = 64
    rng = np.random.RandomState(rows)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    x, h, c = f32(rng.randn(rows, dx)), f32(rng.randn(rows, d)), f32(rng.randn(rows, d))
    K = f32(rng.randn(dx + d, 4 * d) / np.sqrt(dx + d))
    ln = f32(np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]))
    zb, zs = f32(0.1 * rng.randn(1, 4 * d)), f32(rng.randint(1, 40, rows))
    row_scale = 10.0 ** rng.uniform(-9, 3, size=(rows, 1))
    dh_o, dc_o = f32(row_scale * rng.randn(rows, d)), f32(row_scale * rng.randn(rows, d))
    xd, hd, cd, lnd = dev(x, cuda_device), dev(h, cuda_device), dev(c, cuda_device), dev(ln, cuda_device)
    dhd, dcd, zbd, zsd = dev(dh_o, cuda_device), dev(dc_o, cuda_device), dev(zb, cuda_device), dev(zs, cuda_device)
    Kp, KTp = packed_h2(K, cuda_device), packed_h2(np.ascontiguousarray(K.T), cuda_device)
    outs = {}
    for fused in (False, True):
        dz, dc_in = empty((rows, 4 * d), cuda_device), empty((rows, d), cuda_device)
        dxo, dho = empty((rows, dx), cuda_device, 7.0), empty((rows, d), cuda_device, 7.0)
        ln_grad = empty((10 * d,), cuda_device, 0.0)
        wsl = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
        task = _lib.LstmBwdTask(_lib.ptr(xd), dx, _lib.ptr(hd), _lib.ptr(cd), _lib.ptr(Kp), _lib.ptr(lnd), _lib.ptr(dhd),
                                _lib.ptr(dcd), _lib.ptr(dz), _lib.ptr(dc_in), _lib.ptr(ln_grad), _lib.ptr(wsl), rows, None, None,
                                None, _lib.ptr(dho) if fused else None, 0, _lib.ptr(zbd), _lib.ptr(zsd),
                                _lib.ptr(KTp) if fused else None, _lib.ptr(dxo) if fused else None)
        # a second, plain task beside it (the launch sizes LDS and splits workgroups over both)
        r2 = min(77, rows)
        dz2, dc2, lg2 = empty((r2, 4 * d), cuda_device), empty((r2, d), cuda_device), empty((10 * d,), cuda_device, 0.0)
        ws2 = ws("tspgnn_lnlstm_bwd_workspace_floats", d, device=cuda_device)
        other = _lib.LstmBwdTask(_lib.ptr(xd), dx, _lib.ptr(hd), _lib.ptr(cd), _lib.ptr(Kp), _lib.ptr(lnd), None, _lib.ptr(dcd),
                                 _lib.ptr(dz2), _lib.ptr(dc2), _lib.ptr(lg2), _lib.ptr(ws2), r2, None, None, None, None, 0)
        _lib.call_multi("tspgnn_lnlstm_bwd_multi_h2", [other, task], d)
        torch.cuda.synchronize()
        outs[fused] = [t.cpu().numpy() for t in (dz, dc_in, ln_grad, dxo, dho, dz2)]
    for a, b in zip(outs[True][:3] + outs[True][5:], outs[False][:3] + outs[False][5:]):
        assert np.array_equal(a, b)
    want = outs[True][0].astype(np.float64) @ K.astype(np.float64).T
    got = np.concatenate([outs[True][3], outs[True][4]], axis=1).astype(np.float64)
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-300)
    assert (np.abs(got - want) / scale).max() < 2 * TOL


@pytest.mark.parametrize("rows,k", [(1, 256), (333, 256), (5120, 256), (100, 64)])
def test_mlp_backward_h2_projected_head(cuda_device, rows, k):
    """tspgnn_mlp_bwd_task.pre_X: the chain starts from dY = pre_X P^T formed inside the launch (P^T resident in LDS behind
    the layers' weights) -- beside a plain task in the same launch; dX and the pre-activation gradients against float64
    arithmetic from pre_X, row by row, with row scales spread over nine decades."""
    d, L, mask = 64, 4, 0b0111
    rng = np.random.RandomState(rows + k)
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    P = (rng.randn(d, k) / np.sqrt(d)).astype(np.float32)           # y P = projected message: dY = dZx P^T
    acts = rng.randn(L - 1, rows, d).astype(np.float32)
    X = (rng.randn(rows, k) * 10.0 ** rng.uniform(-8, 1, (rows, 1))).astype(np.float32)
    G = X.astype(np.float64) @ P.astype(np.float64).T
    want = [None] * L
    for l in range(L - 1, -1, -1):
        if (mask >> l) & 1:
            G = G * (acts[l] > 0)
        want[l] = G
        G = G @ Ws[l].astype(np.float64).T
    wt = _h2_blocks(Ws, [None] * L, cuda_device, transposed=True)
    pw = packed_h2(np.ascontiguousarray(P.T), cuda_device)          # pack_weights_h2(P^T [k, d])
    dpre, dX = empty((L, rows, d), cuda_device, 7.0), empty((rows, d), cuda_device, 7.0)
    task = _lib.MlpBwdTask(None, _lib.ptr(wt), _lib.ptr(dev(acts, cuda_device)), rows * d, None, _lib.ptr(dpre), rows * d,
                           _lib.ptr(dX), 0, rows, L, mask, None, 0, _lib.ptr(dev(X, cuda_device)), _lib.ptr(pw), k)
    r2 = 500
    dY2 = rng.randn(r2, d).astype(np.float32)
    dX2 = empty((r2, d), cuda_device, 7.0)
    plain = _lib.MlpBwdTask(_lib.ptr(dev(dY2, cuda_device)), _lib.ptr(wt), None, 0, None, None, 0, _lib.ptr(dX2), 0, r2, 1, 0,
                            None, 0)
    _lib.call_multi("tspgnn_mlp_bwd_multi_h2", [plain, task], d)
    torch.cuda.synchronize()
    scale = np.maximum(np.abs(G).max(1, keepdims=True), 1e-300)
    assert (np.abs(dX.cpu().numpy() - G) / scale).max() < 4 * TOL
    for l, ref in enumerate(want):
        scale = np.maximum(np.abs(ref).max(1, keepdims=True), 1e-300)
        assert (np.abs(dpre[l].cpu().numpy() - ref) / scale).max() < 2 * TOL, ("dpre", l)
    ref2 = dY2.astype(np.float64) @ Ws[0].astype(np.float64).T
    assert rel_err(dX2.cpu().numpy(), ref2) < TOL
    # the f32 entry point refuses the field
    with pytest.raises(_lib.TspgnnError):
        _lib.call_multi("tspgnn_mlp_bwd_multi_f32", [task], d)


@pytest.mark.parametrize("d,L", [(64, 4), (32, 3), (128, 2), (64, 1)])
def test_pack_mlp_h2_equals_the_per_layer_packings(cuda_device, d, L):
    """tspgnn_pack_mlp_h2: every layer of an MLP in one launch from the flat {W, b} blocks -- byte for byte what
    tspgnn_pack_weights_h2 per layer plus the scaled bias give (forward form), and what it gives on W^T (backward form);
    the range-guard word ends at the same maximum."""
    rng = np.random.RandomState(d + L)
    Ws = [(rng.randn(d, d) / np.sqrt(d)).astype(np.float32) for _ in range(L)]
    bs = [rng.randn(d).astype(np.float32) for _ in range(L)]
    flat = dev(np.concatenate([np.concatenate([w.reshape(-1), b]) for w, b in zip(Ws, bs)]), cuda_device)
    for transposed in (0, 1):
        per = 4 * d * d + (0 if transposed else 4 * d)
        out = torch.zeros(L * per, dtype=torch.uint8, device=cuda_device)
        guard = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        _lib.call("tspgnn_pack_mlp_h2", _lib.ptr(flat), _lib.ptr(out), d, L, transposed, _lib.ptr(guard), None)
        ref = _h2_blocks(Ws, bs, cuda_device, transposed=bool(transposed))
        g2 = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        for w in Ws:
            tmp = torch.empty(4 * d * d, dtype=torch.uint8, device=cuda_device)
            _lib.call("tspgnn_pack_weights_h2", _lib.ptr(dev(w, cuda_device)), _lib.ptr(tmp), d, d, _lib.ptr(g2), None)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        assert int(guard.item()) == int(g2.item()) != 0
