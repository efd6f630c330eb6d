"""The split-operand (fp32-accurate) dense kernels vs the float64 numpy oracle, through the C ABI, for both
arithmetics: "h2" = two fp16 pieces per operand on the fp16 matrix cores (csrc/dense_h2.hip, the default of the
inference forward) and "x3" = three bf16 pieces on the bf16 matrix cores (csrc/dense_x3.hip).

The tolerance is the same as for the fp32 kernels: both splits keep every term down to 2^-24 relative.  The f16x2
conventions the tests honour: weights packed as 2^s W, Dense biases stored as 2^s b, projected messages (proj_out /
Zx) carrying the factor 2^s."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import h2_zx_pack, h2_zx_unpack, rel_err
from oracle import np_oracle as NO
from tspgnn import _lib

pytestmark = pytest.mark.gpu

_KEEP = []


def dev(a, device, dtype=np.float32):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_uploads():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


ARITHS = ["h2", "x3"]
PIECES = {"x3": 3, "h2": 2}


def zx_in(arith, Zx):
    """Projected messages as the cell of this arithmetic reads them."""
    return h2_zx_pack(Zx, zscale("h2")) if arith == "h2" else Zx


def zx_buffer(arith, n, width, device):
    rows = (n + 15) // 16 * 16 if arith == "h2" else n
    return torch.empty((rows, width), dtype=torch.float32, device=device)


def zx_out(arith, t, n):
    """What a projection of this arithmetic wrote, as the row-major unscaled [n, 4d] matrix."""
    return h2_zx_unpack(t.cpu().numpy(), n, zscale("h2")) if arith == "h2" else t.cpu().numpy()


def zscale(arith):
    """Factor carried by the weights / the z of a cell / the projected messages of this arithmetic."""
    return float(_lib.lib.tspgnn_h2_weight_scale()) if arith == "h2" else 1.0


def packed(arith, W, device):
    """tspgnn_pack_weights_{x3,h2} -> uint8 tensor of pieces*krows*ncols*2 bytes."""
    src = dev(W, device)
    out = torch.empty(PIECES[arith] * W.size * 2, dtype=torch.uint8, device=device)
    _KEEP.append(out)
    if arith == "h2":
        _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src), _lib.ptr(out), W.shape[0], W.shape[1], None, None)
    else:
        _lib.call("tspgnn_pack_weights_" + arith, _lib.ptr(src), _lib.ptr(out), W.shape[0], W.shape[1], None)
    return out


def packed_x3(W, device):
    return packed("x3", W, device)


def bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def test_pack_x3_pieces_sum_to_the_weight(cuda_device):
    for kr, nc in ((64, 64), (32, 32), (64, 256), (128, 256)):
        rng = np.random.RandomState(kr + nc)
        W = (rng.randn(kr, nc) * np.exp(rng.randn(kr, nc))).astype(np.float32)
        P = packed_x3(W, cuda_device).cpu().numpy().view(np.uint16).reshape(3, kr // 32, 4, nc // 16, 16, 8)
        pieces = bf16_to_f64(P)
        back = np.zeros((kr, nc))
        for kb in range(kr // 32):
            for g in range(4):
                for j in range(8):
                    k = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3)
                    back[k] = pieces[:, kb, g, :, :, j].sum(0).reshape(-1)   # [3, NT, 16] -> columns t*16+jl
        # three bf16 pieces carry 24 mantissa bits: the sum is the fp32 value up to its last bit
        assert np.max(np.abs(back - W) / np.abs(W)) < 2.0 ** -22
        assert np.array_equal(pieces[0, 0, 0, 0, :, 0].astype(np.float32),
                              torch.from_numpy(W[0, :16].copy()).to(torch.bfloat16).to(torch.float32).numpy())


def test_pack_h2_pieces_sum_to_the_scaled_weight(cuda_device):
    sc = zscale("h2")
    for kr, nc in ((64, 64), (32, 32), (64, 256), (128, 256)):
        rng = np.random.RandomState(kr + nc)
        W = (0.2 * rng.randn(kr, nc) * np.exp(0.5 * rng.randn(kr, nc))).astype(np.float32)
        P = packed("h2", W, cuda_device).cpu().numpy().view(np.float16).reshape(2, kr // 32, 4, nc // 16, 16, 8)
        pieces = P.astype(np.float64)
        back = np.zeros((kr, nc))
        for kb in range(kr // 32):
            for g in range(4):
                for j in range(8):
                    k = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3)
                    back[k] = pieces[:, kb, g, :, :, j].sum(0).reshape(-1)
        # two fp16 pieces: 2^-24 relative while the lo piece is a normal fp16 (|2^s w| >= 2^-2), 2^-25 absolute below
        err = np.abs(back - sc * W.astype(np.float64))
        assert np.all(err <= np.maximum(2.0 ** -23 * np.abs(sc * W), 2.0 ** -25))
        assert np.array_equal(pieces[0, 0, 0, 0, :, 0], (sc * W[0, :16]).astype(np.float16).astype(np.float64))


def mlp_blocks(arith, layers, device):
    """{packed weights, bias (2^s b for h2)} per layer as one byte tensor."""
    parts = []
    for W, b in layers:
        parts.append(packed(arith, W, device).cpu().numpy())
        parts.append(np.ascontiguousarray(zscale(arith) * b, dtype=np.float32).view(np.uint8))
    return dev(np.concatenate(parts), device, np.uint8)


@pytest.mark.parametrize("d,n_layers,mask", [(64, 4, 0b0111), (64, 3, 0b111), (32, 4, 0b0111), (32, 1, 0), (64, 2, 0b10)])
@pytest.mark.parametrize("rows", [1, 16, 333, 70000])
@pytest.mark.parametrize("arith", ARITHS)
def test_mlp_fwd_split(cuda_device, arith, d, n_layers, mask, rows):
    rng = np.random.RandomState(rows + d)
    X = rng.randn(rows, d).astype(np.float32)
    layers = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32))
              for _ in range(n_layers)]
    wb = mlp_blocks(arith, layers, cuda_device)
    Y = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    acts = torch.empty((max(n_layers - 1, 1), rows, d), dtype=torch.float32, device=cuda_device)
    task = _lib.MlpTask(_lib.ptr(dev(X, cuda_device)), _lib.ptr(wb), _lib.ptr(Y), _lib.ptr(acts), 0, rows, n_layers, mask,
                        None, None)
    _lib.call_multi("tspgnn_mlp_fwd_multi_" + arith, [task], d)
    torch.cuda.synchronize()
    x = X.astype(np.float64)
    for l, (W, b) in enumerate(layers):
        x = NO.dense(x, W.astype(np.float64), b.astype(np.float64), bool((mask >> l) & 1))
        if l < n_layers - 1:
            assert rel_err(acts[l].cpu().numpy(), x) < 2e-6
    assert rel_err(Y.cpu().numpy(), x) < 2e-6


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("rows", [1, 16, 333, 70000])
@pytest.mark.parametrize("keep_rows", [False, True])
def test_mlp_head_h2(cuda_device, d, rows, keep_rows):
    """The vote head in one launch (model.py:107-115,128: three relu layers, then Dense(1) on the rows in hand)."""
    rng = np.random.RandomState(rows + d + 5)
    X = rng.randn(rows, d).astype(np.float32)
    layers = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32))
              for _ in range(3)]
    hw = (rng.randn(d, 1) / np.sqrt(d)).astype(np.float32)
    hb = np.array([0.3], np.float32)
    wb = mlp_blocks("h2", layers, cuda_device)
    Y = torch.full((rows, d), 7.0, dtype=torch.float32, device=cuda_device)
    y = torch.full((rows,), 7.0, dtype=torch.float32, device=cuda_device)
    flag = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    task = _lib.MlpTask(_lib.ptr(dev(X, cuda_device)), _lib.ptr(wb), _lib.ptr(Y) if keep_rows else None, None, 0, rows, 3,
                        0b111, None, None, _lib.ptr(flag))
    _lib.call("tspgnn_mlp_head_fwd_h2", ctypes.cast(ctypes.pointer(task), ctypes.c_void_p), _lib.ptr(dev(hw, cuda_device)),
              _lib.ptr(dev(hb, cuda_device)), _lib.ptr(y), d, None)
    torch.cuda.synchronize()
    x = X.astype(np.float64)
    for W, b in layers:
        x = NO.dense(x, W.astype(np.float64), b.astype(np.float64), True)
    want = NO.dense(x, hw.astype(np.float64), hb.astype(np.float64), False)[:, 0]
    assert rel_err(y.cpu().numpy(), want) < 2e-6
    if keep_rows:
        assert rel_err(Y.cpu().numpy(), x) < 2e-6
    else:
        assert float(Y.min()) == 7.0 and float(Y.max()) == 7.0
    assert int(flag.item()) == 0


def test_mlp_head_h2_rejects_a_projection_and_null_outputs(cuda_device):
    t = torch.zeros((16, 64), dtype=torch.float32, device=cuda_device)
    wb = torch.zeros(3 * (2 * 64 * 64 * 2 + 64 * 4), dtype=torch.uint8, device=cuda_device)
    bad = _lib.MlpTask(_lib.ptr(t), _lib.ptr(wb), None, None, 0, 16, 3, 7, _lib.ptr(wb), _lib.ptr(t))
    ok = _lib.MlpTask(_lib.ptr(t), _lib.ptr(wb), None, None, 0, 16, 3, 7, None, None)
    for task, y in ((bad, t), (ok, None)):
        with pytest.raises(_lib.TspgnnError):
            _lib.call("tspgnn_mlp_head_fwd_h2", ctypes.cast(ctypes.pointer(task), ctypes.c_void_p), _lib.ptr(t), _lib.ptr(t),
                      _lib.ptr(y), 64, None)


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("arith", ARITHS)
def test_mlp_two_tasks_with_projection(cuda_device, arith, d):
    rng = np.random.RandomState(d)
    rows_a, rows_b = 9000, 700
    Xa = rng.randn(rows_a, d).astype(np.float32)
    Xb = rng.randn(rows_b, d).astype(np.float32)
    la = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(3)]
    lb = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(4)]
    P = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    Ya = torch.empty((rows_a, d), dtype=torch.float32, device=cuda_device)
    Yb = torch.empty((rows_b, d), dtype=torch.float32, device=cuda_device)
    Zb = zx_buffer(arith, rows_b, 4 * d, cuda_device)
    ta = _lib.MlpTask(_lib.ptr(dev(Xa, cuda_device)), _lib.ptr(mlp_blocks(arith, la, cuda_device)), _lib.ptr(Ya), None, 0, rows_a, 3,
                      0b111, None, None)
    tb = _lib.MlpTask(_lib.ptr(dev(Xb, cuda_device)), _lib.ptr(mlp_blocks(arith, lb, cuda_device)), _lib.ptr(Yb), None, 0, rows_b, 4,
                      0b0111, _lib.ptr(packed(arith, P, cuda_device)), _lib.ptr(Zb))
    _lib.call_multi("tspgnn_mlp_fwd_multi_" + arith, [ta, tb], d)
    torch.cuda.synchronize()
    xa = Xa.astype(np.float64)
    for W, b in la:
        xa = NO.dense(xa, W.astype(np.float64), b.astype(np.float64), True)
    xb = Xb.astype(np.float64)
    for l, (W, b) in enumerate(lb):
        xb = NO.dense(xb, W.astype(np.float64), b.astype(np.float64), l < 3)
    assert rel_err(Ya.cpu().numpy(), xa) < 2e-6
    assert rel_err(Yb.cpu().numpy(), xb) < 2e-6
    assert rel_err(zx_out(arith, Zb, rows_b), xb @ P.astype(np.float64)) < 2e-6


def ln_params(rng, d):
    ln = np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]).astype(np.float32)
    names = ("input", "transform", "forget", "output", "state")
    return ln, {g: (ln[i, 0].astype(np.float64), ln[i, 1].astype(np.float64)) for i, g in enumerate(names)}


@pytest.mark.parametrize("d,dx", [(64, 64), (32, 32), (32, 64), (64, 0), (64, 192), (64, 32)])
@pytest.mark.parametrize("rows", [1, 17, 1000, 40000])
@pytest.mark.parametrize("arith", ARITHS)
def test_lnlstm_fwd_split(cuda_device, arith, d, dx, rows):
    """dx+d = 128 at d=64 does not fit LDS in three bf16 pieces, dx+d = 256 not in two fp16 pieces either: the
    streamed (lock-step) mode is exercised too."""
    rng = np.random.RandomState(rows * 7 + d + dx)
    x = rng.randn(rows, dx).astype(np.float32)
    h = rng.randn(rows, d).astype(np.float32)
    c = rng.randn(rows, d).astype(np.float32)
    K = (rng.randn(dx + d, 4 * d) / np.sqrt(dx + d)).astype(np.float32)
    ln, lnd = ln_params(rng, d)
    h_out = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    c_out = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    xd = dev(x, cuda_device) if dx else None
    task = _lib.LstmTask(_lib.ptr(xd), dx, _lib.ptr(dev(h, cuda_device)), _lib.ptr(dev(c, cuda_device)),
                         _lib.ptr(packed(arith, K, cuda_device)), _lib.ptr(dev(ln, cuda_device)), _lib.ptr(h_out), _lib.ptr(c_out),
                         rows, None, None, None, None)
    _lib.call_multi("tspgnn_lnlstm_fwd_multi_" + arith, [task], d)
    torch.cuda.synchronize()
    rh, rc = NO.lnlstm(x.astype(np.float64), h.astype(np.float64), c.astype(np.float64), K.astype(np.float64), lnd)
    assert rel_err(c_out.cpu().numpy(), rc) < 5e-6
    assert rel_err(h_out.cpu().numpy(), rh) < 5e-6


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("arith", ARITHS)
def test_lnlstm_gather_init_and_zbias_tasks_in_one_launch(cuda_device, arith, d):
    rng = np.random.RandomState(d)
    N, M = 300, 5000
    uv = np.stack([rng.randint(0, N, M), rng.randint(0, N, M)], 1).astype(np.int32)
    Zx = rng.randn(N, 4 * d).astype(np.float32)
    he = rng.randn(M, d).astype(np.float32); ce = rng.randn(M, d).astype(np.float32)
    Kh = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    ln_e, lnd_e = ln_params(rng, d)
    # vertex task: z = deg * zbias + [x|h] K
    xv = rng.randn(N, d).astype(np.float32)
    hv = rng.randn(N, d).astype(np.float32); cv = rng.randn(N, d).astype(np.float32)
    Kv = (rng.randn(2 * d, 4 * d) / np.sqrt(2 * d)).astype(np.float32)
    zb = rng.randn(4 * d).astype(np.float32)
    deg = rng.randint(0, 40, N).astype(np.float32)
    ln_v, lnd_v = ln_params(rng, d)
    he_o = torch.empty((M, d), dtype=torch.float32, device=cuda_device); ce_o = torch.empty_like(he_o)
    hv_o = torch.empty((N, d), dtype=torch.float32, device=cuda_device); cv_o = torch.empty_like(hv_o)
    te = _lib.LstmTask(None, 0, _lib.ptr(dev(he, cuda_device)), _lib.ptr(dev(ce, cuda_device)), _lib.ptr(packed(arith, Kh, cuda_device)),
                       _lib.ptr(dev(ln_e, cuda_device)), _lib.ptr(he_o), _lib.ptr(ce_o), M,
                       _lib.ptr(dev(uv, cuda_device, np.int32)), _lib.ptr(dev(zx_in(arith, Zx), cuda_device)), None, None)
    tv = _lib.LstmTask(_lib.ptr(dev(xv, cuda_device)), d, _lib.ptr(dev(hv, cuda_device)), _lib.ptr(dev(cv, cuda_device)),
                       _lib.ptr(packed(arith, Kv, cuda_device)), _lib.ptr(dev(ln_v, cuda_device)), _lib.ptr(hv_o), _lib.ptr(cv_o), N,
                       None, None, _lib.ptr(dev(zb, cuda_device)), _lib.ptr(dev(deg, cuda_device)))
    _lib.call_multi("tspgnn_lnlstm_fwd_multi_" + arith, [te, tv], d)
    torch.cuda.synchronize()
    z0 = Zx.astype(np.float64)[uv[:, 0]] + Zx.astype(np.float64)[uv[:, 1]]
    rh, rc = NO.lnlstm(np.zeros((M, 0)), he.astype(np.float64), ce.astype(np.float64), Kh.astype(np.float64), lnd_e, z0=z0)
    assert rel_err(ce_o.cpu().numpy(), rc) < 5e-6
    assert rel_err(he_o.cpu().numpy(), rh) < 5e-6
    z0 = deg.astype(np.float64)[:, None] * zb.astype(np.float64)[None]
    rh, rc = NO.lnlstm(xv.astype(np.float64), hv.astype(np.float64), cv.astype(np.float64), Kv.astype(np.float64), lnd_v, z0=z0)
    assert rel_err(cv_o.cpu().numpy(), rc) < 5e-6
    assert rel_err(hv_o.cpu().numpy(), rh) < 5e-6


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("arith", ARITHS)
def test_cell_fused_with_next_step_mlp(cuda_device, arith, d):
    """tspgnn_lnlstm_mlp_fwd_multi_{h2,x3}: edge cell (gather-init, resident) + 3-layer message MLP, and vertex cell
    (bias-init, streamed) + 4-layer MLP + projection, one launch; both against the float64 oracle."""
    rng = np.random.RandomState(100 + d)
    N, M = 333, 7001
    uv = np.stack([rng.randint(0, N, M), rng.randint(0, N, M)], 1).astype(np.int32)
    Zx = rng.randn(N, 4 * d).astype(np.float32)
    he = rng.randn(M, d).astype(np.float32); ce = rng.randn(M, d).astype(np.float32)
    Kh = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    ln_e, lnd_e = ln_params(rng, d)
    le = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(3)]
    xv = rng.randn(N, d).astype(np.float32)
    hv = rng.randn(N, d).astype(np.float32); cv = rng.randn(N, d).astype(np.float32)
    Kv = (rng.randn(2 * d, 4 * d) / np.sqrt(2 * d)).astype(np.float32)
    zb = rng.randn(4 * d).astype(np.float32)
    deg = rng.randint(0, 40, N).astype(np.float32)
    ln_v, lnd_v = ln_params(rng, d)
    lv = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(4)]
    P = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    f32 = dict(dtype=torch.float32, device=cuda_device)
    he_o, ce_o, ae_o = torch.empty((M, d), **f32), torch.empty((M, d), **f32), torch.empty((M, d), **f32)
    hv_o, cv_o, yv_o = torch.empty((N, d), **f32), torch.empty((N, d), **f32), torch.empty((N, d), **f32)
    zv_o = zx_buffer(arith, N, 4 * d, cuda_device)
    te = _lib.LstmTask(None, 0, _lib.ptr(dev(he, cuda_device)), _lib.ptr(dev(ce, cuda_device)), _lib.ptr(packed(arith, Kh, cuda_device)),
                       _lib.ptr(dev(ln_e, cuda_device)), _lib.ptr(he_o), _lib.ptr(ce_o), M,
                       _lib.ptr(dev(uv, cuda_device, np.int32)), _lib.ptr(dev(zx_in(arith, Zx), cuda_device)), None, None)
    tv = _lib.LstmTask(_lib.ptr(dev(xv, cuda_device)), d, _lib.ptr(dev(hv, cuda_device)), _lib.ptr(dev(cv, cuda_device)),
                       _lib.ptr(packed(arith, Kv, cuda_device)), _lib.ptr(dev(ln_v, cuda_device)), _lib.ptr(hv_o), _lib.ptr(cv_o), N,
                       None, None, _lib.ptr(dev(zb, cuda_device)), _lib.ptr(dev(deg, cuda_device)))
    # f16x2 (the training forward): the hidden activations of both MLPs are saved as well; the vertex task's with an
    # explicit layer stride, the edge task's with the default rows * d
    acts_e = torch.zeros((2, M, d), **f32) if arith == "h2" else None
    acts_v = torch.zeros((3, 2, N, d), **f32) if arith == "h2" else None
    tasks = [_lib.CellMlpTask(te, _lib.ptr(mlp_blocks(arith, le, cuda_device)), 3, 0b111, _lib.ptr(ae_o), None, None, 0, 0,
                              _lib.ptr(acts_e), 0),
             _lib.CellMlpTask(tv, _lib.ptr(mlp_blocks(arith, lv, cuda_device)), 4, 0b0111, _lib.ptr(yv_o),
                              _lib.ptr(packed(arith, P, cuda_device)), _lib.ptr(zv_o), 0, 0,
                              _lib.ptr(acts_v[:, 1]) if arith == "h2" else None, 2 * N * d)]
    _lib.call_multi("tspgnn_lnlstm_mlp_fwd_multi_" + arith, tasks, d)
    torch.cuda.synchronize()
    z0 = Zx.astype(np.float64)[uv[:, 0]] + Zx.astype(np.float64)[uv[:, 1]]
    rh, rc = NO.lnlstm(np.zeros((M, 0)), he.astype(np.float64), ce.astype(np.float64), Kh.astype(np.float64), lnd_e, z0=z0)
    assert rel_err(ce_o.cpu().numpy(), rc) < 5e-6 and rel_err(he_o.cpu().numpy(), rh) < 5e-6
    a = rh
    for l, (W, b) in enumerate(le):
        a = NO.dense(a, W.astype(np.float64), b.astype(np.float64), True)
        if arith == "h2" and l < 2:
            assert rel_err(acts_e[l].cpu().numpy(), a) < 5e-6
    assert rel_err(ae_o.cpu().numpy(), a) < 5e-6
    z0 = deg.astype(np.float64)[:, None] * zb.astype(np.float64)[None]
    rh, rc = NO.lnlstm(xv.astype(np.float64), hv.astype(np.float64), cv.astype(np.float64), Kv.astype(np.float64), lnd_v, z0=z0)
    assert rel_err(cv_o.cpu().numpy(), rc) < 5e-6 and rel_err(hv_o.cpu().numpy(), rh) < 5e-6
    y = rh
    for l, (W, b) in enumerate(lv):
        y = NO.dense(y, W.astype(np.float64), b.astype(np.float64), l < 3)
        if arith == "h2" and l < 3:
            assert rel_err(acts_v[l, 1].cpu().numpy(), y) < 5e-6
    if arith == "h2":
        assert float(acts_v[:, 0].abs().max()) == 0.0          # the other time slot of the strided buffer is untouched
    assert rel_err(yv_o.cpu().numpy(), y) < 5e-6
    assert rel_err(zx_out(arith, zv_o, N), y @ P.astype(np.float64)) < 5e-6


def test_cell_fused_x3_rejects_activation_saving(cuda_device):
    d, rows = 32, 16
    f32 = dict(dtype=torch.float32, device=cuda_device)
    z, ho, co = torch.zeros((rows, 4 * d), **f32), torch.zeros((rows, d), **f32), torch.zeros((rows, d), **f32)
    t = _lib.LstmTask(_lib.ptr(z), d, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(ho), _lib.ptr(co), rows,
                      None, None, None, None)
    task = _lib.CellMlpTask(t, _lib.ptr(z), 2, 0b11, _lib.ptr(z), None, None, 0, 0, _lib.ptr(z), 0)
    with pytest.raises(_lib.TspgnnError, match="f16x2 feature"):
        _lib.call_multi("tspgnn_lnlstm_mlp_fwd_multi_x3", [task], d)


@pytest.mark.parametrize("arith", ARITHS)
def test_split_kernels_reject_unsupported_width(cuda_device, arith):
    t = _lib.MlpTask(None, None, None, None, 0, 0, 1, 0, None, None)
    with pytest.raises(_lib.TspgnnError):
        _lib.call_multi("tspgnn_mlp_fwd_multi_" + arith, [t], 128)


# ---------------------------------------------------------------------------------------- f16x2 range guard
# fp16 pieces overflow where the reference's fp32 (graphnn.py:18) does not: tspgnn_pack_weights_h2 reports max |2^s W|,
# the forward kernels raise their task's range_flag when an operand they split reaches 65504 (include/tspgnn.h).

def test_pack_h2_reports_the_largest_scaled_weight(cuda_device):
    sc = zscale("h2")
    rng = np.random.RandomState(5)
    W = (0.3 * rng.randn(64, 256)).astype(np.float32)
    W[17, 133] = -1500.0     # 2^6 * 1500 = 96000 > 65504: the hi piece is -inf
    word = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    src = dev(W, cuda_device)
    out = torch.empty(2 * W.size * 2, dtype=torch.uint8, device=cuda_device)
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(src), _lib.ptr(out), 64, 256, _lib.ptr(word), None)
    torch.cuda.synchronize()
    assert np.int32(word.item()).view(np.float32) == np.float32(sc * 1500.0)
    assert word.item() >= 0x477fe000                       # the bits of 65504.0f: out of range
    # a second, in-range matrix only ever RAISES the word (atomic max), and a fresh word stays below the limit
    W2 = (0.3 * rng.randn(32, 32)).astype(np.float32)
    word2 = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(dev(W2, cuda_device)), _lib.ptr(out), 32, 32, _lib.ptr(word2), None)
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(dev(W2, cuda_device)), _lib.ptr(out), 32, 32, _lib.ptr(word), None)
    torch.cuda.synchronize()
    assert np.int32(word2.item()).view(np.float32) == np.float32(sc * np.abs(W2).max()) and word2.item() < 0x46ffe000
    assert np.int32(word.item()).view(np.float32) == np.float32(sc * 1500.0)
    Wn = W2.copy(); Wn[3, 3] = np.inf
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(dev(Wn, cuda_device)), _lib.ptr(out), 32, 32, _lib.ptr(word2), None)
    torch.cuda.synchronize()
    assert word2.item() >= 0x7f800000                      # a non-finite weight sorts above everything


@pytest.mark.parametrize("big,hit", [(65000.0, False), (65519.0, False), (65520.0, True), (-7.0e4, True), (3.0e38, True)])
def test_mlp_h2_flags_an_operand_outside_the_fp16_range(cuda_device, big, hit):
    """One activation of one row at / beyond the largest value whose hi piece is finite (65504 = fp16 max; 65520 rounds
    to inf): the launch sets bit 0 of the task's range_flag -- and leaves it alone otherwise."""
    d, rows, n_layers, mask = 64, 777, 3, 0b111
    rng = np.random.RandomState(int(abs(big)) % 1000)
    X = rng.randn(rows, d).astype(np.float32)
    X[401, 13] = big
    layers = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32))
              for _ in range(n_layers)]
    wb = mlp_blocks("h2", layers, cuda_device)
    Y = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    flag = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    task = _lib.MlpTask(_lib.ptr(dev(X, cuda_device)), _lib.ptr(wb), _lib.ptr(Y), None, 0, rows, n_layers, mask, None, None,
                        _lib.ptr(flag))
    _lib.call_multi("tspgnn_mlp_fwd_multi_h2", [task], d)
    torch.cuda.synchronize()
    assert bool(flag.item() & 1) == hit
    if not hit:   # in range: fp32-class result on the row with the large entry as well
        x = X.astype(np.float64)
        for l, (W, b) in enumerate(layers):
            x = NO.dense(x, W.astype(np.float64), b.astype(np.float64), bool((mask >> l) & 1))
        assert rel_err(Y.cpu().numpy(), x) < 2e-6


def test_cell_h2_flags_an_aggregate_outside_the_fp16_range(cuda_device):
    """The vertex cell of the pushed form takes the row-SUM of up to n-1 relu activations as its GEMM operand: an entry
    past 65504 must raise the flag (the cell's x operand), an h past it likewise."""
    d, rows = 64, 100
    rng = np.random.RandomState(3)
    K = (rng.randn(2 * d, 4 * d) / np.sqrt(2 * d)).astype(np.float32)
    ln = np.concatenate([np.ones(d), np.zeros(d)] * 5).astype(np.float32)
    for where in ("x", "h", None):
        x = np.abs(rng.randn(rows, d)).astype(np.float32) * 100
        h = np.abs(rng.randn(rows, d)).astype(np.float32)
        if where == "x":
            x[57, 9] = 1.2e5
        if where == "h":
            h[3, 60] = 9.9e4
        c = rng.randn(rows, d).astype(np.float32)
        flag = torch.zeros(1, dtype=torch.int32, device=cuda_device)
        ho = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
        co = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
        task = _lib.LstmTask(_lib.ptr(dev(x, cuda_device)), d, _lib.ptr(dev(h, cuda_device)), _lib.ptr(dev(c, cuda_device)),
                             _lib.ptr(packed("h2", K, cuda_device)), _lib.ptr(dev(ln, cuda_device)), _lib.ptr(ho), _lib.ptr(co),
                             rows, None, None, None, None, _lib.ptr(flag))
        _lib.call_multi("tspgnn_lnlstm_fwd_multi_h2", [task], d)
        torch.cuda.synchronize()
        assert bool(flag.item() & 1) == (where is not None), where
