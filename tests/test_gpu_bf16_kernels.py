"""bf16-storage / fp32-accumulate kernels (BASELINE config 5) vs a float64 NumPy restatement that rounds to bf16
at the same points: inputs, weights, every stored activation.  Quantities that are not rounded on the way (the
cell state c') are held to the fp32 bar; bf16 outputs may differ by one bf16 ulp (2^-8 relative) where the two
sides land on different sides of a rounding boundary."""
import numpy as np
import pytest
import torch

from conftest import h2_zx_pack, h2_zx_unpack, rel_err
from oracle import np_oracle as NO
from tspgnn import _lib

pytestmark = pytest.mark.gpu

_KEEP = []
BF16_TOL = 1.2e-2     # max-norm relative: a few bf16 ulps through a 4-layer chain
F32_TOL = 5e-6


def rb(x):
    """Round to bf16 (nearest even), back in float64."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def dev(a, device, dtype=np.float32):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)
    _KEEP.append(t)
    return t


def dev_bf16(a, device):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).to(device)
    _KEEP.append(t)
    return t


def f64(t):
    return t.to(torch.float64).cpu().numpy()


@pytest.fixture(autouse=True)
def _release_uploads():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def packed_bf16(W, device):
    """Piece 0 of tspgnn_pack_weights_x3 = the weights rounded to bf16 in MFMA fragment order (bytes)."""
    src = dev(W, device)
    out = torch.empty(3 * W.size * 2, dtype=torch.uint8, device=device)
    _KEEP.append(out)
    _lib.call("tspgnn_pack_weights_x3", _lib.ptr(src), _lib.ptr(out), W.shape[0], W.shape[1], None)
    return out[:W.size * 2]


@pytest.mark.parametrize("d", [32, 64, 128, 24])
def test_gather2_and_rowsum_bf16(cuda_device, d):
    rng = np.random.RandomState(d)
    N, M = 257, 3001
    uv = np.stack([rng.randint(0, N, M), rng.randint(0, N, M)], 1).astype(np.int32)
    X = rng.randn(N, d)
    Y = torch.empty((M, d), dtype=torch.bfloat16, device=cuda_device)
    _lib.call("tspgnn_gather2_sum_bf16", _lib.ptr(dev(uv, cuda_device, np.int32)), _lib.ptr(dev_bf16(X, cuda_device)), _lib.ptr(Y),
              M, N, d, None)
    torch.cuda.synchronize()
    assert np.array_equal(f64(Y), rb(rb(X)[uv[:, 0]] + rb(X)[uv[:, 1]]))      # one fp32 add, one rounding: exact
    if d == 24:
        return
    counts = rng.randint(0, 50, N)
    counts[3] = 0
    rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    eid = rng.randint(0, M, int(rowptr[-1])).astype(np.int32)
    Z = rng.randn(M, d)
    out = torch.empty((N, d), dtype=torch.bfloat16, device=cuda_device)
    _lib.call("tspgnn_csr_rowsum_bf16", _lib.ptr(dev(rowptr, cuda_device, np.int32)), _lib.ptr(dev(eid, cuda_device, np.int32)),
              _lib.ptr(dev_bf16(Z, cuda_device)), _lib.ptr(out), N, M, d, None)
    torch.cuda.synchronize()
    ref = np.stack([rb(Z)[eid[rowptr[r]:rowptr[r + 1]]].sum(0) for r in range(N)])
    got = f64(out)
    assert np.all(got[3] == 0)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 2.0 ** -8 + 1e-6       # one bf16 rounding


def mlp_blocks_bf16(layers, device):
    parts = []
    for W, b in layers:
        parts.append(packed_bf16(W, device).cpu().numpy())
        parts.append(np.ascontiguousarray(b, dtype=np.float32).view(np.uint8))
    return dev(np.concatenate(parts), device, np.uint8)


@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("rows", [1, 333, 20000])
def test_mlp_bf16_two_tasks_with_projection(cuda_device, d, rows):
    rng = np.random.RandomState(d + rows)
    rows_b = max(1, rows // 7)
    Xa, Xb = rng.randn(rows, d), rng.randn(rows_b, d)
    la = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(3)]
    lb = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(4)]
    P = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    Ya = torch.empty((rows, d), dtype=torch.bfloat16, device=cuda_device)
    Yb = torch.empty((rows_b, d), dtype=torch.bfloat16, device=cuda_device)
    Zb = torch.zeros(((rows_b + 15) // 16 * 16, 4 * d), dtype=torch.bfloat16, device=cuda_device)   # blocked by 16 rows
    # task a: X handed over BLOCKED by 16 rows (a loop state between two steps) and the hidden activations saved (training)
    acts_a = torch.zeros((2, rows, d), dtype=torch.bfloat16, device=cuda_device)
    ta = _lib.MlpTaskB(_lib.ptr(dev_bf16(h2_zx_pack(rb(Xa), 1.0), cuda_device)), _lib.ptr(mlp_blocks_bf16(la, cuda_device)), _lib.ptr(Ya),
                       rows, 3, 0b111, None, None, _lib.ptr(acts_a), 0, 1)
    tb = _lib.MlpTaskB(_lib.ptr(dev_bf16(Xb, cuda_device)), _lib.ptr(mlp_blocks_bf16(lb, cuda_device)), _lib.ptr(Yb), rows_b, 4,
                       0b0111, _lib.ptr(packed_bf16(P, cuda_device)), _lib.ptr(Zb))
    _lib.call_multi("tspgnn_mlp_fwd_multi_bf16", [ta, tb], d)
    torch.cuda.synchronize()

    hidden = []

    def chain(x, layers, relus):
        x = rb(x)
        for (W, b), r in zip(layers, relus):
            x = rb(NO.dense(x, rb(W), b.astype(np.float64), r))
            hidden.append(x)
        return x
    ra, rbb = chain(Xa, la, [True] * 3), chain(Xb, lb, [True, True, True, False])
    assert rel_err(f64(acts_a[0]), hidden[0]) < BF16_TOL and rel_err(f64(acts_a[1]), hidden[1]) < BF16_TOL
    assert rel_err(f64(Ya), ra) < BF16_TOL
    assert rel_err(f64(Yb), rbb) < BF16_TOL
    assert rel_err(h2_zx_unpack(f64(Zb), rows_b, 1.0), f64(Yb) @ rb(P)) < 2.0 ** -7       # projection of the kernel's own (stored) Y: one rounding


@pytest.mark.parametrize("d", [32, 64, 128])
@pytest.mark.parametrize("rows", [1, 333, 5003])
def test_mlp_bf16_interleaved_last_layer_stores_the_same_rows(cuda_device, d, rows):
    """tspgnn_mlp_task_bf16.y_interleaved: the last layer packed with its output columns permuted (packed column
    16t+4g+j = true column 32(t/2)+8g+4(t%2)+j) and Y written in 16-byte pieces -- the same row-major Y, bit for bit."""
    rng = np.random.RandomState(3 * d + rows)
    X = rng.randn(rows, d)
    layers = [((rng.randn(d, d) / np.sqrt(d)).astype(np.float32), (0.1 * rng.randn(d)).astype(np.float32)) for _ in range(3)]
    c = np.arange(d)
    t, g, r = c // 16, (c % 16) // 4, c % 4
    perm = 32 * (t // 2) + 8 * g + 4 * (t % 2) + r
    inter = layers[:-1] + [(np.ascontiguousarray(layers[-1][0][:, perm]), np.ascontiguousarray(layers[-1][1][perm]))]
    outs = []
    for lay, flag in ((layers, 0), (inter, 1)):
        Y = torch.zeros((rows, d), dtype=torch.bfloat16, device=cuda_device)
        task = _lib.MlpTaskB(_lib.ptr(dev_bf16(X, cuda_device)), _lib.ptr(mlp_blocks_bf16(lay, cuda_device)), _lib.ptr(Y), rows, 3,
                             0b011, None, None, None, 0, 0, flag)
        _lib.call_multi("tspgnn_mlp_fwd_multi_bf16", [task], d)
        torch.cuda.synchronize()
        outs.append(f64(Y))
    assert np.array_equal(outs[0], outs[1])
    x = rb(X)
    for (W, b), relu in zip(layers, [True, True, False]):
        x = rb(NO.dense(x, rb(W), b.astype(np.float64), relu))
    assert rel_err(outs[1], x) < BF16_TOL


def ln_params(rng, d):
    ln = np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]).astype(np.float32)
    names = ("input", "transform", "forget", "output", "state")
    return ln, {g: (ln[i, 0].astype(np.float64), ln[i, 1].astype(np.float64)) for i, g in enumerate(names)}


@pytest.mark.parametrize("c_blocked", [(False, False), (True, True), (False, True)], ids=["rowmajor", "blocked", "rm-to-blocked"])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_lnlstm_bf16_gather_and_plain_tasks(cuda_device, d, c_blocked):
    """Edge-style task (gather-init from the blocked bf16 Zx, Kh resident) and vertex-style task (x|h with the [2d,4d]
    kernel, streamed through LDS at d=128) in one launch; the states h, c row-major or blocked by 16 rows on either side."""
    def c_dev(c, blocked):
        return dev(h2_zx_pack(c, 1.0) if blocked else c, cuda_device)

    def h_dev(h, blocked):
        return dev_bf16(h2_zx_pack(rb(h), 1.0) if blocked else h, cuda_device)

    def h_host(t, rows, blocked):
        return h2_zx_unpack(f64(t), rows, 1.0) if blocked else f64(t)[:rows]

    def c_host(t, rows, blocked):
        return h2_zx_unpack(t.cpu().numpy(), rows, 1.0) if blocked else t.cpu().numpy()[:rows]
    cin, cout = c_blocked
    rng = np.random.RandomState(7 + d)
    N, M = 301, 5003
    uv = np.stack([rng.randint(0, N, M), rng.randint(0, N, M)], 1).astype(np.int32)
    Zx = rng.randn(N, 4 * d)
    he, ce = rng.randn(M, d), rng.randn(M, d).astype(np.float32)
    Kh = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    ln_e, lnd_e = ln_params(rng, d)
    xv, hv, cv = rng.randn(N, d), rng.randn(N, d), rng.randn(N, d).astype(np.float32)
    Kv = (rng.randn(2 * d, 4 * d) / np.sqrt(2 * d)).astype(np.float32)
    ln_v, lnd_v = ln_params(rng, d)
    he_o = torch.zeros(((M + 15) // 16 * 16, d), dtype=torch.bfloat16, device=cuda_device)
    pad = lambda r: (r + 15) // 16 * 16
    ce_o = torch.zeros((pad(M), d), dtype=torch.float32, device=cuda_device)
    hv_o = torch.zeros((pad(N), d), dtype=torch.bfloat16, device=cuda_device)
    cv_o = torch.zeros((pad(N), d), dtype=torch.float32, device=cuda_device)
    te = _lib.LstmTaskB(None, 0, _lib.ptr(h_dev(he, cin)), _lib.ptr(c_dev(ce, cin)), _lib.ptr(packed_bf16(Kh, cuda_device)),
                        _lib.ptr(dev(ln_e, cuda_device)), _lib.ptr(he_o), _lib.ptr(ce_o), M,
                        _lib.ptr(dev(uv, cuda_device, np.int32)), _lib.ptr(dev_bf16(h2_zx_pack(rb(Zx), 1.0), cuda_device)),
                        int(cin), int(cout))
    tv = _lib.LstmTaskB(_lib.ptr(dev_bf16(xv, cuda_device)), d, _lib.ptr(h_dev(hv, cin)), _lib.ptr(c_dev(cv, cin)),
                        _lib.ptr(packed_bf16(Kv, cuda_device)), _lib.ptr(dev(ln_v, cuda_device)), _lib.ptr(hv_o), _lib.ptr(cv_o), N,
                        None, None, int(cin), int(cout))
    _lib.call_multi("tspgnn_lnlstm_fwd_multi_bf16", [te, tv], d)
    torch.cuda.synchronize()
    z0 = rb(Zx)[uv[:, 0]] + rb(Zx)[uv[:, 1]]
    rh, rc = NO.lnlstm(np.zeros((M, 0)), rb(he), ce.astype(np.float64), rb(Kh), lnd_e, z0=z0)
    assert rel_err(c_host(ce_o, M, cout), rc) < F32_TOL            # nothing is rounded on the way to c'
    assert rel_err(h_host(he_o, M, cout), rb(rh)) < 2.0 ** -7
    rh, rc = NO.lnlstm(rb(xv), rb(hv), cv.astype(np.float64), rb(Kv), lnd_v)
    assert rel_err(c_host(cv_o, N, cout), rc) < F32_TOL
    assert rel_err(h_host(hv_o, N, cout), rb(rh)) < 2.0 ** -7


@pytest.mark.parametrize("n", [0, 5, 8, 1000003])
def test_storage_conversions(cuda_device, n):
    """tspgnn_convert_f32_to_bf16 / _bf16_to_f32: round to nearest even exactly as the reference type conversion, widening
    exact; tails that are not a multiple of 8."""
    rng = np.random.RandomState(n + 1)
    x = (rng.randn(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    if n >= 5:
        x[:5] = [0.0, -0.0, 1.00390625, 1.01171875, -65504.0]     # ties between two bf16 values among them
    xd = dev(x, cuda_device)
    y = torch.full((max(n, 1),), 7.0, dtype=torch.bfloat16, device=cuda_device)
    _lib.call("tspgnn_convert_f32_to_bf16", _lib.ptr(xd) if n else None, _lib.ptr(y), n, None)
    torch.cuda.synchronize()
    want = torch.from_numpy(x).to(torch.bfloat16)
    assert torch.equal(y[:n].cpu(), want)
    back = torch.full((max(n, 1),), 7.0, dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_convert_bf16_to_f32", _lib.ptr(y), _lib.ptr(back), n, None)
    torch.cuda.synchronize()
    assert torch.equal(back[:n].cpu(), want.to(torch.float32))


@pytest.mark.parametrize("d", [64, 128])
def test_lnlstm_bf16_null_cell_state_is_zero(cuda_device, d):
    """tspgnn_lstm_task_bf16.c == NULL (no LSTM_initial_states at the first step of a run): bit-identical to a zero array."""
    rng = np.random.RandomState(d)
    N, M = 77, 1003
    uv = np.stack([rng.randint(0, N, M), rng.randint(0, N, M)], 1).astype(np.int32)
    Zx, he = rng.randn(N, 4 * d), rng.randn(M, d)
    Kh = (rng.randn(d, 4 * d) / np.sqrt(d)).astype(np.float32)
    ln_e, _ = ln_params(rng, d)
    xv, hv = rng.randn(N, d), rng.randn(N, d)
    Kv = (rng.randn(2 * d, 4 * d) / np.sqrt(2 * d)).astype(np.float32)
    ln_v, _ = ln_params(rng, d)
    outs = []
    for null_c in (False, True):
        ce = None if null_c else dev(np.zeros((M, d), np.float32), cuda_device)
        cv = None if null_c else dev(np.zeros((N, d), np.float32), cuda_device)
        he_o = torch.zeros((M, d), dtype=torch.bfloat16, device=cuda_device)
        ce_o = torch.zeros((M, d), dtype=torch.float32, device=cuda_device)
        hv_o = torch.zeros((N, d), dtype=torch.bfloat16, device=cuda_device)
        cv_o = torch.zeros((N, d), dtype=torch.float32, device=cuda_device)
        te = _lib.LstmTaskB(None, 0, _lib.ptr(dev_bf16(he, cuda_device)), _lib.ptr(ce), _lib.ptr(packed_bf16(Kh, cuda_device)),
                            _lib.ptr(dev(ln_e, cuda_device)), _lib.ptr(he_o), _lib.ptr(ce_o), M,
                            _lib.ptr(dev(uv, cuda_device, np.int32)), _lib.ptr(dev_bf16(h2_zx_pack(rb(Zx), 1.0), cuda_device)), 0, 0)
        tv = _lib.LstmTaskB(_lib.ptr(dev_bf16(xv, cuda_device)), d, _lib.ptr(dev_bf16(hv, cuda_device)), _lib.ptr(cv),
                            _lib.ptr(packed_bf16(Kv, cuda_device)), _lib.ptr(dev(ln_v, cuda_device)), _lib.ptr(hv_o), _lib.ptr(cv_o),
                            N, None, None, 0, 0)
        _lib.call_multi("tspgnn_lnlstm_fwd_multi_bf16", [te, tv], d)
        torch.cuda.synchronize()
        outs.append((he_o, ce_o, hv_o, cv_o))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
