"""HIP kernels vs the CPU oracle, kernel by kernel, through the C ABI (ctypes)."""
import numpy as np
import pytest
import torch

from conftest import load_pack, rel_err
from oracle import np_oracle as NO
from tspgnn import _lib
from tspgnn.instance_loader import SparseEV, synthetic_batch

pytestmark = pytest.mark.gpu

F32_TOL = 2e-6   # single fp32 op chains (<= a few hundred fused multiply-adds per output)


_KEEP = []


def dev(a, device, dtype=np.float32):
    """Upload; the tensor is kept alive until the end of the test (an inline temporary would be
    freed -- and its block reused by the next upload -- before the asynchronous kernel runs)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_uploads():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def packed(W, device):
    """tspgnn_pack_weights_f32 of a host matrix -> device tensor (kept alive by dev())."""
    src = dev(W, device)
    out = torch.empty_like(src)
    _KEEP.append(out)
    _lib.call("tspgnn_pack_weights_f32", _lib.ptr(src), _lib.ptr(out), W.shape[0], W.shape[1], 0, None)
    return out


def test_pack_weights_is_a_permutation(cuda_device):
    for kr, nc in ((64, 64), (32, 32), (128, 256), (48, 128), (256, 512)):
        W = np.arange(kr * nc, dtype=np.float32).reshape(kr, nc)
        P_ = packed(W, cuda_device).cpu().numpy().reshape(-1)
        assert np.array_equal(np.sort(P_), W.reshape(-1))
        # spot check the documented formula
        U = nc // 64 if nc != 32 else 0
        for i in (0, 5, 77, kr * nc - 1):
            if nc == 32:
                tt, jl, u, sg = i & 1, (i >> 1) & 15, 0, i >> 5
            else:
                tt, jl, u, sg = i & 3, (i >> 2) & 15, (i >> 6) % U, (i >> 6) // U
            g, s_ = sg & 3, sg >> 2
            krow = ((s_ >> 2) << 4) + (g << 2) + (s_ & 3)
            assert P_[i] == W[krow, (u * 4 + tt) * 16 + jl]


def test_library_is_loaded_in_tree():
    import os
    assert os.path.samefile(os.path.dirname(_lib.LIB_PATH), os.path.dirname(_lib.__file__))
    assert _lib.lib.tspgnn_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("d", [32, 64, 128, 20])
@pytest.mark.parametrize("name", ["ragged_B6", "sparse_B4", "n20_B32"])
def test_gather2_and_rowsum(cuda_device, name, d):
    g = load_pack(name, 1)
    ev = SparseEV(g["ev_uv"], int(g["ev_shape"][1]))
    M, N = ev.shape
    rng = np.random.RandomState(0)
    X = rng.randn(N, d).astype(np.float32)
    Z = rng.randn(M, d).astype(np.float32)
    uv = dev(ev.uv, cuda_device, np.int32)
    Y = torch.empty((M, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(uv), _lib.ptr(dev(X, cuda_device)), _lib.ptr(Y), M, N, d, None)
    torch.cuda.synchronize()
    assert np.array_equal(Y.cpu().numpy(), NO.gather2_sum(ev.uv.astype(np.int64), X))   # one add: bit exact
    rowptr, eid = ev.csr_by_vertex()
    out = torch.empty((N, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(dev(rowptr, cuda_device, np.int32)),
              _lib.ptr(dev(eid, cuda_device, np.int32)), _lib.ptr(dev(Z, cuda_device)), _lib.ptr(out), N, M, d, None)
    torch.cuda.synchronize()
    ref = NO.rowsum_by_vertex(ev.uv.astype(np.int64), Z.astype(np.float64), N)
    assert rel_err(out.cpu().numpy(), ref) < F32_TOL
    # adjoint identity <EV x, z> = <x, EV^T z> on the device results
    lhs = float((Y.double() * dev(Z, cuda_device).double()).sum())
    rhs = float((dev(X, cuda_device).double() * out.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_spmm_pair_equals_the_two_kernels(cuda_device, d):
    g = load_pack("ragged_B6", 0)
    ev = SparseEV(g["ev_uv"], int(g["ev_shape"][1]))
    M, N = ev.shape
    rng = np.random.RandomState(d)
    Xv = dev(rng.randn(N, d), cuda_device); Xe = dev(rng.randn(M, d), cuda_device)
    rowptr, eid = ev.csr_by_vertex()
    uv = dev(ev.uv, cuda_device, np.int32); rp = dev(rowptr, cuda_device, np.int32); ei = dev(eid, cuda_device, np.int32)
    Ye1 = torch.empty((M, d), device=cuda_device); Yv1 = torch.empty((N, d), device=cuda_device)
    Ye2 = torch.empty_like(Ye1); Yv2 = torch.empty_like(Yv1)
    _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(uv), _lib.ptr(Xv), _lib.ptr(Ye1), M, N, d, None)
    _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(rp), _lib.ptr(ei), _lib.ptr(Xe), _lib.ptr(Yv1), N, M, d, None)
    _lib.call("tspgnn_spmm_pair_f32", _lib.ptr(uv), _lib.ptr(Xv), _lib.ptr(Ye2), _lib.ptr(rp), _lib.ptr(ei), _lib.ptr(Xe),
              _lib.ptr(Yv2), M, N, d, None)
    torch.cuda.synchronize()
    assert torch.equal(Ye1, Ye2) and torch.equal(Yv1, Yv2)      # same per-row arithmetic: bit exact


def test_rowsum_high_degree_and_empty_rows(cuda_device):
    # n=200 complete graph: degree 199 (> one wavefront of edge ids); plus isolated vertices
    n = 200
    iu = np.triu_indices(n, 1)
    uv = np.stack(iu, 1).astype(np.int32)
    ev = SparseEV(uv, n + 3)            # 3 trailing vertices with no edges
    rowptr, eid = ev.csr_by_vertex()
    M, N, d = uv.shape[0], n + 3, 64
    Z = np.random.RandomState(1).randn(M, d).astype(np.float32)
    out = torch.full((N, d), 7.0, dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(dev(rowptr, cuda_device, np.int32)),
              _lib.ptr(dev(eid, cuda_device, np.int32)), _lib.ptr(dev(Z, cuda_device)), _lib.ptr(out), N, M, d, None)
    torch.cuda.synchronize()
    ref = NO.rowsum_by_vertex(uv.astype(np.int64), Z.astype(np.float64), N)
    assert rel_err(out.cpu().numpy(), ref) < F32_TOL
    assert np.all(out.cpu().numpy()[n:] == 0)


def test_csr_spmm_valued(cuda_device):
    rng = np.random.RandomState(2)
    R, C, d = 37, 23, 64
    A = rng.randn(R, C) * (rng.rand(R, C) < 0.3)
    r, c = np.nonzero(A)
    rowptr = np.zeros(R + 1, dtype=np.int32); rowptr[1:] = np.cumsum(np.bincount(r, minlength=R))
    X = rng.randn(C, d).astype(np.float32)
    out = torch.empty((R, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_csr_spmm_f32", _lib.ptr(dev(rowptr, cuda_device, np.int32)), _lib.ptr(dev(c, cuda_device, np.int32)),
              _lib.ptr(dev(A[r, c], cuda_device)), _lib.ptr(dev(X, cuda_device)), _lib.ptr(out), R, C, d, None)
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), A.astype(np.float32).astype(np.float64) @ X.astype(np.float64)) < F32_TOL


@pytest.mark.parametrize("d,n_layers,mask", [(64, 4, 0b0111), (64, 3, 0b111), (32, 4, 0b0111), (32, 1, 0),
                                             (128, 2, 0b11), (128, 1, 0), (64, 2, 0b10)])
@pytest.mark.parametrize("rows", [1, 16, 333, 5000])
def test_mlp_fwd(cuda_device, d, n_layers, mask, rows):
    rng = np.random.RandomState(rows + d)
    X = rng.randn(rows, d).astype(np.float32)
    layers, flat = [], []
    for l in range(n_layers):
        W = (rng.randn(d, d) / np.sqrt(d)).astype(np.float32)
        b = (0.1 * rng.randn(d)).astype(np.float32)
        layers.append((W.astype(np.float64), b.astype(np.float64)))
        flat += [packed(W, cuda_device).cpu().numpy().reshape(-1), b]
    acts_flags = [bool((mask >> l) & 1) for l in range(n_layers)]
    Y = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    acts = torch.empty((max(n_layers - 1, 1), rows, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_mlp_fwd_f32", _lib.ptr(dev(X, cuda_device)), _lib.ptr(dev(np.concatenate(flat), cuda_device)),
              _lib.ptr(Y), _lib.ptr(acts), 0, rows, d, n_layers, mask, None)
    torch.cuda.synchronize()
    x = X.astype(np.float64)
    for l, ((W, b), a) in enumerate(zip(layers, acts_flags)):
        x = NO.dense(x, W, b, a)
        if l < n_layers - 1:
            assert rel_err(acts[l].cpu().numpy(), x) < F32_TOL
    assert rel_err(Y.cpu().numpy(), x) < F32_TOL


@pytest.mark.parametrize("d,dx", [(64, 64), (32, 32), (32, 64), (64, 0), (32, 16), (128, 128), (64, 192)])
@pytest.mark.parametrize("rows", [1, 17, 1000])
def test_lnlstm_fwd(cuda_device, d, dx, rows):
    rng = np.random.RandomState(rows * 7 + d + dx)
    x = rng.randn(rows, dx).astype(np.float32)
    h = rng.randn(rows, d).astype(np.float32)
    c = rng.randn(rows, d).astype(np.float32)
    K = (rng.randn(dx + d, 4 * d) / np.sqrt(dx + d)).astype(np.float32)
    ln = np.stack([np.stack([1 + 0.2 * rng.randn(d), 0.2 * rng.randn(d)]) for _ in range(5)]).astype(np.float32)
    h_out = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    c_out = torch.empty((rows, d), dtype=torch.float32, device=cuda_device)
    xd = dev(x, cuda_device) if dx else None
    _lib.call("tspgnn_lnlstm_fwd_f32", _lib.ptr(xd), dx, _lib.ptr(dev(h, cuda_device)), _lib.ptr(dev(c, cuda_device)),
              _lib.ptr(packed(K, cuda_device)), _lib.ptr(dev(ln, cuda_device)), _lib.ptr(h_out), _lib.ptr(c_out), rows, d, None)
    torch.cuda.synchronize()
    names = ("input", "transform", "forget", "output", "state")
    lnd = {g: (ln[i, 0].astype(np.float64), ln[i, 1].astype(np.float64)) for i, g in enumerate(names)}
    rh, rc = NO.lnlstm(x.astype(np.float64), h.astype(np.float64), c.astype(np.float64), K.astype(np.float64), lnd)
    assert rel_err(c_out.cpu().numpy(), rc) < 5e-6
    assert rel_err(h_out.cpu().numpy(), rh) < 5e-6


def test_lnlstm_known_answer_zero_kernel(cuda_device):
    d, rows = 64, 40
    rng = np.random.RandomState(0)
    h = rng.randn(rows, d).astype(np.float32); c = rng.randn(rows, d).astype(np.float32)
    x = rng.randn(rows, d).astype(np.float32)
    ln = np.tile(np.stack([np.ones(d), np.zeros(d)])[None], (5, 1, 1)).astype(np.float32)
    h_out = torch.empty((rows, d), dtype=torch.float32, device=cuda_device); c_out = torch.empty_like(h_out)
    _lib.call("tspgnn_lnlstm_fwd_f32", _lib.ptr(dev(x, cuda_device)), d, _lib.ptr(dev(h, cuda_device)),
              _lib.ptr(dev(c, cuda_device)), _lib.ptr(torch.zeros((2 * d, 4 * d), device=cuda_device)),
              _lib.ptr(dev(ln, cuda_device)), _lib.ptr(h_out), _lib.ptr(c_out), rows, d, None)
    torch.cuda.synchronize()
    expect_c = NO.layer_norm(c.astype(np.float64) * NO.sigmoid(1.0), np.ones(d), np.zeros(d))
    assert rel_err(c_out.cpu().numpy(), expect_c) < 2e-6
    assert rel_err(h_out.cpu().numpy(), np.maximum(expect_c, 0) * 0.5) < 2e-6


@pytest.mark.parametrize("d", [32, 64, 128])
def test_einit_tile_rowdot(cuda_device, d):
    rng = np.random.RandomState(d)
    M = 777
    WC = rng.rand(M, 2).astype(np.float32)
    dims = [2, d // 8, d // 4, d // 2, d]
    layers, flat = [], []
    for a, b_ in zip(dims[:-1], dims[1:]):
        W = rng.randn(a, b_).astype(np.float32); b = (0.1 * rng.randn(b_)).astype(np.float32)
        layers.append((W.astype(np.float64), b.astype(np.float64))); flat += [W.reshape(-1), b]
    E0 = torch.empty((M, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_einit_fwd_f32", _lib.ptr(dev(WC, cuda_device)), _lib.ptr(dev(np.concatenate(flat), cuda_device)),
              _lib.ptr(E0), M, d, None)
    ref = NO.mlp(WC.astype(np.float64), layers, [True, True, True, False])
    torch.cuda.synchronize()
    assert rel_err(E0.cpu().numpy(), ref) < F32_TOL
    v = rng.randn(d).astype(np.float32)
    Y = torch.empty((50, d), dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_tile_rows_f32", _lib.ptr(dev(v, cuda_device)), 1.0 / np.sqrt(d), _lib.ptr(Y), 50, d, None)
    w = rng.randn(d).astype(np.float32); b = np.array([0.3], dtype=np.float32)
    y = torch.empty(M, dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_rowdot_f32", _lib.ptr(E0), _lib.ptr(dev(w, cuda_device)), _lib.ptr(dev(b, cuda_device)), _lib.ptr(y), M, d, None)
    torch.cuda.synchronize()
    assert rel_err(Y.cpu().numpy(), np.tile(v.astype(np.float64) / np.sqrt(np.float32(d)), (50, 1))) < 1e-6
    assert rel_err(y.cpu().numpy(), E0.cpu().numpy().astype(np.float64) @ w.astype(np.float64) + 0.3) < F32_TOL


def test_segment_mean_and_bce_metrics(cuda_device):
    rng = np.random.RandomState(3)
    n_edges = np.array([3, 780, 21, 1, 19900, 190])
    seg = np.concatenate([[0], np.cumsum(n_edges)]).astype(np.int32)
    vote = rng.randn(seg[-1]).astype(np.float32)
    B = len(n_edges)
    logits = torch.empty(B, dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_segment_mean_f32", _lib.ptr(dev(vote, cuda_device)), _lib.ptr(dev(seg, cuda_device, np.int32)),
              _lib.ptr(logits), B, None)
    torch.cuda.synchronize()
    ref = np.array([vote[seg[i]:seg[i + 1]].astype(np.float64).mean() for i in range(B)])
    assert np.abs(logits.cpu().numpy() - ref).max() < 1e-6
    # metrics incl. the half-to-even case sigmoid(0)=0.5 -> round -> 0
    lg = np.array([0.0, 2.0, -1.5, 0.3, -0.2, 0.0], dtype=np.float32)
    lab = np.array([0, 1, 0, 0, 1, 1], dtype=np.float32)
    pred = torch.empty(B, dtype=torch.float32, device=cuda_device); stats = torch.empty(6, dtype=torch.float32, device=cuda_device)
    _lib.call("tspgnn_bce_metrics_f32", _lib.ptr(dev(lg, cuda_device)), _lib.ptr(dev(lab, cuda_device)), _lib.ptr(pred),
              _lib.ptr(stats), B, None)
    torch.cuda.synchronize()
    p = NO.sigmoid(lg.astype(np.float64)); rp = np.round(p); eq = (lab == rp).astype(np.float64)
    loss = (np.maximum(lg, 0) - lg * lab + np.log1p(np.exp(-np.abs(lg)))).mean()
    expect = [loss, eq.mean(), (lab * eq).sum(), (lab * (1 - eq)).sum(), ((1 - lab) * eq).sum(), ((1 - lab) * (1 - eq)).sum()]
    assert np.allclose(stats.cpu().numpy(), expect, rtol=1e-6, atol=1e-7)
    assert rel_err(pred.cpu().numpy(), p) < 1e-6


def test_spmm_properties_at_full_c2_size(cuda_device):
    """C2 (n=40, B=128, d=64): degree check, adjoint identity, block independence."""
    EV, *_ = synthetic_batch([40] * 128, seed=1234)
    M, N = EV.shape
    assert (M, N) == (99840, 5120)
    rowptr, eid = EV.csr_by_vertex()
    uv = dev(EV.uv, cuda_device, np.int32); rp = dev(rowptr, cuda_device, np.int32); ei = dev(eid, cuda_device, np.int32)
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    X = torch.randn((N, 64), generator=g).to(cuda_device); Z = torch.randn((M, 64), generator=g).to(cuda_device)
    Y = torch.empty((M, 64), device=cuda_device); out = torch.empty((N, 64), device=cuda_device)
    _lib.call("tspgnn_gather2_sum_f32", _lib.ptr(uv), _lib.ptr(X), _lib.ptr(Y), M, N, 64, None)
    _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(rp), _lib.ptr(ei), _lib.ptr(Z), _lib.ptr(out), N, M, 64, None)
    ones = torch.ones((M, 64), device=cuda_device); deg = torch.empty((N, 64), device=cuda_device)
    _lib.call("tspgnn_csr_rowsum_f32", _lib.ptr(rp), _lib.ptr(ei), _lib.ptr(ones), _lib.ptr(deg), N, M, 64, None)
    torch.cuda.synchronize()
    assert torch.all(deg == 39.0)
    lhs = (Y.double() * Z.double()).sum().item(); rhs = (X.double() * out.double()).sum().item()
    assert abs(lhs - rhs) < 1e-6 * abs(lhs) + 1e-3
    # torch index ops as a second opinion at full size (sum order differs -> tolerance)
    ref = torch.zeros((N, 64), dtype=torch.float64, device=cuda_device)
    ref.index_add_(0, uv[:, 0].long(), Z.double()); ref.index_add_(0, uv[:, 1].long(), Z.double())
    assert (out.double() - ref).abs().max().item() < 1e-4
    assert torch.equal(Y, X[uv[:, 0].long()] + X[uv[:, 1].long()])


@pytest.mark.parametrize("n", [0, 5, 115529])
@pytest.mark.parametrize("with_grad", [True, False])
def test_bucket_pack_unpack(cuda_device, n, with_grad):
    """tspgnn_bucket_pack_f32 / _unpack_f32 against the tensor arithmetic of Session._pack_bucket / _unpack_bucket (the
    CPU plumbing path of the gloo tests), with a two-rank sum done by hand in between."""
    rng = np.random.RandomState(n + 3)
    g = [rng.randn(n).astype(np.float32) for _ in range(2)]
    stats = [np.array([0.7, 0.5, 3, 1, 2, 0], np.float32), np.array([0.4, 0.75, 1, 0, 5, 2], np.float32)]
    nb, flags = [6.0, 10.0], [4, 2]      # guard words: rank 0 "my variables were assigned" (bit 2), rank 1 "variance floor" (bit 1)
    buckets = []
    for r in range(2):
        b = dev(np.concatenate([g[r], np.full(10, 123.0, np.float32)]), cuda_device)
        fl = dev(np.array([flags[r]], np.uint32), cuda_device, dtype=np.uint32)
        _lib.call("tspgnn_bucket_pack_f32", _lib.ptr(b), n, int(with_grad), nb[r], _lib.ptr(dev(stats[r], cuda_device)),
                  _lib.ptr(fl), None)
        got = b.cpu().numpy()
        want_tail = np.array([nb[r], nb[r] * stats[r][0], nb[r] * stats[r][1], *stats[r][2:6],
                              flags[r] & 1, (flags[r] >> 1) & 1, (flags[r] >> 2) & 1], np.float32)   # one slot per guard bit
        assert np.array_equal(got[n:], want_tail)
        assert np.array_equal(got[:n], g[r] * np.float32(nb[r]) if with_grad else g[r])
        buckets.append(b)
    total = buckets[0] + buckets[1]
    _KEEP.append(total)
    st_out = torch.full((6,), -1.0, device=cuda_device)
    fl_out = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    _KEEP.extend([st_out, fl_out])
    before = total.cpu().numpy().copy()
    _lib.call("tspgnn_bucket_unpack_f32", _lib.ptr(total), n, int(with_grad), _lib.ptr(st_out), _lib.ptr(fl_out), None)
    got = total.cpu().numpy()
    inv = np.float32(1.0) / np.float32(16.0)
    assert np.array_equal(got[:n], before[:n] * inv if with_grad else before[:n])
    want_stats = np.concatenate([before[n + 1:n + 3] * inv, before[n + 3:n + 7]])
    assert np.array_equal(st_out.cpu().numpy(), want_stats)
    assert int(fl_out.item()) == 6      # the bits keep their meaning through the sum (they used to collapse to 1)
    # no statistics, no flag: nothing is dereferenced
    _lib.call("tspgnn_bucket_pack_f32", _lib.ptr(total), n, 0, 4.0, None, None, None)
    _lib.call("tspgnn_bucket_unpack_f32", _lib.ptr(total), n, 0, None, None, None)
    assert float(total[n].item()) == 4.0
