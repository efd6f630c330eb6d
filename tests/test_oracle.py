"""The CPU oracle checked against itself (two independent restatements), against analytic
known answers, against the golden regression vectors, and through metamorphic properties
(SURVEY.md §4 items 1-4).  The reference holds no vectors for this path -> "parity unpinned"."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_pack, rel_err
from oracle import np_oracle as NO
from oracle import params as P
from oracle import torch_oracle as TO


def pack_batch(name, seed=0):
    g = load_pack(name, seed)
    return {k: g[k] for k in ("ev_uv", "W", "C", "route_exists", "n_vertices", "n_edges")}


def test_param_inventory():
    assert P.n_params(64) == 115529 and P.n_params(128) == 457617
    shapes = P.param_shapes(64)
    assert shapes["E_init_MLP_MLP_layer_1/kernel"] == (2, 8)         # int(d/8) quirk, model.py:34
    assert shapes["TSP/E_cell/layer_norm_basic_lstm_cell/kernel"] == (128, 256)
    assert shapes["E_vote_MLP_layer_4/kernel"] == (64, 1)


@pytest.mark.parametrize("name,d,T", [("n5_B2", 32, 3), ("ragged_B6", 64, 2), ("sparse_B4", 32, 5)])
def test_numpy_and_torch_restatements_agree(name, d, T):
    batch = pack_batch(name)
    params = P.init_params(d, seed=3, perturb=True)
    a = NO.forward(params, batch, T, dtype=np.float64)
    b = TO.forward(TO.to_torch(params), batch, T)
    assert rel_err(a["predictions"], b["predictions"].numpy()) < 1e-12
    assert rel_err(a["E"][0], b["last_states"]["E"][0].numpy()) < 1e-11
    assert rel_err(a["V"][1], b["last_states"]["V"][1].numpy()) < 1e-11
    assert abs(a["loss"] - b["loss"].item()) < 1e-12
    for k in ("TP", "FP", "TN", "FN", "acc"):
        assert float(a[k]) == float(b[k])


def test_dense_variant_equals_index_variant():
    batch = pack_batch("ragged_B6", 2)
    params = TO.to_torch(P.init_params(32, seed=5, perturb=True))
    a = TO.forward(params, batch, 3, dense=True)
    b = TO.forward(params, batch, 3, dense=False)
    assert rel_err(a["logits"].numpy(), b["logits"].numpy()) < 1e-12
    assert rel_err(a["last_states"]["E"][0].numpy(), b["last_states"]["E"][0].numpy()) < 1e-12


@pytest.mark.parametrize("fname", ["oracle_n5_B2_d32_T3.npz", "oracle_ragged_B6_d64_T4.npz"])
def test_golden_regression(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    name = fname.split("_d")[0].replace("oracle_", "")
    d, T = int(z["d"]), int(z["T"])
    batch = pack_batch(name)
    params = P.init_params(d, seed=int(z["param_seed"]), perturb=True)
    out, grads = TO.loss_and_grads(params, batch, T)
    assert rel_err(out["predictions"].detach().numpy(), z["predictions"]) < 1e-12
    assert abs(out["loss"].item() - float(z["loss"])) < 1e-13
    assert rel_err(out["last_states"]["V"][0].detach().numpy(), z["Vh"]) < 1e-11
    gn = np.array([np.sqrt((g ** 2).sum()) for g in grads.values()])
    assert rel_err(gn, z["grad_norms"]) < 1e-9


# ----------------------------------------------------------------- analytic known answers
def test_kat_aggregation():
    batch = pack_batch("ragged_B6")
    uv, N = batch["ev_uv"].astype(np.int64), int(batch["n_vertices"].sum())
    X = np.random.RandomState(0).randn(N, 8)
    Y = NO.gather2_sum(uv, X)
    assert np.array_equal(Y[5], X[uv[5, 0]] + X[uv[5, 1]])
    ones = np.ones((uv.shape[0], 4))
    deg = NO.rowsum_by_vertex(uv, ones, N)[:, 0]
    offs = np.concatenate([[0], np.cumsum(batch["n_vertices"])])
    for i, n in enumerate(batch["n_vertices"]):
        assert np.all(deg[offs[i]:offs[i + 1]] == n - 1)          # EV^T 1 = degree = n-1
    rowptr, eid = NO.csr_by_vertex(uv, N)
    Z = np.random.RandomState(1).randn(uv.shape[0], 8)
    assert np.allclose(NO.csr_rowsum(rowptr, eid, Z), NO.rowsum_by_vertex(uv, Z, N), atol=1e-13)
    # adjoint identity <EV x, y> = <x, EV^T y>
    assert abs((NO.gather2_sum(uv, X) * Z).sum() - (X * NO.rowsum_by_vertex(uv, Z, N)).sum()) < 1e-9


def test_kat_layer_norm_of_ramp():
    x = np.arange(8, dtype=np.float64)[None, :] * 3.0 + 5.0
    y = NO.layer_norm(x, np.ones(8), np.zeros(8))
    expect = (np.arange(8) - 3.5) / np.sqrt(5.25)
    assert np.allclose(y[0], expect, atol=1e-10)
    y2 = NO.layer_norm(x, 2 * np.ones(8), np.ones(8))
    assert np.allclose(y2[0], 2 * expect + 1, atol=1e-10)


def test_kat_lstm_with_zero_kernel():
    """K=0, gamma=1, beta=0: every gate LN sees a constant row -> (x-mean)*inv = 0 exactly, so
    i=j=f=o=0, c' = LN(c*sigmoid(1) + sigmoid(0)*relu(0)) = LN(c*sigmoid(1)), h' = relu(c')/2."""
    d, rows = 8, 5
    rng = np.random.RandomState(0)
    c = rng.randn(rows, d); h = rng.randn(rows, d); x = rng.randn(rows, d)
    ln = {g: (np.ones(d), np.zeros(d)) for g in ("input", "transform", "forget", "output", "state")}
    nh, nc = NO.lnlstm(x, h, c, np.zeros((2 * d, 4 * d)), ln)
    expect_c = NO.layer_norm(c * NO.sigmoid(1.0), np.ones(d), np.zeros(d))
    assert np.allclose(nc, expect_c, atol=1e-12) and np.allclose(nh, np.maximum(expect_c, 0) * 0.5, atol=1e-12)


def test_zero_time_steps_returns_initial_states():
    batch = pack_batch("n5_B2")
    params = TO.to_torch(P.init_params(32, seed=1))
    out = TO.forward(params, batch, 0)
    Vh, Vc = out["last_states"]["V"]
    assert torch.equal(Vc, torch.zeros_like(Vc))
    assert torch.allclose(Vh, (params["V_init"] / np.sqrt(32.0)).repeat(Vh.shape[0], 1))


# ----------------------------------------------------------------- metamorphic properties
def test_block_diagonal_independence():
    """A batch of B graphs == B separate runs (EV is block diagonal, instance_loader.py:56-66)."""
    g = load_pack("ragged_B6", 1)
    params = TO.to_torch(P.init_params(32, seed=2, perturb=True))
    whole = TO.forward(params, {k: g[k] for k in ("ev_uv", "W", "C", "route_exists", "n_vertices", "n_edges")}, 3)
    eo = np.concatenate([[0], np.cumsum(g["n_edges"])]); vo = np.concatenate([[0], np.cumsum(g["n_vertices"])])
    for i in range(len(g["n_edges"])):
        sub = {"ev_uv": g["ev_uv"][eo[i]:eo[i + 1]] - vo[i], "W": g["W"][eo[i]:eo[i + 1]],
               "C": g["C"][eo[i]:eo[i + 1]], "route_exists": g["route_exists"][i:i + 1],
               "n_vertices": g["n_vertices"][i:i + 1], "n_edges": g["n_edges"][i:i + 1]}
        one = TO.forward(params, sub, 3)
        assert abs(one["logits"][0].item() - whole["logits"][i].item()) < 1e-12


def test_edge_permutation_equivariance():
    g = load_pack("n5_B2", 2)
    params = TO.to_torch(P.init_params(32, seed=4, perturb=True))
    batch = {k: g[k] for k in ("ev_uv", "W", "C", "route_exists", "n_vertices", "n_edges")}
    base = TO.forward(params, batch, 3)
    m0 = int(g["n_edges"][0])
    perm = np.concatenate([np.random.RandomState(0).permutation(m0), np.arange(m0, g["ev_uv"].shape[0])])
    pb = dict(batch, ev_uv=g["ev_uv"][perm], W=g["W"][perm], C=g["C"][perm])
    out = TO.forward(params, pb, 3)
    assert rel_err(out["logits"].numpy(), base["logits"].numpy()) < 1e-12
    assert rel_err(out["last_states"]["E"][0].numpy(), base["last_states"]["E"][0].numpy()[perm]) < 1e-12


def test_fp32_restatement_error_budget():
    """fp32 op-for-op restatement vs the fp64 oracle: the size of the error any fp32
    implementation of this recurrence carries (the budget the HIP path is held to)."""
    batch = pack_batch("ragged_B6")
    p = P.init_params(64, seed=0)
    ref = TO.forward(TO.to_torch(p, torch.float64), batch, 8)
    f32 = TO.forward(TO.to_torch(p, torch.float32), batch, 8, dense=True)
    assert rel_err(f32["predictions"].numpy(), ref["predictions"].numpy()) < 1e-5
    assert rel_err(f32["last_states"]["E"][0].numpy(), ref["last_states"]["E"][0].numpy()) < 1e-4


def test_clip_and_adam_formulas():
    g = {"a": np.array([3.0, 4.0])}
    clipped, gn = TO.clip_by_global_norm(g, 0.65)
    assert gn == 5.0 and np.allclose(clipped["a"], g["a"] * 0.65 / 5.0)
    small, _ = TO.clip_by_global_norm({"a": np.array([0.3, 0.4])}, 0.65)
    assert np.allclose(small["a"], [0.3, 0.4])                        # below the threshold: unchanged
    p, m, v = TO.adam_step({"a": np.zeros(2)}, {"a": np.array([1.0, -2.0])}, {"a": np.zeros(2)}, {"a": np.zeros(2)}, 1)
    # first Adam step moves by ~lr*sign(g)
    assert np.allclose(p["a"], [-2e-5, 2e-5], rtol=1e-6)


def test_lnlstm_cell_reproduces_tensorflow_unit_test_constants():
    """The one externally published vector for tf.contrib.rnn.LayerNormBasicLSTMCell: TensorFlow 1.x's own unit test
    (tensorflow/contrib/rnn/python/kernel_tests/rnn_cell_test.py, LayerNormBasicLSTMCellTest.testBasicLSTMCell) runs
    two stacked 2-unit cells (default tanh activation) with every kernel entry 0.5 (scope initialiser
    constant_initializer(0.5); gamma = 1, beta = 0 keep their own initialisers), x = [[1, 1]],
    (c0, h0, c1, h1) = 0.1 * ([0, 1], [2, 3], [4, 5], [6, 7]) and expects
        h = state_h = [[-0.38079708, 0.38079708]],  state_c = [[-1, 1]]   for BOTH layers.
    With a constant kernel the two units of every gate get the same pre-activation, so all four gate LayerNorms give
    0, c' = LN(c * sigmoid(0 + forget_bias 1.0)) = -/+1 and h' = tanh(-/+1) * sigmoid(0).  Both oracle cells (torch
    and NumPy restatement) must reproduce the constants; it pins the gate order's irrelevance here but, more to the
    point, the forget bias placement, LN-of-the-state-before-the-output, and the epsilon."""
    want_h = np.array([[-0.38079708, 0.38079708]])
    want_c = np.array([[-1.0, 1.0]])
    ones, zeros = np.ones(2), np.zeros(2)
    base = "root/cell"
    params = {base + "/kernel": np.full((4, 8), 0.5)}
    for g in ("input", "transform", "forget", "output", "state"):
        params[base + "/%s/gamma" % g], params[base + "/%s/beta" % g] = ones, zeros
    tp = TO.to_torch(params, torch.float64)
    ln = {g: (ones, zeros) for g in ("input", "transform", "forget", "output", "state")}
    x = np.array([[1.0, 1.0]])
    states = [(0.1 * np.array([[0.0, 1.0]]), 0.1 * np.array([[2.0, 3.0]])),
              (0.1 * np.array([[4.0, 5.0]]), 0.1 * np.array([[6.0, 7.0]]))]
    inp_t, inp_n = torch.tensor(x), x
    for c, h in states:                           # MultiRNNCell: layer k feeds its h to layer k+1
        h_t, c_t = TO.lnlstm_cell(inp_t, torch.tensor(h), torch.tensor(c), tp, None, activation=torch.tanh, base=base)
        h_n, c_n = NO.lnlstm(inp_n, h, c, params[base + "/kernel"], ln, activation=np.tanh)
        for got_h, got_c in ((h_t.numpy(), c_t.numpy()), (h_n, c_n)):
            assert np.abs(got_h - want_h).max() < 5e-9 and np.abs(got_c - want_c).max() < 5e-9
        inp_t, inp_n = h_t, h_n


def test_tensorflow_published_constants_for_clip_round_and_l2():
    """More of the little that TensorFlow 1.x itself publishes as numbers for ops on this path (the reference pins none):
    * tf.clip_by_global_norm -- clip_ops_test.ClipTest.testClipByGlobalNormClipped: x0 = [[-2, 0, 0], [4, 0, 0]],
      x1 = [1, -2], clip_norm 4 -> global norm 5, answers [[-1.6, 0, 0], [3.2, 0, 0]] and [0.8, -1.6];
    * tf.round -- its docstring: [0.9, 2.5, 2.3, 1.5, -4.5] -> [1, 2, 2, 2, -4] (half to even), which the oracle's metrics
      (model.py:150-154) take from torch.round and the device from rintf;
    * tf.nn.l2_loss -- nn_test.L2LossTest.testL2Loss: [1, 0, 3, 2] -> 7 (sum of squares / 2, no square root), the form
      the oracle's vars_cost (model.py:163) is written in.
    They narrow the 'parity unpinned' gap for model.py:150-167; they do not close it."""
    g = {"x0": np.array([[-2.0, 0.0, 0.0], [4.0, 0.0, 0.0]]), "x1": np.array([1.0, -2.0])}
    clipped, gn = TO.clip_by_global_norm(g, 4.0)
    assert gn == 5.0
    assert np.allclose(clipped["x0"], [[-1.6, 0.0, 0.0], [3.2, 0.0, 0.0]], rtol=0, atol=1e-15)
    assert np.allclose(clipped["x1"], [0.8, -1.6], rtol=0, atol=1e-15)
    x = torch.tensor([0.9, 2.5, 2.3, 1.5, -4.5], dtype=torch.float64)
    assert torch.round(x).tolist() == [1.0, 2.0, 2.0, 2.0, -4.0]
    assert np.rint(x.numpy()).tolist() == [1.0, 2.0, 2.0, 2.0, -4.0]
    # the oracle's vars_cost term, evaluated as loss_and_grads does: d/dp (sum p^2 / 2) = p, and the value is 7
    p = torch.tensor([1.0, 0.0, 3.0, 2.0], dtype=torch.float64, requires_grad=True)
    cost = (p ** 2).sum() / 2
    assert float(cost) == 7.0
    (grad,) = torch.autograd.grad(cost, [p])
    assert grad.tolist() == [1.0, 0.0, 3.0, 2.0]


def test_layer_norm_reproduces_tensorflow_layers_test():
    """tf.contrib.layers.layer_norm as TensorFlow 1.x's own test states it (tensorflow/contrib/layers/python/layers/
    layers_test.py, LayerNormTest.doOutputTest -- testOutput2DInput runs it on shape (10, 300)): for inputs
    randn * sigma + mu with mu in (0, 1e2), sigma in (1, 0.1), gamma = 1, beta = 0 (the variables' own initialisers),
    normalising over axis 1, the output must have mean 0 and variance 1 per row, and equal
        gamma * (x - mean) / sqrt(1e-12 + var) + beta        -- BIASED variance, epsilon 1e-12 inside the root
    to the test's tolerance (1e-5 at these shapes; the TensorFlow source is not in this image -- the set-up is quoted from
    memory of it, the numbers follow from the formula).  That is the function LayerNormBasicLSTMCell._norm calls (graphnn.py:168-170 through
    tf.contrib.rnn) -- the epsilon and the biased variance are what the HIP kernels' ln_gate restates (csrc/mfma_tile.h)."""
    rng = np.random.RandomState(0)
    for mu in (0.0, 1e2):
        for sigma in (1.0, 0.1):
            x = rng.randn(10, 300) * sigma + mu
            want = (x - x.mean(1, keepdims=True)) / np.sqrt(1e-12 + x.var(1, keepdims=True))
            got_t = TO.layer_norm(torch.tensor(x), torch.ones(300, dtype=torch.float64), torch.zeros(300, dtype=torch.float64)).numpy()
            got_n = NO.layer_norm(x, np.ones(300), np.zeros(300))
            for got in (got_t, got_n):
                assert np.abs(got.mean(1)).max() < 1e-5 and np.abs(got.var(1) - 1.0).max() < 1e-5
                assert np.abs(got - want).max() < 1e-5
    # a constant row: variance 0, and the 1e-12 INSIDE the root decides -- 0 * 1e6 = 0, not NaN (TF: rsqrt(var + eps))
    flat = np.full((2, 8), 3.0)
    assert np.array_equal(TO.layer_norm(torch.tensor(flat), torch.ones(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64)).numpy(),
                          np.zeros((2, 8)))


def test_adam_reproduces_tensorflow_adam_test_basic():
    """tf.train.AdamOptimizer as TensorFlow 1.x's own test states it (tensorflow/python/training/adam_test.py,
    AdamOptimizerTest.testBasic): var0 = [1, 2], var1 = [3, 4], CONSTANT gradients g0 = [0.1, 0.1], g1 = [0.01, 0.01],
    default hyper-parameters (lr 1e-3, beta1 0.9, beta2 0.999, epsilon 1e-8), three steps, expected values from the test's
    own adam_update_numpy:  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t),  m, v exponential averages,
    p -= lr_t * m / (sqrt(v) + eps)  -- epsilon OUTSIDE the root and un-corrected ("epsilon hat").  Written out here
    independently of oracle.adam_step (for a constant gradient m_t = (1 - beta1^t) g and v_t = (1 - beta2^t) g^2 in closed
    form; the TensorFlow source is not in this image, the set-up is quoted from memory of it).  The reference's step
    (model.py:160-167) uses lr = 2e-5; the rule is the same."""
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    p = {"var0": np.array([1.0, 2.0]), "var1": np.array([3.0, 4.0])}
    g = {"var0": np.array([0.1, 0.1]), "var1": np.array([0.01, 0.01])}
    m = {k: np.zeros(2) for k in p}
    v = {k: np.zeros(2) for k in p}
    want = {k: a.copy() for k, a in p.items()}
    for t in range(1, 4):
        p, m, v = TO.adam_step(p, g, m, v, t, lr=lr)
        for k in want:       # the closed form of adam_update_numpy under a constant gradient
            lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            want[k] = want[k] - lr_t * ((1 - b1 ** t) * g[k]) / (np.sqrt((1 - b2 ** t) * g[k] ** 2) + eps)
            assert np.allclose(p[k], want[k], rtol=0, atol=1e-15), (t, k)
            assert np.allclose(m[k], (1 - b1 ** t) * g[k], atol=1e-17) and np.allclose(v[k], (1 - b2 ** t) * g[k] ** 2, atol=1e-19)
    # three steps of ~lr each: the numbers the TF test arrives at (to the 1e-6 of its assertAllCloseAccordingToType)
    assert np.allclose(p["var0"], [0.997, 1.997], atol=1e-6) and np.allclose(p["var1"], [2.997, 3.997], atol=1e-6)
    # and the first step is NOT lr * g: Adam's first move is lr * sign(g) (up to epsilon) whatever the gradient's size
    assert abs((1.0 - 0.997) / 3 - lr) < 1e-7


def test_sigmoid_cross_entropy_follows_tensorflows_documented_form():
    """tf.nn.sigmoid_cross_entropy_with_logits as TensorFlow's API page defines it -- z * -log(sigmoid(x)) + (1 - z) *
    -log(1 - sigmoid(x)), evaluated as max(x, 0) - x z + log(1 + exp(-|x|)) "to ensure stability and avoid overflow" -- on
    closed-form values (x = 0: log 2 whatever the label; x = 1, z = 1: log(1 + 1/e)) and where the naive form breaks
    (|x| = 50, 800); the loss the HIP path is compared with (model.py:147) goes through this function."""
    import math
    f = TO.sigmoid_cross_entropy_with_logits
    t = lambda *v: torch.tensor(v, dtype=torch.float64)
    got = f(t(0.0, 0.0, 1.0, -2.0, 2.0), t(0.0, 1.0, 1.0, 1.0, 0.0)).numpy()
    want = [math.log(2.0), math.log(2.0), math.log1p(math.exp(-1.0)), 2.0 + math.log1p(math.exp(-2.0)), 2.0 + math.log1p(math.exp(-2.0))]
    assert np.abs(got - np.array(want)).max() < 1e-15
    assert abs(want[2] - 0.31326168751822286) < 1e-15 and abs(want[3] - 2.1269280110429727) < 1e-15
    x = t(-50.0, 50.0, -800.0, 800.0, 3.7, -0.4)
    for z in (0.0, 1.0):
        zz = torch.full_like(x, z)
        sig = torch.sigmoid(x)
        with np.errstate(divide="ignore"):
            naive = zz * -torch.log(sig) + (1 - zz) * -torch.log1p(-sig)
        got = f(x, zz)
        assert torch.isfinite(got).all()
        ok = torch.isfinite(naive) & (x.abs() < 30)
        assert (got[ok] - naive[ok]).abs().max() < 1e-12
        # the saturated ends: the loss is |x| on the wrong side of the label, ~0 on the right side
        assert abs(float(got[2]) - (800.0 if z == 1.0 else 0.0)) < 1e-12 and abs(float(got[3]) - (0.0 if z == 1.0 else 800.0)) < 1e-12
