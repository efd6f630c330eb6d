"""End-to-end parity of the HIP path (build_network / Session / GraphNN) with the CPU oracle on
identical instances and identical weights.  Tolerance: 1e-5 relative on predictions / last
states (BASELINE.json north_star), measured against the float64 oracle; the fp32 op-for-op
restatement is shown next to it as the error any fp32 implementation carries."""
import numpy as np
import pytest
import torch

import tspgnn
from conftest import batch_from_tuple, load_pack, rel_err
from oracle import params as P
from oracle import torch_oracle as TO

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
MOVE_TOL_BY_CLASS = {}   # (filled from the round-5 measurement below)
MOVE_TOL = 0.05    # three Adam steps: error of a variable's movement relative to its largest movement (see the test;
                   # measured 3.9e-2 in round 4 -- entries whose gradient sits at fp32 noise level, normalised by sqrt(v))


def run_hip(d, params, batch_tuple, T, fetch=("loss", "acc", "predictions", "TP", "FP", "TN", "FN", "last_states"),
            gemm=None):
    model = tspgnn.build_network(d)
    if gemm is not None:
        model["gnn"].gemm = gemm
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = batch_tuple
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    vals = sess.run([model[k] for k in fetch], feed_dict=feed)
    return dict(zip(fetch, vals))


def pack_tuple(name, seed=0, dense=False):
    g = load_pack(name, seed)
    ev = tspgnn.SparseEV(g["ev_uv"], int(g["ev_shape"][1]))
    return (ev.todense() if dense else ev, g["W"], g["C"], g["route_exists"], g["n_vertices"], g["n_edges"])


@pytest.mark.parametrize("name,d,T", [("n5_B2", 32, 3), ("ragged_B6", 64, 4), ("sparse_B4", 64, 8),
                                      ("n20_B32", 64, 8), ("target_B4", 32, 2), ("n20_B32", 128, 2)])
@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "f32"])
def test_forward_parity_with_oracle(cuda_device, name, d, T, gemm):
    """Every GEMM arithmetic of the inference forward (fp16 matrix cores on 2-way splits -- the default --, bf16
    matrix cores on exact 3-way splits, fp32 MFMA) meets the same 1e-5 budget; d=128 has no split-operand kernel
    and runs fp32 either way."""
    t = pack_tuple(name)
    params = P.init_params(d, seed=11, perturb=True)
    hip = run_hip(d, params, t, T, gemm=gemm)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    f32 = TO.forward(TO.to_torch(params, torch.float32), batch, T, dense=True)
    e_pred = rel_err(hip["predictions"], ref["predictions"].numpy())
    e_Eh = rel_err(hip["last_states"]["E"].h, ref["last_states"]["E"][0].numpy())
    e_Vc = rel_err(hip["last_states"]["V"].c, ref["last_states"]["V"][1].numpy())
    b_Eh = rel_err(f32["last_states"]["E"][0].numpy(), ref["last_states"]["E"][0].numpy())
    print("\n[%s d=%d T=%d %s] HIP vs f64: pred %.2e  E.h %.2e  V.c %.2e | fp32 restatement vs f64: E.h %.2e"
          % (name, d, T, gemm, e_pred, e_Eh, e_Vc, b_Eh))
    assert e_pred < REL_TOL and e_Eh < REL_TOL and e_Vc < REL_TOL
    assert abs(float(hip["loss"]) - ref["loss"].item()) < REL_TOL
    for k in ("acc", "TP", "FP", "TN", "FN"):
        assert float(hip[k]) == float(ref[k]), k


def test_centered_gate_kernels_change_nothing_but_rounding(cuda_device):
    """The f16x2 forward multiplies with cell kernels whose columns are centred per gate (and skips the mean pass of the
    gate LayerNorms: tspgnn_lstm_task.z_centered) -- LayerNorm subtracts that mean anyway.  Against the same forward on
    the plain kernels: equal to rounding, two orders below the 1e-5 parity budget; and against the float64 oracle the
    budget holds with weights that have large per-gate column means (a bias-like offset on every kernel row)."""
    t = pack_tuple("n20_B32")
    d, T = 64, 8
    params = P.init_params(d, seed=4, perturb=True)
    for name in list(params):
        if name.endswith("lstm_cell/kernel"):
            params[name] = (params[name] + 0.05).astype(params[name].dtype)   # an offset on every column: the mean LayerNorm removes
    outs = {}
    for centered in (True, False):
        model = tspgnn.build_network(d)
        model["gnn"].center_gates = centered
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        outs[centered] = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
    assert rel_err(outs[True][0], outs[False][0]) < 2e-6
    for var in ("E", "V"):
        assert rel_err(outs[True][1][var].h, outs[False][1][var].h) < 2e-6
        assert rel_err(outs[True][1][var].c, outs[False][1][var].c) < 2e-6
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    assert rel_err(outs[True][0], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(outs[True][1]["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert rel_err(outs[True][1]["V"].c, ref["last_states"]["V"][1].numpy()) < REL_TOL


@pytest.mark.parametrize("name,d,fold", [("ragged_B6", 64, False), ("sparse_B4", 64, True), ("n5_B2", 32, True)])
def test_centered_gate_kernels_in_every_plan_shape(cuda_device, name, d, fold):
    """The centred packings cover every way the fused f16x2 plan feeds a cell -- gather-init (Kh + the Kx behind Zx), the
    pushed last layer ([W Kx ; Kh] and b Kx), and with fold_adjacency off the plain [Kx ; Kh] kernel: parity with the
    float64 oracle on ragged / sparse / tiny batches."""
    t = pack_tuple(name)
    T = 4
    params = P.init_params(d, seed=8, perturb=True)
    model = tspgnn.build_network(d)
    model["gnn"].fold_adjacency = fold
    assert model["gnn"].center_gates
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    assert rel_err(pred, ref["predictions"].numpy()) < REL_TOL
    assert rel_err(last["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert rel_err(last["V"].h, ref["last_states"]["V"][0].numpy()) < REL_TOL


def test_reference_hyperparameters_c1(cuda_device):
    """BASELINE configs[0]: n=20, B=32, d=64, T=8 with the reference's initialisers."""
    t = pack_tuple("n20_B32", 2)
    params = P.init_params(64, seed=0)
    hip = run_hip(64, params, t, 8)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, 8)
    assert rel_err(hip["predictions"], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(hip["last_states"]["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL


def test_dense_feed_equals_sparse_feed(cuda_device):
    params = P.init_params(32, seed=3, perturb=True)
    a = run_hip(32, params, pack_tuple("ragged_B6", 1, dense=False), 3)
    b = run_hip(32, params, pack_tuple("ragged_B6", 1, dense=True), 3)
    assert np.array_equal(a["predictions"], b["predictions"])          # same kernels, same order: bit exact
    assert np.array_equal(a["last_states"]["E"].c, b["last_states"]["E"].c)


def test_zero_time_steps(cuda_device):
    params = P.init_params(32, seed=5)
    out = run_hip(32, params, pack_tuple("n5_B2"), 0)
    V = out["last_states"]["V"]
    assert np.all(V.c == 0)
    assert rel_err(V.h, np.tile(params["V_init"] / np.sqrt(32.0), (V.h.shape[0], 1))) < 1e-6


def test_block_diagonal_independence_on_device(cuda_device):
    """Each problem's prediction in a ragged batch equals the prediction of that problem alone."""
    g = load_pack("ragged_B6", 2)
    params = P.init_params(64, seed=8, perturb=True)
    whole = run_hip(64, params, pack_tuple("ragged_B6", 2), 4)
    eo = np.concatenate([[0], np.cumsum(g["n_edges"])]); vo = np.concatenate([[0], np.cumsum(g["n_vertices"])])
    for i in (0, 2, 5):
        ev = tspgnn.SparseEV(g["ev_uv"][eo[i]:eo[i + 1]] - vo[i], int(g["n_vertices"][i]))
        t = (ev, g["W"][eo[i]:eo[i + 1]], g["C"][eo[i]:eo[i + 1]], g["route_exists"][i:i + 1],
             g["n_vertices"][i:i + 1], g["n_edges"][i:i + 1])
        one = run_hip(64, params, t, 4)
        assert abs(float(one["predictions"][0]) - float(whole["predictions"][i])) < 2e-6


def test_generic_graphnn_wiring_two_inputs(cuda_device):
    """A GraphNN the TSP model does not use: two loop entries for one variable (identity entry +
    a valued, non-incidence matrix) -> concatenated cell input of width 2d (graphnn.py:142-173)."""
    from tspgnn import variables as V
    d = 32
    store = V.VariableStore()
    rng = np.random.RandomState(0)
    A = (rng.randn(9, 7) * (rng.rand(9, 7) < 0.5)).astype(np.float32)       # valued "M": U x W
    gnn = tspgnn.GraphNN({"U": d, "W": d}, {"M": ("U", "W")}, {"c": ("W", "U"), "b": ("U", "W")},
                         {"U": [{"var": "U"}, {"mat": "M", "msg": "c", "var": "W"}],
                          "W": [{"mat": "M", "transpose?": True, "msg": "b", "var": "U"}]}, name="G", store=store)
    store.finalize(cuda_device); store.initialize(seed=4)
    sd = {k: v.astype(np.float64) for k, v in store.state_dict().items()}
    U0 = rng.randn(9, d).astype(np.float32); W0 = rng.randn(7, d).astype(np.float32)
    out = gnn({"M": A}, {"U": torch.from_numpy(U0).to(cuda_device), "W": torch.from_numpy(W0).to(cuda_device)}, 3)
    torch.cuda.synchronize()
    # oracle of the same wiring from the NumPy helpers
    from oracle import np_oracle as NO
    def mlp(x, name):
        return NO.mlp(x, [(sd["G/%s_MLP_layer_%d/kernel" % (name, i)], sd["G/%s_MLP_layer_%d/bias" % (name, i)])
                          for i in range(1, 5)], [True, True, True, False])
    def cell(x, h, c, v):
        base = "G/%s_cell/layer_norm_basic_lstm_cell" % v
        ln = {g: (sd["%s/%s/gamma" % (base, g)], sd["%s/%s/beta" % (base, g)])
              for g in ("input", "transform", "forget", "output", "state")}
        return NO.lnlstm(x, h, c, sd[base + "/kernel"], ln)
    Uh, Uc, Wh, Wc = U0.astype(np.float64), np.zeros((9, d)), W0.astype(np.float64), np.zeros((7, d))
    A64 = A.astype(np.float64)
    for _ in range(3):
        xu = np.concatenate([Uh, A64 @ mlp(Wh, "c")], axis=1)
        xw = A64.T @ mlp(Uh, "b")
        (Uh, Uc), (Wh, Wc) = cell(xu, Uh, Uc, "U"), cell(xw, Wh, Wc, "W")
    assert rel_err(out["U"].h.cpu().numpy(), Uh) < REL_TOL and rel_err(out["W"].c.cpu().numpy(), Wc) < REL_TOL


class _Square(object):
    """A loop entry's 'fun' that brings its own vector-Jacobian product along (GraphNN._fun_vjp)."""

    def __call__(self, x):
        return x * x

    def vjp(self, h, g_out):
        return 2.0 * h * g_out


@pytest.mark.parametrize("gemm", ["f16x2", "f32"])
def test_generic_wiring_trains_through_fun_and_appended_matrix_entries(cuda_device, gemm):
    """The last two loop-entry kinds of graphnn.py:142-173 under tf.gradients (model.py:166): a Python 'fun' ahead of the
    message MLP (graphnn.py:149-151) -- one differentiated by autograd on the call, one with its own vjp -- and a matrix
    appended to the cell input as it is (graphnn.py:163-165; a constant: its columns only meet the cell kernel).  Forward
    states and EVERY variable's gradient (plus the gradient w.r.t. the initial embeddings) against float64 autograd over the
    oracle's op-for-op helpers on the same wiring, for a loss that is a fixed linear functional of the final states."""
    from tspgnn import variables as V
    d, T = 32, 3
    store = V.VariableStore()
    rng = np.random.RandomState(1)
    A = (rng.randn(9, 7) * (rng.rand(9, 7) < 0.6)).astype(np.float32)       # valued "M": U x W
    F = rng.randn(9, 32).astype(np.float32)                                  # appended to U's cell input as it is
    half_tanh = lambda x: 0.5 * torch.tanh(x)    # noqa: E731
    gnn = tspgnn.GraphNN({"U": d, "W": d}, {"M": ("U", "W"), "F": ("U", 32)}, {"c": ("W", "U"), "b": ("U", "W")},
                         {"U": [{"var": "U", "fun": half_tanh}, {"mat": "M", "msg": "c", "var": "W"}, {"mat": "F"}],
                          "W": [{"mat": "M", "transpose?": True, "msg": "b", "var": "U", "fun": _Square()}]},
                         name="G", store=store)
    gnn.gemm = gemm
    store.finalize(cuda_device); store.initialize(seed=6)
    sd = {k: v.astype(np.float64) for k, v in store.state_dict().items()}
    U0 = rng.randn(9, d).astype(np.float32); W0 = rng.randn(7, d).astype(np.float32)
    wU = rng.randn(9, d).astype(np.float32); wW = rng.randn(7, d).astype(np.float32)
    dev_t = lambda a: torch.from_numpy(a).to(cuda_device)    # noqa: E731
    store.zero_grad()
    states, tape = gnn.forward_train({"M": A, "F": F}, {"U": dev_t(U0), "W": dev_t(W0)}, T)
    dinit = gnn.backward(tape, {"U": (dev_t(wU), None), "W": (None, dev_t(wW))})
    torch.cuda.synchronize()
    got = store.grad_dict()

    # float64 autograd over the oracle's helpers, same wiring
    p = TO.to_torch(sd, torch.float64, requires_grad=True)
    A64, F64 = torch.tensor(A, dtype=torch.float64), torch.tensor(F, dtype=torch.float64)
    Uh = torch.tensor(U0, dtype=torch.float64, requires_grad=True)
    Wh = torch.tensor(W0, dtype=torch.float64, requires_grad=True)
    uh, wh, uc, wc = Uh, Wh, torch.zeros(9, d, dtype=torch.float64), torch.zeros(7, d, dtype=torch.float64)
    for _ in range(T):
        xu = torch.cat([0.5 * torch.tanh(uh), A64 @ TO.mlp(wh, p, "G/c"), F64], dim=1)
        xw = A64.T @ TO.mlp(uh * uh, p, "G/b")
        (uh, uc), (wh, wc) = (TO.lnlstm_cell(xu, uh, uc, p, None, base="G/U_cell/layer_norm_basic_lstm_cell"),
                              TO.lnlstm_cell(xw, wh, wc, p, None, base="G/W_cell/layer_norm_basic_lstm_cell"))
    assert rel_err(states["U"].h.cpu().numpy(), uh.detach().numpy()) < REL_TOL
    assert rel_err(states["W"].c.cpu().numpy(), wc.detach().numpy()) < REL_TOL
    loss = (uh * torch.tensor(wU, dtype=torch.float64)).sum() + (wc * torch.tensor(wW, dtype=torch.float64)).sum()
    names = list(p.keys())
    grads = torch.autograd.grad(loss, [p[k] for k in names] + [Uh, Wh], allow_unused=True)
    gscale = max(float(g.abs().max()) for g in grads if g is not None)
    for k, g in zip(names, grads[:len(names)]):
        ref = np.zeros_like(sd[k]) if g is None else g.numpy()
        scale = max(np.abs(ref).max(), 1e-3 * gscale)
        assert np.abs(got[k].astype(np.float64) - ref).max() / scale < 2e-5, k
    # the F columns of U's kernel saw a gradient (rows [2d, 2d + 32) of kernel[dx + d, 4d]), F itself none
    kU = got["G/U_cell/layer_norm_basic_lstm_cell/kernel"]
    assert np.abs(kU[2 * d:2 * d + 32]).max() > 1e-4 * gscale
    for v, g in (("U", grads[-2]), ("W", grads[-1])):
        assert rel_err(dinit[v][0].cpu().numpy(), g.numpy()) < 2e-5, v


def test_c2_full_size_properties(cuda_device):
    """BASELINE configs[1] (n=40, B=128, d=64, T=32) at full size: finite outputs, the paired
    instances (same graph, C*(1-/+dev)) differ, and shuffling the problems permutes predictions."""
    sizes = [40] * 128
    t = tspgnn.synthetic_batch(sizes, seed=1234)
    params = P.init_params(64, seed=0)
    out = run_hip(64, params, t, 32, fetch=("predictions", "loss"))
    assert np.all(np.isfinite(out["predictions"])) and np.isfinite(out["loss"])
    # problem-level permutation: reverse the order of the problems
    EV, W, C, r, nv, ne = t
    eo = np.concatenate([[0], np.cumsum(ne)]); vo = np.concatenate([[0], np.cumsum(nv)])
    order = np.arange(len(ne))[::-1]
    uv2, W2, C2 = [], [], []
    acc_v = 0
    for i in order:
        uv2.append(EV.uv[eo[i]:eo[i + 1]] - vo[i] + acc_v); W2.append(W[eo[i]:eo[i + 1]]); C2.append(C[eo[i]:eo[i + 1]])
        acc_v += nv[i]
    t2 = (tspgnn.SparseEV(np.concatenate(uv2), EV.shape[1]), np.concatenate(W2), np.concatenate(C2), r[order], nv[order], ne[order])
    out2 = run_hip(64, params, t2, 32, fetch=("predictions",))
    assert rel_err(out2["predictions"], out["predictions"][order]) < 2e-6


# ----------------------------------------------------------------------------- training path
def grads_hip(d, params, batch_tuple, T, float_dtype=torch.float32, chunk_bytes=None):
    model = tspgnn.build_network(d, float_dtype=float_dtype)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    if chunk_bytes is not None:
        model["gnn"].wgrad_chunk_bytes = chunk_bytes
    EV, W, C, route_exists, n_vertices, n_edges = batch_tuple
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    out = sess.loss_and_grads(feed)
    torch.cuda.synchronize()
    return model, sess, feed, out, model.store.grad_dict()


@pytest.mark.parametrize("name,d,T", [("n5_B2", 32, 3), ("ragged_B6", 64, 4), ("sparse_B4", 64, 6), ("n20_B32", 64, 8),
                                      ("ragged_B6", 128, 2), ("n5_B2", 32, 0)])
def test_gradient_parity_with_autograd_oracle(cuda_device, name, d, T):
    """tf.gradients(loss) restated by torch autograd on the float64 oracle vs the HIP backward."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=21, perturb=True)
    model, sess, feed, out, g = grads_hip(d, params, t, T)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref_out, ref_g = TO.loss_and_grads(params, batch, T, dtype=torch.float64)
    _, f32_g = TO.loss_and_grads(params, batch, T, dtype=torch.float32, dense=True)   # the fp32 error budget
    assert abs(float(out["stats"][0].item()) - ref_out["loss"].item()) < REL_TOL
    l2 = {k: TO.L2NORM_SCALING * params[k] for k in params}          # the oracle's grads include the L2 term
    worst, worst32 = 0.0, 0.0
    gscale = max(np.abs(ref_g[k] - l2[k]).max() for k in ref_g)
    for k in ref_g:
        ref = ref_g[k] - l2[k]
        # per-variable relative error, floored at 1e-3 of the largest gradient entry overall
        scale = max(np.abs(ref).max(), 1e-3 * gscale)
        err = np.abs(g[k] - ref).max() / scale
        err32 = np.abs(f32_g[k] - ref_g[k]).max() / scale
        worst, worst32 = max(worst, err), max(worst32, err32)
        # 1e-5 (the forward's own budget; measured <= 6e-6), or -- for sums with heavy cancellation -- twice what an
        # op-for-op fp32 autograd run loses itself
        assert err < max(1e-5, 2 * err32), (k, err, err32)
    print("\n[%s d=%d T=%d] worst per-variable gradient rel err: HIP %.2e, fp32 restatement %.2e"
          % (name, d, T, worst, worst32))


@pytest.mark.parametrize("name,d,T", [("ragged_B6", 64, 4), ("n5_B2", 32, 1), ("n20_B32", 64, 3)])
def test_training_forward_fused_messages_write_the_same_tape(cuda_device, name, d, T):
    """The training forward with the message MLPs inside the cell launches (f16x2 default) leaves the tape of the
    two-launch form: the same device functions run on the same operands, so every saved tensor is bit-identical."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=5, perturb=True)
    tapes = []
    for fuse in (True, False):   # (the default is the two-launch form)
        model = tspgnn.build_network(d)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        gnn = model["gnn"]
        gnn.fuse_training_messages = fuse
        gnn.push_training = False           # (the fused form keeps the whole message MLP on the edge rows)
        EV, W, C = t[0], t[1], t[2]
        dev = sess.device
        M, N = EV.shape
        E0 = torch.randn((M, d), generator=torch.Generator().manual_seed(3)).to(dev)
        V0 = torch.randn((N, d), generator=torch.Generator().manual_seed(4)).to(dev)
        states, tape = gnn.forward_train({"EV": EV}, {"V": V0, "E": E0}, T)
        torch.cuda.synchronize()
        assert tape.fused == fuse
        tapes.append(tape)
    a, b = tapes
    for name_ in ("H", "C", "X"):
        for v in getattr(a, name_):
            assert torch.equal(getattr(a, name_)[v], getattr(b, name_)[v]), (name_, v)

    def rows_of(zx, n):   # blocked projected messages -> row-major, without the padding rows of the last block
        steps, pad, w = zx.shape
        return zx.view(steps, pad // 16, w // 16, 4, 16, 4).permute(0, 1, 4, 2, 3, 5).reshape(steps, pad, w)[:, :n]
    for v in a.ZX:
        n_src = a.X[v].shape[1]
        assert torch.equal(rows_of(a.ZX[v], n_src), rows_of(b.ZX[v], n_src)), ("ZX", v)
    for key in a.acts:
        assert torch.equal(a.acts[key], b.acts[key]), key


def test_train_steps_follow_the_oracle(cuda_device):
    """Three sess.run(train_step) calls (L2 + clip-by-global-norm 0.65 + Adam lr 2e-5) vs the oracle's."""
    t = pack_tuple("ragged_B6", 0)
    d, T = 64, 4
    params = P.init_params(d, seed=2, perturb=True)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    batch = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
    p = {k: v.copy() for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v) for k, v in p.items()}
    for step in (1, 2, 3):
        vals = sess.run([model["train_step"], model["loss"], model["acc"], model["predictions"]], feed_dict=feed)
        ref_out, p, m, v, gn = TO.train_step(p, batch, T, m, v, step)
        assert vals[0] is None and abs(float(vals[1]) - ref_out["loss"].item()) < REL_TOL
        assert abs(float(sess._adam["gnorm"].item()) - gn) < 2e-5 * gn
    now = model.store.state_dict()
    worst = 0.0
    by_class = {}
    for k in p:
        # three Adam steps move every weight by ~6e-5; compare the MOVEMENT, not just the value
        moved_ref = p[k] - params[k]
        moved = now[k].astype(np.float64) - params[k]
        # (Adam normalises by sqrt(v): an entry whose gradient sits at the fp32 noise level moves by a
        # noise-dependent fraction of the step, hence a budget relative to the largest movement)
        ratio = np.abs(moved - moved_ref).max() / np.abs(moved_ref).max()
        worst = max(worst, ratio)
        klass = k.rsplit("/", 1)[-1] if "/" in k else k
        by_class[klass] = max(by_class.get(klass, 0.0), ratio)
        assert np.abs(moved - moved_ref).max() < 2e-7 + MOVE_TOL_BY_CLASS.get(klass, MOVE_TOL) * np.abs(moved_ref).max(), (k, ratio)
    print("\n[train steps] worst movement error relative to the largest movement of its variable: %.2e; by class: %s"
          % (worst, "  ".join("%s %.2e" % kv for kv in sorted(by_class.items()))))


def test_captured_graph_replay_matches_eager(cuda_device):
    t = pack_tuple("ragged_B6", 2)
    params = P.init_params(64, seed=4, perturb=True)
    model = tspgnn.build_network(64)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 5,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    eager = sess.forward_device(b)
    e_pred, e_h = eager["predictions"].clone(), eager["last_states"]["E"].h.clone()
    replay = sess.capture_forward(b)
    for _ in range(3):
        out = replay()
    torch.cuda.synchronize()
    assert torch.equal(out["predictions"], e_pred) and torch.equal(out["last_states"]["E"].h, e_h)
    model.store.load(params)                      # bumps the variable version
    with pytest.raises(RuntimeError, match="capture again"):
        replay()


def test_binary_search_caller(cuda_device):
    """experiments/binary_search.py get_cost on the HIP predictions: same probes and decisions as the loop
    driven by the float64 oracle (untrained weights, so only the mechanics are checked), and the k-ary
    variant brackets the same threshold crossing."""
    rng = np.random.RandomState(5)
    inst = tspgnn.random_instance(9, rng)
    d, T = 32, 3
    params = P.init_params(d, seed=12, perturb=True)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    wpred, pred, route_cost, iters = tspgnn.get_cost(sess, model, inst, T)
    # the same loop on the oracle
    Ma, Mw, route = inst
    from tspgnn.binary_search import cost_bounds
    wmin, wmax = cost_bounds(Mw, 9)
    EV, W, _, r, nv, ne = tspgnn.InstanceLoader.create_batch([inst], target_cost=0.0)
    tp = TO.to_torch(params, torch.float64)
    w, it = (wmin + wmax) / 2, 0
    while wmin < w * 0.99 or w * 1.01 < wmax:
        b = {"ev_uv": EV.uv, "W": W, "C": np.ones_like(W) * w, "route_exists": r, "n_vertices": nv, "n_edges": ne}
        p = TO.forward(tp, b, T)["predictions"].item()
        if p < 0.5:
            wmin = w
        else:
            wmax = w
        w, it = (wmin + wmax) / 2, it + 1
    assert it == iters and abs(w - wpred) < 1e-12 and 0 < iters < 40
    assert abs(route_cost - sum(Mw[min(i, j), max(i, j)] for i, j in zip(route, route[1:] + route[:1])) / 9) < 1e-12
    w8, _, _, it8 = tspgnn.get_cost(sess, model, inst, T, parallel=8)
    assert it8 <= (iters + 2) // 3 + 1


def test_c4_ragged_full_size(cuda_device):
    """BASELINE configs[3]: mixed n in {20..80}, batch 512, one GPU, CSR-packed block-diagonal adjacency
    (seed 0 -> N=25 362, M=695 849 per SURVEY.md §8d M2).  Full-size properties: sizes, finite outputs,
    per-problem prediction of three problems equal to running each alone (ragged segments)."""
    rng = np.random.RandomState(0)
    sizes = rng.randint(20, 81, size=512)
    t = tspgnn.synthetic_batch(sizes, seed=1234)
    EV, W, C, r, nv, ne = t
    assert EV.shape == (int((sizes * (sizes - 1) // 2).sum()), int(sizes.sum()))
    params = P.init_params(64, seed=0)
    out = run_hip(64, params, t, 8, fetch=("predictions", "loss"))
    assert np.all(np.isfinite(out["predictions"])) and np.isfinite(out["loss"])
    eo = np.concatenate([[0], np.cumsum(ne)]); vo = np.concatenate([[0], np.cumsum(nv)])
    for i in (0, 255, 511):
        ev = tspgnn.SparseEV(EV.uv[eo[i]:eo[i + 1]] - vo[i], int(nv[i]))
        one = run_hip(64, params, (ev, W[eo[i]:eo[i + 1]], C[eo[i]:eo[i + 1]], r[i:i + 1], nv[i:i + 1], ne[i:i + 1]), 8,
                      fetch=("predictions",))
        assert abs(float(one["predictions"][0]) - float(out["predictions"][i])) < 2e-6


def test_c5_shape_d128_n200(cuda_device):
    """BASELINE configs[4] shape at a reduced batch: n=200 (vertex degree 199 > one wavefront of edge ids),
    d=128 (LSTM kernel matrix streamed through LDS in chunks), fp32; parity against the oracle at T=2."""
    t = tspgnn.synthetic_batch([200, 200], seed=3)
    params = P.init_params(128, seed=6, perturb=True)
    hip = run_hip(128, params, t, 2, fetch=("predictions", "last_states"))
    ref = TO.forward(TO.to_torch(params, torch.float64), batch_from_tuple(t), 2)
    assert rel_err(hip["predictions"], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(hip["last_states"]["V"].h, ref["last_states"]["V"][0].numpy()) < REL_TOL


@pytest.mark.parametrize("float_dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16-storage"])
def test_captured_training_step_matches_eager(cuda_device, float_dtype):
    """capture_train_step (two HIP graphs around the all-reduce) must walk the same trajectory as eager
    train_step: identical weights after 3 steps, bit for bit (same kernels, same order) -- in the bf16-storage mode too,
    whose backward swaps a rounded copy of the variables in and out inside the captured region."""
    t = pack_tuple("ragged_B6", 1)
    params = P.init_params(64, seed=9, perturb=True)
    finals = []
    for captured in (False, True):
        model = tspgnn.build_network(64, float_dtype=float_dtype)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 3,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        b = sess.prepare(feed)
        step = sess.capture_train_step(b) if captured else (lambda: sess.train_step(b))
        losses = []
        for _ in range(3):
            out = step()
            losses.append(float(out["stats"][0].item()))
        torch.cuda.synchronize()
        finals.append((model.store.theta.clone(), losses, int(sess._adam["t"].item())))
    assert finals[0][2] == finals[1][2] == 3
    assert finals[0][1] == finals[1][1]
    assert torch.equal(finals[0][0], finals[1][0])


def test_batch_prefetcher_feeds_identical_batches(cuda_device):
    """parallel.BatchPrefetcher (background packing, side-stream upload, pinned or pageable staging) must deliver
    the same device batches as Session.prepare, in order."""
    rng = np.random.RandomState(3)
    host_batches = []
    for _ in range(4):
        base = [tspgnn.random_instance(int(rng.randint(6, 15)), rng) for _ in range(3)]
        host_batches.append(tspgnn.InstanceLoader.create_batch([i for i in base for _ in (0, 1)], dev=0.02))
    params = P.init_params(32, seed=1)
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    want = []
    for t in host_batches:
        EV, W, C, r, nv, ne = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 3, model["route_exists"]: r,
                model["n_vertices"]: nv, model["n_edges"]: ne}
        want.append(sess.forward(feed)["predictions"].clone())
    for pinned in (False, True):
        got = [sess.forward_device(b)["predictions"].clone()
               for b in tspgnn.BatchPrefetcher(sess, host_batches, 3, pinned=pinned)]
        torch.cuda.synchronize()
        assert len(got) == 4 and all(torch.equal(a, b) for a, b in zip(got, want))
    # many batches without any synchronisation between them (batches are dropped while their kernels are still
    # queued: the uploads' memory must not be recycled underneath them), and a failing producer is reported
    stream = [host_batches[i % 4] for i in range(40)]
    for workers in (1, 3):   # (several packer threads, each with its own upload stream: still in order)
        got = [sess.forward_device(b)["predictions"] for b in tspgnn.BatchPrefetcher(sess, stream, 3, workers=workers)]
        torch.cuda.synchronize()
        assert all(torch.equal(g, want[i % 4]) for i, g in enumerate(got))

    def broken():
        yield host_batches[0]
        raise KeyError("producer died")
    with pytest.raises(RuntimeError):
        for _ in tspgnn.BatchPrefetcher(sess, broken(), 3):
            pass


def test_captured_graph_serves_other_batches_of_the_same_shape(cuda_device):
    """DeviceBatch.copy_from: one captured forward graph replayed over a stream of same-shaped batches."""
    rng = np.random.RandomState(11)
    params = P.init_params(32, seed=1)
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)

    def feed_of(t):
        EV, W, C, r, nv, ne = t
        return {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 3, model["route_exists"]: r,
                model["n_vertices"]: nv, model["n_edges"]: ne}
    batches = [tspgnn.InstanceLoader.create_batch([tspgnn.random_instance(9, rng) for _ in range(4)], dev=0.02) for _ in range(3)]
    static = sess.prepare(feed_of(batches[0]))
    replay = sess.capture_forward(static)
    for t in batches:
        want = sess.forward(feed_of(t))["predictions"].clone()
        static.copy_from(sess.prepare(feed_of(t)))
        assert torch.equal(replay()["predictions"], want)
    other = tspgnn.InstanceLoader.create_batch([tspgnn.random_instance(8, rng) for _ in range(4)], dev=0.02)
    with pytest.raises(ValueError):
        static.copy_from(sess.prepare(feed_of(other)))


def test_save_and_load_weights_roundtrip_with_optimizer_state(cuda_device, tmp_path, capsys):
    """util.save_weights / load_weights (reference util.py:5-37): TensorFlow-bundle files keyed by TF variable names;
    a restored session continues training on exactly the trajectory of the one that saved."""
    from tspgnn.train import run_batch
    t = pack_tuple("ragged_B6", 0)
    params = P.init_params(32, seed=4, perturb=True)

    def fresh():
        model = tspgnn.build_network(32)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        return model, sess
    model, sess = fresh()
    model.store.load(params)
    for i in range(2):
        run_batch(sess, model, t, i, 0, 3, train=True, verbose=(i == 0))
    assert "[train] epoch 0  batch=0" in capsys.readouterr().out
    path = str(tmp_path / "checkpoints" / "epoch=7")
    tspgnn.save_weights(sess, path)
    names = set(tspgnn.tf_checkpoint.read_bundle(path + "/model.ckpt"))
    assert "TSP/E_cell/layer_norm_basic_lstm_cell/kernel" in names and "TSP/V_init/Adam" not in names
    assert "V_init/Adam_1" in names and "beta2_power" in names
    out_a = run_batch(sess, model, t, 2, 0, 3, train=True, verbose=False)
    theta_a = model.store.theta.clone()

    model_b, sess_b = fresh()
    assert tspgnn.load_weights(sess_b, path) == 7
    assert int(sess_b._adam["t"].item()) == 2
    out_b = run_batch(sess_b, model_b, t, 2, 0, 3, train=True, verbose=False)
    assert out_a[0] == out_b[0] and torch.equal(theta_a, model_b.store.theta)
    # scope-restricted restore only touches that scope
    model_c, sess_c = fresh()
    before = model_c.store.state_dict()
    tspgnn.load_weights(sess_c, path, scope="TSP/")
    after = model_c.store.state_dict()
    saved = tspgnn.tf_checkpoint.read_bundle(path + "/model.ckpt")
    for k in before:
        if k.startswith("TSP/"):
            assert np.array_equal(after[k], saved[k])
        else:
            assert np.array_equal(after[k], before[k])
    with pytest.raises(Exception):
        tspgnn.load_weights(sess_c, str(tmp_path / "nope" / "epoch=1"))


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("sizes,T", [((2,), 3), ((2, 3), 1), ((3, 2, 5, 2), 4), ((17,), 2)])
def test_tiny_and_single_graph_batches(cuda_device, d, sizes, T):
    """Degenerate shapes: a 2-vertex graph is ONE edge (M=1 < one 16-row tile), batches of one graph,
    tiles that straddle several graphs.  Same parity bar as the full-size cases."""
    rng = np.random.RandomState(sum(sizes) + d)
    instances = [tspgnn.random_instance(n, rng) for n in sizes]
    t = tspgnn.InstanceLoader.create_batch(instances, dev=0.02)
    params = P.init_params(d, seed=21, perturb=True)
    hip = run_hip(d, params, t, T)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    assert hip["predictions"].shape == (len(sizes),)
    assert rel_err(hip["predictions"], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(hip["last_states"]["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert rel_err(hip["last_states"]["V"].c, ref["last_states"]["V"][1].numpy()) < REL_TOL
    assert abs(float(hip["loss"]) - ref["loss"].item()) < REL_TOL


def test_vertex_without_edges_and_training_on_tiny_batch(cuda_device):
    """A vertex of degree 0 (empty CSR row: its cell input is the zero vector, its bias-init scale is 0) and a
    training step on a batch smaller than one tile."""
    d, T = 64, 3
    Ma = np.zeros((4, 4), dtype=int)
    Ma[0, 1] = Ma[1, 2] = 1                      # vertex 3 is isolated
    Mw = np.random.RandomState(0).rand(4, 4)
    t = tspgnn.InstanceLoader.create_batch([(Ma, Mw, [0, 1, 2, 3]), (Ma, Mw, [0, 1, 2, 3])], dev=0.02)
    params = P.init_params(d, seed=2, perturb=True)
    hip = run_hip(d, params, t, T)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    assert rel_err(hip["last_states"]["V"].h, ref["last_states"]["V"][0].numpy()) < REL_TOL
    assert rel_err(hip["predictions"], ref["predictions"].numpy()) < REL_TOL
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    out = tspgnn.run_batch(sess, model, t, 0, 0, T, train=True, verbose=False)
    ref_out, _, _, _, gn = TO.train_step({k: v.copy() for k, v in params.items()}, batch, T,
                                         {k: np.zeros_like(v) for k, v in params.items()},
                                         {k: np.zeros_like(v) for k, v in params.items()}, 1)
    assert abs(float(out[0]) - ref_out["loss"].item()) < REL_TOL
    assert abs(float(sess._adam["gnorm"].item()) - gn) < 2e-5 * gn


@pytest.mark.parametrize("name,d,T", [("n5_B2", 32, 3), ("ragged_B6", 64, 4), ("n20_B32", 64, 8), ("n20_B32", 128, 4)])
def test_bf16_storage_mode(cuda_device, name, d, T):
    """build_network(d, float_dtype=torch.bfloat16): BASELINE config 5's "bf16 embeddings with fp32 accumulate".
    Against the oracle restating the same rounding points (a handful of bf16 ulps: the two sides can land on
    different sides of a rounding boundary), and against the fp32 path (bf16 storage is an approximation of it)."""
    t = pack_tuple(name)
    params = P.init_params(d, seed=11, perturb=True)
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, last, loss = sess.run([model["predictions"], model["last_states"], model["loss"]], feed_dict=feed)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T, bf16=True)
    full = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    e_h = rel_err(last["E"].h, ref["last_states"]["E"][0].numpy())
    e_c = rel_err(last["V"].c, ref["last_states"]["V"][1].numpy())
    e_p = rel_err(pred, ref["predictions"].numpy())
    a_h = rel_err(last["E"].h, full["last_states"]["E"][0].numpy())
    a_p = rel_err(pred, full["predictions"].numpy())

    def rms(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.sqrt(((a - b) ** 2).mean()) / np.abs(b).max())
    r_h = rms(last["E"].h, ref["last_states"]["E"][0].numpy())
    print("\n[%s d=%d T=%d bf16] vs bf16 oracle: E.h %.2e (rms %.2e) V.c %.2e pred %.2e | vs fp32 semantics: E.h %.2e pred %.2e"
          % (name, d, T, e_h, r_h, e_c, e_p, a_h, a_p))
    # against the oracle that rounds at the same points: single entries may sit on the other side of a bf16 rounding
    # boundary (whole ulps of 2^-8 at the top of the range), the bulk agrees to a fraction of an ulp
    assert e_h < 2e-2 and e_c < 1e-2 and r_h < 2e-3 and e_p < 1e-3
    assert a_h < 4e-2 and a_p < 1e-2
    assert abs(float(loss) - ref["loss"].item()) < 5e-4


def teacher_forced_bf16_grads(params_np, batch, T, H, C):
    """Back-propagation through time of the bf16-storage forward, step by step on the STORED states of a run: the
    vector-Jacobian product of every step (oracle step_bf16 in float64, roundings straight through) is taken at the
    H[t], C[t] the device kept -- exactly what a backward pass that reads a tape computes.  Unlike the end-to-end
    oracle gradient (whose own forward takes different rounding decisions, amplified over the recurrence), this
    differs from the device's only by the roundings inside one step.  -> {name: gradient} without the L2 term."""
    params = TO.to_torch(params_np, torch.float64, requires_grad=True)
    names, plist = list(params.keys()), list(params.values())
    total = [torch.zeros_like(p) for p in plist]
    uv = torch.as_tensor(np.asarray(batch["ev_uv"]), dtype=torch.long)

    def leaf(a):
        return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)

    def vjp(scalar, leaves):
        g = torch.autograd.grad(scalar, leaves + plist, allow_unused=True)
        for k, gk in enumerate(g[len(leaves):]):
            if gk is not None:
                total[k] += gk
        return [torch.zeros_like(l) if gi is None else gi for l, gi in zip(leaves, g[:len(leaves)])]
    Eh = leaf(H["E"][T])
    dEh, = vjp(TO.vote_head(params, batch, Eh)["loss"], [Eh])
    dVh, dVc, dEc = torch.zeros(H["V"][T].shape, dtype=torch.float64), torch.zeros(C["V"][T].shape, dtype=torch.float64), \
        torch.zeros(C["E"][T].shape, dtype=torch.float64)
    for t in range(T - 1, -1, -1):
        leaves = [leaf(H["V"][t]), leaf(C["V"][t]), leaf(H["E"][t]), leaf(C["E"][t])]
        nVh, nVc, nEh, nEc = TO.step_bf16(params, uv, *leaves)
        dVh, dVc, dEh, dEc = vjp((nVh * dVh).sum() + (nVc * dVc).sum() + (nEh * dEh).sum() + (nEc * dEc).sum(), leaves)
    V0, E0 = TO.initial_embeddings(params, batch)      # (their rounding for storage passes the gradient through)
    vjp((V0 * dVh).sum() + (E0 * dEh).sum(), [])
    return {k: g.detach().numpy() for k, g in zip(names, total)}


@pytest.mark.parametrize("name,d,T", [("n5_B2", 32, 3), ("ragged_B6", 64, 4), ("n20_B32", 64, 6), ("ragged_B6", 128, 3)])
def test_bf16_storage_training_gradients(cuda_device, name, d, T):
    """Mixed-precision training in the bf16-storage mode: bf16 tape, fp32 gradients of the function the forward
    evaluated (roundings passed straight through, GEMM weights rounded to bf16), fp32 master variables.
      (a) tight: against step-by-step autograd on the float64 oracle AT THE STATES THE DEVICE STORED
          (teacher_forced_bf16_grads);
      (b) end to end: against autograd through the oracle's own bf16 forward.  That gradient is itself only defined up
          to the rounding decisions of its forward -- a 1e-6 relative perturbation of the variables moves it by
          percents on these batches (the +-dev instance pairs nearly cancel in the mean) -- so the device's gradient is
          asked to sit within that spread."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=21, perturb=True)
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    out = sess.loss_and_grads(feed, keep_tape=True)
    torch.cuda.synchronize()
    g = model.store.grad_dict()
    tape = out["tape"]
    assert tape.H["E"].dtype == torch.bfloat16 and tape.C["E"].dtype == torch.float32
    H = {v: tape.H[v].to(torch.float32).cpu().numpy() for v in ("V", "E")}
    Cs = {v: tape.C[v].cpu().numpy() for v in ("V", "E")}
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}

    def l2_dist(a, b):
        return float(np.sqrt(sum(((a[k] - b[k]) ** 2).sum() for k in b) / sum((b[k] ** 2).sum() for k in b)))
    forced = teacher_forced_bf16_grads(params, batch, T, H, Cs)
    gscale = max(np.abs(forced[k]).max() for k in forced)
    worst = max(np.abs(g[k] - forced[k]).max() / max(np.abs(forced[k]).max(), 1e-2 * gscale) for k in forced)
    tight = l2_dist(g, forced)
    ref_out, ref_g = TO.loss_and_grads(params, batch, T, dtype=torch.float64, bf16=True)
    l2 = {k: TO.L2NORM_SCALING * params[k] for k in params}
    ref = {k: ref_g[k] - l2[k] for k in ref_g}
    nudged = {k: (v * (1 + 1e-6 * np.random.RandomState(1).randn(*v.shape))).astype(v.dtype) for k, v in params.items()}
    _, ref2 = TO.loss_and_grads(nudged, batch, T, dtype=torch.float64, bf16=True)
    spread = l2_dist({k: ref2[k] - l2[k] for k in ref2}, ref)
    end_to_end = l2_dist(g, ref)
    print("\n[%s d=%d T=%d bf16 training] gradient vs teacher-forced oracle: L2 %.2e, worst per-variable %.2e; vs the oracle's "
          "own forward: L2 %.2e (its spread under a 1e-6 nudge: %.2e)" % (name, d, T, tight, worst, end_to_end, spread))
    assert abs(float(out["stats"][0].item()) - ref_out["loss"].item()) < 5e-4
    # (measured, native bf16-reading backward == widened-tape backward to the printed digits: L2 1.5e-5 .. 2.4e-3, worst
    # variable 6e-5 .. 6.0e-3; what is left is the teacher forcing itself -- the oracle re-derives the step's bf16
    # messages / aggregates from the forced states and may round one of them the other way than the device did)
    assert tight < 3e-3 and worst < 8e-3
    assert end_to_end < 3 * max(spread, 1e-2)


def test_bf16_training_large_weight_keeps_the_fp32_mlp_backward(cuda_device):
    """The bf16-storage backward runs the message MLPs' data gradient on the fp16 matrix cores (tspgnn_mlp_bwd_multi_h2), whose
    packing 2^6 W^T overflows fp16 for |w| >= 1024 -- a magnitude bf16 storage itself carries.  The pass packs first, looks at
    the range guard's weight word and keeps the fp32 matrix instruction when it is beyond half the fp16 range: a network with
    one such weight gets the gradients of the fp32 path, bit for bit, and they are finite."""
    d, T = 64, 3
    t = pack_tuple("ragged_B6", 1)
    params = P.init_params(d, seed=4, perturb=True)
    params = {k: np.array(v, copy=True) for k, v in params.items()}
    params["TSP/E_msg_V_MLP_layer_2/kernel"][3, 5] = 2000.0      # (bf16-exact; 2^6 * 2000 > 65504)
    res = []
    for h2 in (True, False):
        model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        model["gnn"].mlp_backward_h2 = h2
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        sess.loss_and_grads(feed)
        torch.cuda.synchronize()
        res.append(model.store.grad_dict())
    for k in res[0]:
        assert np.isfinite(res[0][k]).all(), k
        assert np.array_equal(res[0][k], res[1][k]), k


def test_bf16_storage_train_steps(cuda_device):
    """sess.run(train_step) in the bf16-storage mode: the loss follows the straight-through oracle's over three Adam
    steps and the fp32 master variables move as the oracle's do."""
    t = pack_tuple("ragged_B6", 0)
    d, T = 64, 4
    params = P.init_params(d, seed=2, perturb=True)
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    batch = {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices, "n_edges": n_edges}
    p = {k: v.copy() for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v) for k, v in p.items()}
    for step in (1, 2, 3):
        vals = sess.run([model["train_step"], model["loss"]], feed_dict=feed)
        out, g = TO.loss_and_grads(p, batch, T, dtype=torch.float64, bf16=True)
        g, gn = TO.clip_by_global_norm(g)
        p, m, v = TO.adam_step(p, g, m, v, step)
        assert abs(float(vals[1]) - out["loss"].item()) < 5e-4
        assert abs(float(sess._adam["gnorm"].item()) - gn) < 3e-2 * gn
    now = model.store.state_dict()
    moved = sum(float(np.abs(now[k] - params[k]).max() > 0) for k in p)
    assert moved == len(p)                               # every variable took its Adam steps
    for k in p:   # Adam's sign-like first steps: the directions agree where the gradient is not at the noise level
        ref, got = p[k] - params[k], now[k].astype(np.float64) - params[k]
        big = np.abs(ref) > 0.5 * np.abs(ref).max()
        # (bf16 storage: an entry whose gradient is within the mode's rounding noise of zero may step the other way -- which
        # entries those are depends on the last bit of E_init's fp32 outputs before they are rounded to bf16 -- so the bar
        # is a fraction, not every entry)
        agree = float(np.mean(np.sign(ref[big]) == np.sign(got[big])))
        assert agree > 0.995, (k, agree)


@pytest.mark.parametrize("name,d,T", [("ragged_B6", 64, 4), ("n20_B32", 64, 5)])
def test_pushed_training_gradients_equal_the_plain_form(cuda_device, name, d, T):
    """Training with the message MLP's last linear layer pushed through the row-sum into the vertex cell (the default,
    f16x2) against the plain form: same loss, same gradients up to the order of the fp32 sums -- including the three
    variables whose gradients the pushed form assembles from d(W Kx) and d(b Kx)."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=8, perturb=True)
    grads = []
    for push in (True, False):
        model = tspgnn.build_network(d)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        model["gnn"].push_training = push
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        out = sess.loss_and_grads(feed, keep_tape=True)
        torch.cuda.synchronize()
        assert out["tape"].pushed["V"] == push and not out["tape"].pushed["E"]
        grads.append((float(out["stats"][0].item()), model.store.grad_dict()))
    (loss_p, gp), (loss_u, gu) = grads
    assert abs(loss_p - loss_u) < 1e-6
    gscale = max(np.abs(gu[k]).max() for k in gu)
    for k in gu:
        scale = max(np.abs(gu[k]).max(), 1e-3 * gscale)
        # (two fp32 evaluation orders of a batch whose +-dev instance pairs nearly cancel in the mean: a few 1e-5; each
        # form separately meets the oracle bar in test_gradient_parity_with_autograd_oracle)
        assert np.abs(gp[k] - gu[k]).max() / scale < 1e-4, k


@pytest.mark.parametrize("name,d,T", [("ragged_B6", 64, 4), ("n20_B32", 64, 5), ("n5_B2", 64, 1)])
def test_recomputed_training_gradients_equal_the_taped_form(cuda_device, name, d, T):
    """Training with the pushed message MLP's backward recomputing its hidden activations and forming the MLP's weight
    gradients in the same launch (tspgnn_mlp_bwd_rc_h2, opt-in; the forward then tapes only the messages and runs the MLP
    inside the cell launch) against the taped form: same loss, bit-identical states, gradients equal up to the arithmetic of
    the data gradient (fp16 matrix cores on scaled splits instead of the fp32 matrix instruction)."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=8, perturb=True)
    grads = []
    for rc in (True, False):
        model = tspgnn.build_network(d)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        model["gnn"].recompute_messages = rc
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        out = sess.loss_and_grads(feed, keep_tape=True)
        torch.cuda.synchronize()
        tape = out["tape"]
        assert tape.pushed["V"] and bool(tape.rc.get(("V", 0))) == rc and tape.fused == rc
        assert tape.acts[("V", 0)].shape[0] == (1 if rc else 3)
        grads.append((float(out["stats"][0].item()), model.store.grad_dict(), tape))
    (loss_r, gr, tr), (loss_t, gt, tt) = grads
    assert loss_r == loss_t
    for v in ("V", "E"):
        assert torch.equal(tr.H[v], tt.H[v]) and torch.equal(tr.C[v], tt.C[v])
    assert torch.equal(tr.acts[("V", 0)][0], tt.acts[("V", 0)][2])     # the messages = the last hidden activation
    gscale = max(np.abs(gt[k]).max() for k in gt)
    for k in gt:
        scale = max(np.abs(gt[k]).max(), 1e-3 * gscale)
        assert np.abs(gr[k] - gt[k]).max() / scale < 1e-4, k


@pytest.mark.parametrize("name,d,T", [("ragged_B6", 64, 4), ("n20_B32", 64, 5), ("n5_B2", 64, 1)])
def test_data_gradient_gemms_inside_the_backward_launches_equal_their_own_launches(cuda_device, monkeypatch, name, d, T):
    """The backward step with the vertex side's two data-gradient GEMMs riding in the cell and message-MLP launches
    (tspgnn_lstm_bwd_task.KTg, tspgnn_mlp_bwd_task.pre_X: the default) against TSPGNN_FUSE_DATA_GRADIENTS=0, where each is a
    tspgnn_linear_f32 launch on the fp32 matrix instruction: same loss, gradients equal up to the arithmetic of those two
    GEMMs (fp16 matrix cores on scaled splits)."""
    t = pack_tuple(name, 1)
    params = P.init_params(d, seed=8, perturb=True)
    grads = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("TSPGNN_FUSE_DATA_GRADIENTS", fuse)
        model = tspgnn.build_network(d)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        out = sess.loss_and_grads(feed)
        torch.cuda.synchronize()
        grads.append((float(out["stats"][0].item()), model.store.grad_dict()))
    (loss_f, gf), (loss_u, gu) = grads
    assert loss_f == loss_u
    gscale = max(np.abs(gu[k]).max() for k in gu)
    for k in gu:
        scale = max(np.abs(gu[k]).max(), 1e-3 * gscale)
        assert np.abs(gf[k] - gu[k]).max() / scale < 2e-5, k
    assert any(not np.array_equal(gf[k], gu[k]) for k in gu)      # (the switch does switch something)


def test_weight_gradient_chunks_agree(cuda_device):
    """GraphNN.backward reduces the weight gradients per chunk of time steps (all T when they fit the budget): one,
    two and five chunks give the same gradients up to the order of the fp32 sums."""
    t = pack_tuple("ragged_B6", 1)
    d, T = 64, 5
    params = P.init_params(d, seed=3, perturb=True)
    _, _, _, _, g_all = grads_hip(d, params, t, T)
    M, N = t[0].shape
    per_step = (M + N) * 4 * d * 4 + 4 * (M + N) * d * 4     # dz + the four layers' dpre, per row and step
    for steps in (3, 1):
        _, _, _, _, g = grads_hip(d, params, t, T, chunk_bytes=steps * per_step + per_step // 2)
        for k in g_all:
            scale = max(np.abs(g_all[k]).max(), 1e-12)
            assert np.abs(g[k] - g_all[k]).max() / scale < 2e-5, (steps, k)


def test_experiment_sweeps_on_graph_files(cuda_device, tmp_path):
    """experiments.acceptance_curve / accuracy_by_size over InstanceLoader directories of .graph files: the values
    are the plain means of per-batch fetches (cross-checked against direct sess.run calls)."""
    from tspgnn import experiments as X
    rng = np.random.RandomState(3)
    dirs = {}
    for n in (6, 9):
        p = tmp_path / ("n=%d" % n)
        p.mkdir()
        for i in range(4):
            Ma, Mw, route = tspgnn.random_instance(n, rng)
            tspgnn.write_graph(Ma, Mw, str(p / ("%d.graph" % i)), route=route)
        dirs[n] = str(p)
    model = tspgnn.build_network(32)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer(seed=5))
    loaders = {n: tspgnn.InstanceLoader(d) for n, d in dirs.items()}
    devs = [-0.5, 0.0, 0.5]
    curve = X.acceptance_curve(sess, model, loaders[6], 3, devs, batch_size=2, max_batches=2)
    assert curve.shape == (3,) and np.all((curve > 0) & (curve < 1))
    loaders[6].reset()
    direct = np.mean(np.concatenate([X.get_predictions(sess, model, b, 3) for b in loaders[6].get_batches(2, 0.0)]))
    assert abs(curve[1] - direct) < 1e-7
    acc = X.accuracy_by_size(sess, model, loaders, 3, 0.02, batch_size=2, max_batches=2)
    assert set(acc) == {6, 9} and all(0.0 <= a <= 1.0 for a in acc.values())


def test_random_shapes_parity(cuda_device):
    """A fixed prefix of tests/fuzz_parity.py's random sequence: batch size, graph sizes (3..45), sparsity, d and T
    drawn at random; forward parity for both GEMM arithmetics, gradients on every third case.  (The full sweep,
    60 cases: forward <= 1.9e-6, run with `python tests/fuzz_parity.py 60`.)"""
    import fuzz_parity
    rng = np.random.RandomState(2024)
    for i in range(10):
        fuzz_parity.run_case(i, fuzz_parity.draw_case(rng), with_grads=(i % 3 == 0))


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "f32"])
def test_edge_order_within_a_problem_is_free(cuda_device, gemm):
    """The reference orders a problem's edges as np.nonzero walks the upper triangle (instance_loader.py:60-66); the
    kernels must not depend on it: edges shuffled within each problem (and the endpoints of half of them swapped)
    give the same predictions, and states that are the same rows in the new order -- against the oracle on the
    shuffled batch, and against the unshuffled HIP run."""
    t = pack_tuple("ragged_B6", 2)
    EV, W, C, r, nv, ne = t
    rng = np.random.RandomState(4)
    eo = np.concatenate([[0], np.cumsum(ne)])
    perm = np.concatenate([eo[i] + rng.permutation(int(ne[i])) for i in range(len(ne))])
    uv = EV.uv[perm].copy()
    swap = rng.rand(len(uv)) < 0.5
    uv[swap] = uv[swap][:, ::-1]
    t2 = (tspgnn.SparseEV(uv, EV.shape[1]), W[perm], C[perm], r, nv, ne)
    params = P.init_params(64, seed=5, perturb=True)
    a = run_hip(64, params, t, 5, fetch=("predictions", "last_states"), gemm=gemm)
    b = run_hip(64, params, t2, 5, fetch=("predictions", "last_states"), gemm=gemm)
    ref = TO.forward(TO.to_torch(params, torch.float64), batch_from_tuple(t2), 5)
    assert rel_err(b["predictions"], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(b["last_states"]["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert rel_err(b["predictions"], a["predictions"]) < 2e-6
    assert rel_err(b["last_states"]["E"].h, a["last_states"]["E"].h[perm]) < 5e-6
    assert rel_err(b["last_states"]["V"].c, a["last_states"]["V"].c) < 5e-6


# ---------------------------------------------------------------------------------------- f16x2 range guard
# "No input that is finite in the reference's fp32 path is non-finite here": the default arithmetic (fp16 pieces) has a
# narrower range than the reference's float32 (graphnn.py:18, model.py:18-27); weights are vetted when packed, activations
# where the kernels split them, and the batch runs on bf16x3 (fp32's range) instead.

def _oracle_batch(t):
    return {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}


@pytest.mark.parametrize("which", ["TSP/E_cell/layer_norm_basic_lstm_cell/kernel", "TSP/V_msg_E_MLP_layer_2/kernel",
                                   "E_vote_MLP_layer_1/kernel"])
def test_f16x2_falls_back_for_weights_beyond_the_fp16_range(cuda_device, which):
    """One weight of 2000 (2^6 * 2000 overflows the fp16 hi piece): the packing's guard word vetoes f16x2 before a kernel
    multiplies with it; predictions and states stay within 1e-5 of the float64 oracle."""
    d, T = 64, 4
    t = pack_tuple("ragged_B6", 1)
    params = P.init_params(d, seed=21, perturb=True)
    name = [k for k in params if k.endswith(which) or k == which]
    assert len(name) == 1, (which, sorted(params))
    params[name[0]] = params[name[0]].copy()
    params[name[0]][5, 7] = 2000.0
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    assert model["gnn"].active_arith() == "h2"
    pred, last, loss = sess.run([model["predictions"], model["last_states"], model["loss"]], feed_dict=feed)
    assert model["gnn"].active_arith() == "x3"          # latched for these variables ...
    ref = TO.forward(TO.to_torch(params, torch.float64), _oracle_batch(t), T)
    assert np.all(np.isfinite(pred)) and rel_err(pred, ref["predictions"].numpy()) < REL_TOL
    assert rel_err(last["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert abs(float(loss) - ref["loss"].item()) < REL_TOL
    model.store.load(P.init_params(d, seed=21, perturb=True))
    assert model["gnn"].active_arith() == "h2"          # ... and f16x2 is back after the next assignment


def test_f16x2_falls_back_for_activations_beyond_the_fp16_range(cuda_device):
    """A batch whose pushed row-sum exceeds 65504: the E->V message MLP's last hidden layer gets a bias of 3000, so every
    edge sends ~3000 per unit and a vertex of a 20-vertex graph sums 19 of them (57 000) -- in range --, of a ragged
    batch's 40-vertex graph 39 (117 000) -- out of range.  run() notices the kernels' flag and repeats the batch on
    bf16x3; the weights themselves stay eligible for f16x2 (the next batch tries it again)."""
    d, T = 64, 3
    params = P.init_params(d, seed=5, perturb=True)
    key = [k for k in params if k.endswith("E_msg_V_MLP_layer_3/bias")]
    assert len(key) == 1, sorted(params)
    params[key[0]] = np.full_like(params[key[0]], 3000.0)
    rng = np.random.RandomState(0)
    small = tspgnn.synthetic_batch([20, 20], seed=3)
    large = tspgnn.synthetic_batch([20, 40], seed=3)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    for t, overflow in ((small, False), (large, True), (small, False)):
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        # the unguarded forward: finite for the small batch, flagged for the large one
        sess.forward_device(sess.prepare(feed))
        assert sess.range_exceeded() == overflow
        pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
        ref = TO.forward(TO.to_torch(params, torch.float64), _oracle_batch(t), T)
        assert np.all(np.isfinite(pred)) and np.all(np.isfinite(last["V"].c))
        assert rel_err(pred, ref["predictions"].numpy()) < REL_TOL
        assert rel_err(last["V"].h, ref["last_states"]["V"][0].numpy()) < REL_TOL
        assert rel_err(last["E"].c, ref["last_states"]["E"][1].numpy()) < REL_TOL
        assert model["gnn"].active_arith() == "h2" and not sess.range_exceeded()


def test_training_step_survives_an_activation_overflow(cuda_device):
    """sess.run(train_step) on the overflowing batch: the optimiser kernel skips the update of the flagged f16x2 attempt
    on the device, the step is repeated on bf16x3 -- the variables end where a session restricted to bf16x3 puts them."""
    d, T = 64, 2
    params = P.init_params(d, seed=6, perturb=True)
    key = [k for k in params if k.endswith("E_msg_V_MLP_layer_3/bias")][0]
    params[key] = np.full_like(params[key], 3000.0)
    t = tspgnn.synthetic_batch([20, 40], seed=4)
    ends = {}
    for gemm in ("f16x2", "bf16x3"):
        model = tspgnn.build_network(d)
        model["gnn"].gemm = gemm
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        for _ in range(2):
            _, loss = sess.run([model["train_step"], model["loss"]], feed_dict=feed)
            assert np.isfinite(loss)
        assert int(sess._adam["t"].item()) == 2 and sess._adam["step"] == 2
        ends[gemm] = model.store.state_dict()
    for k in ends["bf16x3"]:
        assert np.all(np.isfinite(ends["f16x2"][k])), k
        assert np.array_equal(ends["f16x2"][k], ends["bf16x3"][k]), k


def test_f16x2_falls_back_when_only_the_pushed_product_leaves_the_range(cuda_device):
    """The pushed form multiplies with K' = [W Kx ; Kh]: W and Kx can each be inside the fp16 range while an entry of their
    product is not (here 40 * 40 * 64 columns' worth).  The product's packing is vetted before the plan's first launch."""
    d, T = 64, 3
    t = pack_tuple("ragged_B6", 2)
    params = P.init_params(d, seed=9, perturb=True)
    wk = [k for k in params if k.endswith("E_msg_V_MLP_layer_4/kernel")][0]
    kk = [k for k in params if k.endswith("V_cell/layer_norm_basic_lstm_cell/kernel")][0]
    params[wk] = params[wk].copy(); params[kk] = params[kk].copy()
    params[wk][3, :] = 40.0          # row 3 of W (in [64, 64]) ...
    params[kk][:64, 17] = 40.0       # ... times column 17 of Kx: (W Kx)[3, 17] = 64 * 1600 = 102 400; 2^6 * that overflows
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
    assert model["gnn"].active_arith() == "x3"
    ref = TO.forward(TO.to_torch(params, torch.float64), _oracle_batch(t), T)
    assert np.all(np.isfinite(pred)) and rel_err(pred, ref["predictions"].numpy()) < REL_TOL
    assert rel_err(last["V"].c, ref["last_states"]["V"][1].numpy()) < REL_TOL
    # and a training step with these variables runs on bf16x3 from its first launch
    _, loss = sess.run([model["train_step"], model["loss"]], feed_dict=feed)
    assert np.isfinite(loss) and all(np.all(np.isfinite(v)) for v in model.store.state_dict().values())


# ---------------------------------------------------------------- the LOW end of f16x2's range
def _low_end_params(case, d, seed):
    """Well-conditioned in the reference's fp32 (graphnn.py:18), small for an fp16 piece:
    a: the E->V message MLP's last layer scaled by 1e-4;  b: both cells' state/gamma = 1e-3 (all of h tiny);
    c: V_init = 0 and the E_init weights x 1e-3;  d: b AND every message-MLP bias zero AND every state/beta zero -- then
    every term of a cell's z is tiny and only LayerNorm's rescaling brings the gates back to O(1)."""
    p = {k: v.copy() for k, v in P.init_params(d, seed=seed, perturb=True).items()}
    if case == "a":
        for k in p:
            if "E_msg_V_MLP_layer_4/" in k:
                p[k] *= 1e-4
    if case in ("b", "d"):
        for k in p:
            if k.endswith("state/gamma"):
                p[k] = np.full_like(p[k], 1e-3) * np.sign(p[k] + 1e-30)
    if case == "c":
        p["V_init"] = np.zeros_like(p["V_init"])
        for k in p:
            if k.startswith("E_init_MLP") and k.endswith("kernel"):
                p[k] *= 1e-3
    if case == "d":
        for k in p:
            if ("_msg_" in k and k.endswith("bias")) or k.endswith("state/beta"):
                p[k] = np.zeros_like(p[k])
    return p


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "f32"])
def test_small_activations_keep_the_1e5_budget(cuda_device, case, gemm):
    """Inputs the reference's fp32 handles at 1e-7 whose activations sit where an fp16 piece is subnormal: every
    arithmetic stays within 1e-5 of the float64 oracle -- f16x2 either by itself or because the kernels' variance floor
    (range_flag bit 1) sent the batch to bf16x3."""
    d, T = 64, 6
    t = pack_tuple("ragged_B6", 1)
    params = _low_end_params(case, d, 31)
    model = tspgnn.build_network(d)
    model["gnn"].gemm = gemm
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, last, loss = sess.run([model["predictions"], model["last_states"], model["loss"]], feed_dict=feed)
    ref = TO.forward(TO.to_torch(params, torch.float64), _oracle_batch(t), T)
    errs = {"pred": rel_err(pred, ref["predictions"].numpy()),
            "E.h": rel_err(last["E"].h, ref["last_states"]["E"][0].numpy()),
            "E.c": rel_err(last["E"].c, ref["last_states"]["E"][1].numpy()),
            "V.h": rel_err(last["V"].h, ref["last_states"]["V"][0].numpy()),
            "V.c": rel_err(last["V"].c, ref["last_states"]["V"][1].numpy())}
    print("\n[low end %s %s] %s  guard bits %d  max|E.h| %.2e" % (
        case, gemm, "  ".join("%s %.2e" % kv for kv in errs.items()), sess.last_range_bits,
        float(np.abs(ref["last_states"]["E"][0].numpy()).max())))
    assert all(e < REL_TOL for e in errs.values()), errs
    assert abs(float(loss) - ref["loss"].item()) < REL_TOL
    if gemm != "f16x2":
        assert sess.last_range_bits == 0


def test_variance_floor_routes_a_tiny_z_to_bf16x3(cuda_device):
    """Case d (every term of z tiny): the unguarded f16x2 forward raises bit 1 of the range flag -- and only bit 1 --;
    run() answers with bf16x3, a training step on the batch ends where a bf16x3-only session ends, and a batch of
    ordinary size on the same session is not flagged."""
    d, T = 64, 6
    t = pack_tuple("ragged_B6", 1)
    params = _low_end_params("d", d, 31)
    ends = {}
    for gemm in ("f16x2", "bf16x3"):
        model = tspgnn.build_network(d)
        model["gnn"].gemm = gemm
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        EV, W, C, route_exists, n_vertices, n_edges = t
        feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
        if gemm == "f16x2":
            out = sess.forward_device(sess.prepare(feed))
            raw = out["last_states"]["E"].h.cpu().numpy().copy()
            assert int(sess.store.h2_guard()[0].item()) == 2
            assert sess.range_exceeded() and sess.last_range_bits == 2
            ref = TO.forward(TO.to_torch(params, torch.float64), _oracle_batch(t), T)
            print("\n[variance floor] unguarded f16x2 E.h error %.2e (bar %.0e)"
                  % (rel_err(raw, ref["last_states"]["E"][0].numpy()), REL_TOL))
        _, loss = sess.run([model["train_step"], model["loss"]], feed_dict=feed)
        assert np.isfinite(loss)
        assert int(sess._adam["t"].item()) == 1 and sess._adam["step"] == 1
        ends[gemm] = model.store.state_dict()
        if gemm == "f16x2":
            model.store.load(P.init_params(d, seed=31, perturb=True))
            sess.last_range_bits = 0
            sess.run(model["predictions"], feed_dict=feed)
            assert sess.last_range_bits == 0 and model["gnn"].active_arith() == "h2"
    for k in ends["bf16x3"]:
        assert np.array_equal(ends["f16x2"][k], ends["bf16x3"][k]), k


def test_train_step_clears_a_stale_range_flag(cuda_device):
    """A flag left up by an unchecked forward() must not make the optimiser kernel skip a direct train_step() -- and
    every one after it: train_step clears the word before the pass it judges."""
    d, T = 64, 2
    t = tspgnn.synthetic_batch([12, 12], seed=8)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    before = model.store.theta.clone()
    sess.store.h2_guard()[0:1].fill_(1)            # what an overflowing, unchecked forward() leaves behind
    for k in range(3):
        sess.train_step(sess.prepare(feed))
        assert int(sess._adam["t"].item()) == k + 1 and sess._adam["step"] == k + 1
    assert not torch.equal(before, model.store.theta)
    assert model["gnn"].active_arith() == "h2"


def test_replayed_training_raises_at_a_fixed_lag(cuda_device):
    """The range guard under capture_train_step: replay i looks at the guard words as replay i - 2 left them (a fixed lag, so
    that every rank of a data-parallel session raises at the same replay index).  A flag that goes up after replay 2 makes
    the device skip replays 3 and 4 (variables and step counter untouched) and replay 5 raise; the host mirror of the step
    counter is then the device's, f16x2 is off for these variables, and a new capture runs (on bf16x3)."""
    t = pack_tuple("ragged_B6", 1)
    model = tspgnn.build_network(64)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(P.init_params(64, seed=9, perturb=True))
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 2,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    sess.store.h2_guard()[0:1].fill_(1)          # a stale flag from an unchecked forward(): not this capture's business
    step = sess.capture_train_step(b)
    assert model["gnn"].active_arith() == "h2"
    for _ in range(3):                           # replays 0, 1, 2: applied
        step()
    torch.cuda.synchronize()
    assert int(sess._adam["t"].item()) == 3 and sess._adam["step"] == 3
    theta3 = model.store.theta.clone()
    sess.store.h2_guard()[0:1].fill_(2)          # what a kernel of replay 3 would raise (here: the variance floor's bit)
    step()                                       # replay 3: looks at replay 1's words (clean); skipped on the device
    step()                                       # replay 4: looks at replay 2's words (clean); skipped on the device
    torch.cuda.synchronize()
    assert int(sess._adam["t"].item()) == 3 and torch.equal(model.store.theta, theta3)
    with pytest.raises(RuntimeError, match="capture_train_step\\(\\) again"):
        step()                                   # replay 5: replay 3's words carry the flag
    assert sess._adam["step"] == 3 and sess.last_range_bits == 2
    assert model["gnn"].active_arith() == "x3" and int(sess.store.h2_guard()[0].item()) == 0
    step = sess.capture_train_step(b)
    step()
    torch.cuda.synchronize()
    assert int(sess._adam["t"].item()) == 4 and sess._adam["step"] == 4 and not torch.equal(model.store.theta, theta3)


def test_batch_stager_serves_fresh_batches_through_one_captured_graph(cuda_device):
    """parallel.BatchStager: fresh instances -> one native staging call into a pinned slot -> one upload -> one device copy
    into the buffer the captured graph reads.  Every batch's replayed predictions equal the plain prepare + forward of the
    same instances (bit for bit: same device arrays, same kernels), in the iterator's order; a batch of another shape is
    refused; the device batch the graph is bound to equals Session.prepare's arrays."""
    rng = np.random.RandomState(11)
    sizes = [12, 9, 12, 9, 20, 20, 7, 7]
    pool = [[tspgnn.random_instance(n, rng) for n in sizes] for _ in range(6)]
    d, T = 64, 3
    params = P.init_params(d, seed=6, perturb=True)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)

    def feed_of(inst):
        EV, W, C, r, nv, ne = tspgnn.InstanceLoader.create_batch(inst, dev=0.02)
        return {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
                model["n_vertices"]: nv, model["n_edges"]: ne}
    want = [sess.forward_device(sess.prepare(feed_of(inst)))["predictions"].clone() for inst in pool]
    stager = tspgnn.BatchStager(sess, pool[0], T)
    ref = sess.prepare(feed_of(pool[0]))
    for a, b in zip(stager.batch.tensors(), ref.tensors()):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
    replay = sess.capture_forward(stager.batch)
    got = []
    for _ in stager.feed(pool * 3):
        got.append(replay()["predictions"].clone())
    torch.cuda.synchronize()
    assert len(got) == 18
    for i, p in enumerate(got):
        assert torch.equal(p, want[i % 6]), i
    with pytest.raises(RuntimeError):
        for _ in stager.feed([pool[0][:-1]]):
            pass
    assert not sess.range_exceeded()
