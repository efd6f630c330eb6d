"""CPU restatement of the TSP-GNN hot path in torch (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED: the reference's arithmetic lives in TensorFlow 1.x (tf.contrib), which
is neither vendored nor installable here and the reference holds no
golden vectors for this path (O5).  What this file restates, line by line:

* Mlp.__call__                      /root/reference/mlp.py:57-63
* GraphNN.__call__ / while_body      /root/reference/graphnn.py:134-179
* build_network pre-loop             /root/reference/model.py:33-51
* build_network post-loop            /root/reference/model.py:107-157
* L2 / clip / Adam                   /root/reference/model.py:160-167
* tf.contrib.rnn.LayerNormBasicLSTMCell, tf.contrib.layers.layer_norm, tf.layers.Dense,
  tf.nn.sigmoid_cross_entropy_with_logits, tf.clip_by_global_norm, tf.train.AdamOptimizer:
  restated from the published TF-1.x semantics.

Two variants of the adjacency product (graphnn.py:156-160):
``dense=True``  multiplies by the dense block-diagonal EV[M,N] exactly like tf.matmul does
                (op-for-op; this is what ``bench.py``'s cpu_baseline times), and
``dense=False`` uses the endpoint list (index_add / gather) -- same mathematics, no zeros.
Gradients come from torch autograd on this restatement (float64 for the oracle).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import params as P

LN_EPS = 1e-12  # tf.contrib.layers.layer_norm: variance_epsilon = 1e-12
FORGET_BIAS = 1.0  # LayerNormBasicLSTMCell default
LEARNING_RATE = 2e-5  # model.py:13
L2NORM_SCALING = 1e-10  # model.py:14
CLIP_NORM = 0.65  # model.py:15
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8  # tf.train.AdamOptimizer defaults


def to_torch(params, dtype=torch.float64, requires_grad=False):
    out = OrderedDict()
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def mlp(x, params, prefix, n_layers=4, last_activation=False):
    """mlp.py:57-63 with activations [relu]*3 + [None] (graphnn.py:116-119, model.py:34-37)."""
    for i in range(n_layers):
        W = params["%s_MLP_layer_%d/kernel" % (prefix, i + 1)]
        b = params["%s_MLP_layer_%d/bias" % (prefix, i + 1)]
        x = x @ W + b  # tf.layers.Dense: x @ kernel + bias, kernel[in,out]
        if i < n_layers - 1 or last_activation:
            x = torch.relu(x)
    return x


def layer_norm(x, gamma, beta):
    """tf.contrib.layers.layer_norm over the last axis: nn.moments (biased variance) +
    nn.batch_normalization evaluated as x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + LN_EPS) * gamma
    return x * inv + (beta - mean * inv)


def lnlstm_cell(x, h, c, params, cell, activation=torch.relu, base=None):
    """tf.contrib.rnn.LayerNormBasicLSTMCell.call with activation=relu (graphnn.py:110,168-170).
    args = concat([inputs, h]); no bias; gates i,j,f,o; forget bias added after LN;
    the *normalised* new_c is both stored and fed to the output.
    ``activation`` (the cell's own constructor argument; the hot path always passes relu) exists so that the
    cell can be checked against the one vector TensorFlow itself publishes for it, which uses the default tanh
    (tests/test_oracle.py::test_lnlstm_cell_reproduces_tensorflow_unit_test_constants)."""
    base = base if base is not None else "TSP/%s_cell/layer_norm_basic_lstm_cell" % cell
    z = torch.cat([x, h], dim=1) @ params[base + "/kernel"]
    i, j, f, o = torch.chunk(z, 4, dim=1)
    i = layer_norm(i, params[base + "/input/gamma"], params[base + "/input/beta"])
    j = layer_norm(j, params[base + "/transform/gamma"], params[base + "/transform/beta"])
    f = layer_norm(f, params[base + "/forget/gamma"], params[base + "/forget/beta"])
    o = layer_norm(o, params[base + "/output/gamma"], params[base + "/output/beta"])
    g = activation(j)
    new_c = c * torch.sigmoid(f + FORGET_BIAS) + torch.sigmoid(i) * g
    new_c = layer_norm(new_c, params[base + "/state/gamma"], params[base + "/state/beta"])
    new_h = activation(new_c) * torch.sigmoid(o)
    return new_h, new_c


def dense_ev(ev_uv, n_total, dtype):
    """The block-diagonal dense EV[M,N] of instance_loader.py:45,63-66 (exactly two ones per row)."""
    M = ev_uv.shape[0]
    EV = torch.zeros((M, n_total), dtype=dtype)
    idx = torch.arange(M)
    uv = torch.as_tensor(np.asarray(ev_uv), dtype=torch.long)
    EV[idx, uv[:, 0]] = 1
    EV[idx, uv[:, 1]] = 1
    return EV


def message_passing(params, ev_uv, V0, E0, time_steps, dense=False, EV=None, trace=None):
    """GraphNN.__call__ for the TSP wiring (graphnn.py:134-179, model.py:57-94).
    Both updates read the *old* states (new_states is a fresh dict, graphnn.py:143)."""
    N = V0.shape[0]
    uv = torch.as_tensor(np.asarray(ev_uv), dtype=torch.long)
    if dense and EV is None:
        EV = dense_ev(ev_uv, N, V0.dtype)
    Vh, Vc = V0, torch.zeros_like(V0)  # graphnn.py:135-138
    Eh, Ec = E0, torch.zeros_like(E0)
    for t in range(int(time_steps)):
        # V <- LSTM_V( EV^T x E_msg_V(E.h) )
        y = mlp(Eh, params, "TSP/E_msg_V")
        if dense:
            vagg = EV.t() @ y  # tf.matmul(..., adjoint_a=True)
        else:
            vagg = torch.zeros_like(Vh).index_add(0, uv[:, 0], y).index_add(0, uv[:, 1], y)
        # E <- LSTM_E( EV x V_msg_E(V.h) )
        y2 = mlp(Vh, params, "TSP/V_msg_E")
        if dense:
            eagg = EV @ y2
        else:
            eagg = y2[uv[:, 0]] + y2[uv[:, 1]]
        nVh, nVc = lnlstm_cell(vagg, Vh, Vc, params, "V")
        nEh, nEc = lnlstm_cell(eagg, Eh, Ec, params, "E")
        Vh, Vc, Eh, Ec = nVh, nVc, nEh, nEc
        if trace is not None:
            trace.append((Vh.detach().clone(), Eh.detach().clone()))
    return {"V": (Vh, Vc), "E": (Eh, Ec)}


def _rb(x):
    """Round to bf16 (nearest even) and return in the original dtype.  Straight-through for autograd: the gradient
    passes unchanged and in the working precision (a plain .to(bfloat16) round trip would also round the gradient
    flowing back through it) -- the mixed-precision convention the build's bf16 training follows."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def message_passing_bf16(params, ev_uv, V0, E0, time_steps):
    """The build's bf16-storage mode (BASELINE config 5: "bf16 embeddings with fp32 accumulate"; the reference has
    no reduced-precision path, this restates where tsp-gnn_amd rounds): embeddings h, every stored MLP activation,
    the V<-E aggregate and the projected vertex messages Zx = V_msg_E(V.h) Kx are rounded to bf16, GEMM weights are
    the variables rounded to bf16, sums accumulate in the working precision; the cell state c, LayerNorm, biases
    and gates are not rounded.  The edge cell is evaluated in the folded form (EV y) Kx = EV (y Kx)."""
    uv = torch.as_tensor(np.asarray(ev_uv), dtype=torch.long)
    Vh, Vc = _rb(V0), torch.zeros_like(V0)
    Eh, Ec = _rb(E0), torch.zeros_like(E0)
    for t in range(int(time_steps)):
        Vh, Vc, Eh, Ec = step_bf16(params, uv, Vh, Vc, Eh, Ec)
    return {"V": (Vh, Vc), "E": (Eh, Ec)}


def step_bf16(params, uv, Vh, Vc, Eh, Ec):
    """One message-passing step of message_passing_bf16 on stored (bf16-valued) h and fp32-class c: -> the next
    (Vh, Vc, Eh, Ec), h rounded for storage.  uv: long tensor [M,2]."""
    d = Vh.shape[1]

    def mlp_b(x, prefix):
        for i in range(4):
            W = _rb(params["%s_MLP_layer_%d/kernel" % (prefix, i + 1)])
            x = x @ W + params["%s_MLP_layer_%d/bias" % (prefix, i + 1)]
            x = _rb(torch.relu(x) if i < 3 else x)
        return x

    def cell_b(z, c, cell):
        base = "TSP/%s_cell/layer_norm_basic_lstm_cell" % cell
        i, j, f, o = torch.chunk(z, 4, dim=1)
        i = layer_norm(i, params[base + "/input/gamma"], params[base + "/input/beta"])
        j = layer_norm(j, params[base + "/transform/gamma"], params[base + "/transform/beta"])
        f = layer_norm(f, params[base + "/forget/gamma"], params[base + "/forget/beta"])
        o = layer_norm(o, params[base + "/output/gamma"], params[base + "/output/beta"])
        new_c = c * torch.sigmoid(f + FORGET_BIAS) + torch.sigmoid(i) * torch.relu(j)
        new_c = layer_norm(new_c, params[base + "/state/gamma"], params[base + "/state/beta"])
        return _rb(torch.relu(new_c) * torch.sigmoid(o)), new_c
    KV = _rb(params["TSP/V_cell/layer_norm_basic_lstm_cell/kernel"])
    KE = _rb(params["TSP/E_cell/layer_norm_basic_lstm_cell/kernel"])
    y = mlp_b(Eh, "TSP/E_msg_V")
    vagg = _rb(torch.zeros_like(Vh).index_add(0, uv[:, 0], y).index_add(0, uv[:, 1], y))
    zx = _rb(mlp_b(Vh, "TSP/V_msg_E") @ KE[:d])
    nVh, nVc = cell_b(torch.cat([vagg, Vh], dim=1) @ KV, Vc, "V")
    nEh, nEc = cell_b(zx[uv[:, 0]] + zx[uv[:, 1]] + Eh @ KE[d:], Ec, "E")
    return nVh, nVc, nEh, nEc


def forward(params, batch, time_steps, dense=False, trace=None, bf16=False):
    """build_network forward (model.py:18-157) on a packed batch.

    batch: dict with ev_uv int[M,2], W[M], C[M], route_exists[B], n_vertices[B], n_edges[B].
    bf16=True: the message passing in the build's bf16-storage mode (message_passing_bf16).
    """
    V0, E0 = initial_embeddings(params, batch)
    if bf16:
        last = message_passing_bf16(params, batch["ev_uv"], V0, E0, time_steps)
    else:
        last = message_passing(params, batch["ev_uv"], V0, E0, time_steps, dense=dense, trace=trace)
    out = vote_head(params, batch, last["E"][0])
    out["last_states"] = last
    return out


def initial_embeddings(params, batch):
    """(V0, E0) of model.py:33-51: V_init / sqrt(d) tiled over the vertices, E_init_MLP([W, C])."""
    some = params["V_init"]
    dtype = some.dtype
    d = some.shape[1]
    W = torch.as_tensor(np.asarray(batch["W"]), dtype=dtype).reshape(-1, 1)
    C = torch.as_tensor(np.asarray(batch["C"]), dtype=dtype).reshape(-1, 1)
    N = int(np.asarray(batch["n_vertices"]).astype(np.int64).sum())
    # model.py:43
    E0 = mlp(torch.cat([W, C], dim=1), params, "E_init_MLP")
    # model.py:48-51
    V0 = (params["V_init"] / math.sqrt(float(d))).repeat(N, 1)
    return V0, E0


def vote_head(params, batch, E_n):
    """model.py:107-157 from the final edge embeddings: votes, per-problem mean, predictions, metrics, loss."""
    dtype = E_n.dtype
    labels = torch.as_tensor(np.asarray(batch["route_exists"]), dtype=dtype)
    n_edges = np.asarray(batch["n_edges"]).astype(np.int64)
    # model.py:128
    E_vote = mlp(E_n, params, "E_vote").reshape(-1)
    # model.py:134-145: mean of each problem's edge-vote segment
    offs = np.concatenate([[0], np.cumsum(n_edges)])
    logits = torch.stack([E_vote[offs[i]:offs[i + 1]].mean() for i in range(len(n_edges))])
    pred = torch.sigmoid(logits)
    # tf.round = half-to-even; torch.round too
    rp = torch.round(pred)
    eq = (labels == rp).to(dtype)
    ne = (labels != rp).to(dtype)
    out = {
        "E_vote": E_vote,
        "logits": logits,
        "predictions": pred,
        # formulas kept verbatim, including the FP/FN mislabelling (model.py:150-153)
        "TP": (labels * eq).sum(),
        "FP": (labels * ne).sum(),
        "TN": ((1 - labels) * eq).sum(),
        "FN": ((1 - labels) * ne).sum(),
        "acc": eq.mean(),
    }
    out["loss"] = sigmoid_cross_entropy_with_logits(logits, labels).mean()
    return out


def sigmoid_cross_entropy_with_logits(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits (model.py:147): z * -log(sigmoid(x)) + (1 - z) * -log(1 - sigmoid(x)) in the
    form TensorFlow documents and evaluates, max(x, 0) - x * z + log(1 + exp(-|x|))."""
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))


def loss_and_grads(params_np, batch, time_steps, dtype=torch.float64, dense=False, bf16=False):
    """tf.gradients(loss + 1e-10 * sum l2_loss(var)) (model.py:163-166), unclipped.  bf16=True: of the bf16-storage
    forward (message_passing_bf16), roundings passed straight through, gradients w.r.t. the unrounded variables."""
    params = to_torch(params_np, dtype=dtype, requires_grad=True)
    out = forward(params, batch, time_steps, dense=dense, bf16=bf16)
    vars_cost = sum((p ** 2).sum() / 2 for p in params.values())
    total = out["loss"] + L2NORM_SCALING * vars_cost
    grads = torch.autograd.grad(total, list(params.values()))
    g = OrderedDict((k, gi.detach().numpy().copy()) for k, gi in zip(params.keys(), grads))
    return out, g


def clip_by_global_norm(grads, clip_norm=CLIP_NORM):
    """tf.clip_by_global_norm: g * clip / max(global_norm, clip)."""
    gn = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values()))
    scale = clip_norm / max(gn, clip_norm)
    return OrderedDict((k, g * scale) for k, g in grads.items()), gn


def adam_step(params, grads, m, v, step, lr=LEARNING_RATE):
    """tf.train.AdamOptimizer._apply_dense: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; p -= lr_t * m / (sqrt(v) + eps).  ``step`` is 1-based."""
    lr_t = lr * math.sqrt(1 - ADAM_B2 ** step) / (1 - ADAM_B1 ** step)
    new_p, new_m, new_v = OrderedDict(), OrderedDict(), OrderedDict()
    for k in params:
        g = grads[k]
        new_m[k] = ADAM_B1 * m[k] + (1 - ADAM_B1) * g
        new_v[k] = ADAM_B2 * v[k] + (1 - ADAM_B2) * g * g
        new_p[k] = params[k] - lr_t * new_m[k] / (np.sqrt(new_v[k]) + ADAM_EPS)
    return new_p, new_m, new_v


def train_step(params_np, batch, time_steps, m, v, step, dtype=torch.float64):
    """One ``sess.run(train_step)`` (model.py:160-167)."""
    out, g = loss_and_grads(params_np, batch, time_steps, dtype=dtype)
    g, gn = clip_by_global_norm(g)
    p, m, v = adam_step(params_np, g, m, v, step)
    return out, p, m, v, gn


def time_dense_forward(d, batch, time_steps, threads, warm=1, iters=1, seed=0):
    """cpu_baseline leg of bench.py: fp32, dense EV matmuls both directions, unfused Dense
    layers and LayerNorm-LSTM -- the FLOPs and memory traffic of the TF CPU graph
    (SURVEY.md §8d M5).  Returns (median seconds per forward, threads used)."""
    import time

    torch.set_num_threads(int(threads))
    params = to_torch(P.init_params(d, seed=seed), dtype=torch.float32)
    N = int(np.asarray(batch["n_vertices"]).sum())
    EV = dense_ev(batch["ev_uv"], N, torch.float32)
    times = []
    with torch.no_grad():
        W = torch.as_tensor(np.asarray(batch["W"]), dtype=torch.float32).reshape(-1, 1)
        C = torch.as_tensor(np.asarray(batch["C"]), dtype=torch.float32).reshape(-1, 1)
        E0 = mlp(torch.cat([W, C], dim=1), params, "E_init_MLP")
        V0 = (params["V_init"] / math.sqrt(float(d))).repeat(N, 1)
        for it in range(warm + iters):
            t0 = time.perf_counter()
            message_passing(params, batch["ev_uv"], V0, E0, time_steps, dense=True, EV=EV)
            dt = time.perf_counter() - t0
            if it >= warm:
                times.append(dt)
    return float(np.median(times)), torch.get_num_threads()
