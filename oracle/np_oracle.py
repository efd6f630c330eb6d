"""Independent NumPy restatement of the hot path, forward only (TEST INFRASTRUCTURE).

PARITY UNPINNED (see torch_oracle.py header).  Written separately from torch_oracle.py so
the two restatements cross-check each other (SURVEY.md §4 test plan item 1).  Every function
cites the reference line it follows.  Works in float64 (truth) or float32 (same summation
widths as the reference's fp32 graph).
"""
import numpy as np

LN_EPS = 1e-12
FORGET_BIAS = 1.0


def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def dense(x, W, b, act):
    """tf.layers.Dense as built at mlp.py:39-52."""
    y = x @ W + b
    return relu(y) if act else y


def mlp(x, layers, acts):
    """mlp.py:57-63.  layers: [(W,b)...], acts: [bool...]."""
    for (W, b), a in zip(layers, acts):
        x = dense(x, W, b, a)
    return x


def layer_norm(x, gamma, beta):
    """tf.contrib.layers.layer_norm (biased variance, eps=1e-12)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    inv = gamma / np.sqrt(var + x.dtype.type(LN_EPS))
    return x * inv + (beta - mean * inv)


def lnlstm(x, h, c, K, ln, z0=None, activation=relu):
    """LayerNormBasicLSTMCell.call as used at graphnn.py:168-170.
    ln: dict gate -> (gamma, beta) for gates input/transform/forget/output/state.
    z0: optional pre-activation offset (the part of [x,h] K a caller has already formed, e.g. an
    aggregation pushed through Kx); None for the plain cell.
    activation: the cell's constructor argument (relu on the hot path, graphnn.py:110; tanh is TF's default and
    what TensorFlow's own unit test of the cell uses)."""
    d = h.shape[1]
    z = np.concatenate([x, h], axis=1) @ K
    if z0 is not None:
        z = z + z0
    i, j, f, o = z[:, :d], z[:, d:2 * d], z[:, 2 * d:3 * d], z[:, 3 * d:]
    i = layer_norm(i, *ln["input"])
    j = layer_norm(j, *ln["transform"])
    f = layer_norm(f, *ln["forget"])
    o = layer_norm(o, *ln["output"])
    new_c = c * sigmoid(f + x.dtype.type(FORGET_BIAS)) + sigmoid(i) * activation(j)
    new_c = layer_norm(new_c, *ln["state"])
    new_h = activation(new_c) * sigmoid(o)
    return new_h, new_c


def gather2_sum(ev_uv, X):
    """Row e of EV @ X when EV has exactly two ones per row (graphnn.py:156-160,
    instance_loader.py:64-65): X[u_e] + X[v_e]."""
    return X[ev_uv[:, 0]] + X[ev_uv[:, 1]]


def rowsum_by_vertex(ev_uv, X, n_total):
    """Row v of EV^T @ X: sum of X[e] over the edges incident to v (graphnn.py:156-160,
    adjoint_a=True)."""
    out = np.zeros((n_total, X.shape[1]), dtype=X.dtype)
    np.add.at(out, ev_uv[:, 0], X)
    np.add.at(out, ev_uv[:, 1], X)
    return out


def csr_by_vertex(ev_uv, n_total):
    """CSR of EV^T: for each vertex the ascending list of incident edge ids."""
    M = ev_uv.shape[0]
    flat_v = ev_uv.reshape(-1)
    flat_e = np.repeat(np.arange(M, dtype=np.int64), 2)
    order = np.lexsort((flat_e, flat_v))
    eid = flat_e[order].astype(np.int32)
    counts = np.bincount(flat_v, minlength=n_total)
    rowptr = np.zeros(n_total + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(counts)
    return rowptr, eid


def csr_rowsum(rowptr, eid, X):
    out = np.zeros((len(rowptr) - 1, X.shape[1]), dtype=X.dtype)
    for v in range(len(rowptr) - 1):
        out[v] = X[eid[rowptr[v]:rowptr[v + 1]]].sum(axis=0)
    return out


def _mlp_params(params, prefix):
    return [(params["%s_MLP_layer_%d/kernel" % (prefix, i)], params["%s_MLP_layer_%d/bias" % (prefix, i)])
            for i in range(1, 5)]


def _ln_params(params, cell):
    base = "TSP/%s_cell/layer_norm_basic_lstm_cell" % cell
    return params[base + "/kernel"], {g: (params["%s/%s/gamma" % (base, g)], params["%s/%s/beta" % (base, g)])
                                      for g in ("input", "transform", "forget", "output", "state")}


def forward(params, batch, time_steps, dtype=np.float64):
    """build_network forward, index variant (model.py:18-157, graphnn.py:134-179)."""
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    acts = [True, True, True, False]
    ev_uv = np.asarray(batch["ev_uv"]).astype(np.int64)
    W = np.asarray(batch["W"], dtype=dtype).reshape(-1, 1)
    C = np.asarray(batch["C"], dtype=dtype).reshape(-1, 1)
    n_vertices = np.asarray(batch["n_vertices"]).astype(np.int64)
    n_edges = np.asarray(batch["n_edges"]).astype(np.int64)
    labels = np.asarray(batch["route_exists"], dtype=dtype)
    N = int(n_vertices.sum())
    d = p["V_init"].shape[1]
    Eh = mlp(np.concatenate([W, C], axis=1), _mlp_params(p, "E_init_MLP"), acts)  # model.py:43
    Vh = np.tile(p["V_init"] / dtype(np.sqrt(dtype(d))), (N, 1))  # model.py:48-51
    Vc, Ec = np.zeros_like(Vh), np.zeros_like(Eh)  # graphnn.py:137
    KV, lnV = _ln_params(p, "V")
    KE, lnE = _ln_params(p, "E")
    for _ in range(int(time_steps)):
        vagg = rowsum_by_vertex(ev_uv, mlp(Eh, _mlp_params(p, "TSP/E_msg_V"), acts), N)
        eagg = gather2_sum(ev_uv, mlp(Vh, _mlp_params(p, "TSP/V_msg_E"), acts))
        nV = lnlstm(vagg, Vh, Vc, KV, lnV)
        nE = lnlstm(eagg, Eh, Ec, KE, lnE)
        (Vh, Vc), (Eh, Ec) = nV, nE
    vote = mlp(Eh, _mlp_params(p, "E_vote"), acts).reshape(-1)  # model.py:128
    offs = np.concatenate([[0], np.cumsum(n_edges)])
    logits = np.array([vote[offs[i]:offs[i + 1]].mean() for i in range(len(n_edges))], dtype=dtype)
    pred = sigmoid(logits)
    loss = (np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))).mean()
    rp = np.round(pred)  # half-to-even like tf.round
    eq = (labels == rp).astype(dtype)
    return {"V": (Vh, Vc), "E": (Eh, Ec), "E_vote": vote, "logits": logits, "predictions": pred,
            "loss": loss, "acc": eq.mean(), "TP": (labels * eq).sum(), "FP": (labels * (1 - eq)).sum(),
            "TN": ((1 - labels) * eq).sum(), "FN": ((1 - labels) * (1 - eq)).sum()}
