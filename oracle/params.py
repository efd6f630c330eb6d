"""Parameter inventory of the TSP-GNN hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Names follow the TF-1.x variable names the reference graph would create
(SURVEY.md §8a R1/R6/R8/R9):

* ``E_init_MLP``  : model.py:33-43  (Dense 2 -> d/8 -> d/4 -> d/2 -> d, zero biases)
* ``V_init``      : model.py:47     (random_normal (1,d))
* ``TSP/<msg>``   : graphnn.py:114-125 (Mlp([d]*3, output_size=d); biases use the
                    *weight* initialiser -- the xavier quirk at graphnn.py:121)
* ``TSP/<v>_cell``: graphnn.py:107-112 (LayerNormBasicLSTMCell(d, activation=relu))
* ``E_vote``      : model.py:107-115 (Dense d -> d -> d -> d -> 1, zero biases)

The order of ``param_names`` is the canonical flat order used for the gradient
bucket (all-reduce), the global-norm clip and Adam.
"""
import math
from collections import OrderedDict

import numpy as np

LN_GATES = ("input", "transform", "forget", "output", "state")


def mlp_layer_sizes(d):
    """(in,out) per Dense layer of each MLP; int() mirrors tf.layers.Dense(int(units))
    for the float sizes d/8, d/4, d/2 at model.py:34."""
    return {
        "E_init_MLP": [(2, int(d / 8)), (int(d / 8), int(d / 4)), (int(d / 4), int(d / 2)), (int(d / 2), d)],
        "TSP/V_msg_E": [(d, d)] * 4,
        "TSP/E_msg_V": [(d, d)] * 4,
        "E_vote": [(d, d), (d, d), (d, d), (d, 1)],
    }


def param_shapes(d):
    """OrderedDict name -> shape, in canonical flat order."""
    shapes = OrderedDict()
    sizes = mlp_layer_sizes(d)
    for i, (a, b) in enumerate(sizes["E_init_MLP"]):
        shapes["E_init_MLP_MLP_layer_%d/kernel" % (i + 1)] = (a, b)
        shapes["E_init_MLP_MLP_layer_%d/bias" % (i + 1)] = (b,)
    shapes["V_init"] = (1, d)
    for msg in ("TSP/V_msg_E", "TSP/E_msg_V"):
        for i, (a, b) in enumerate(sizes[msg]):
            shapes["%s_MLP_layer_%d/kernel" % (msg, i + 1)] = (a, b)
            shapes["%s_MLP_layer_%d/bias" % (msg, i + 1)] = (b,)
    for v in ("V", "E"):
        base = "TSP/%s_cell/layer_norm_basic_lstm_cell" % v
        shapes[base + "/kernel"] = (2 * d, 4 * d)
        for g in LN_GATES:
            shapes["%s/%s/gamma" % (base, g)] = (d,)
            shapes["%s/%s/beta" % (base, g)] = (d,)
    for i, (a, b) in enumerate(sizes["E_vote"]):
        shapes["E_vote_MLP_layer_%d/kernel" % (i + 1)] = (a, b)
        shapes["E_vote_MLP_layer_%d/bias" % (i + 1)] = (b,)
    return shapes


def param_names(d):
    return list(param_shapes(d).keys())


def n_params(d):
    return sum(int(np.prod(s)) for s in param_shapes(d).values())


def _xavier(rng, shape):
    # tf.contrib.layers.xavier_initializer (uniform): limit = sqrt(6/(fan_in+fan_out));
    # for a 1-D shape TF takes fan_in = fan_out = shape[0].
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def init_params(d, seed=0, perturb=False, round_f32=True):
    """float64 numpy parameters with the reference's initialisers.  ``round_f32`` (default) rounds
    every value to the nearest float32 so that the fp32 device copy and the float64 oracle hold
    IDENTICAL weights (otherwise the 6e-8 rounding of the upload is amplified by the recurrence).

    ``perturb=True`` additionally randomises every bias / LayerNorm gain / shift so
    that parity tests exercise the terms that are 0 or 1 at initialisation.
    Weight-*stream* parity with TF's RNG is not attempted.
    """
    rng = np.random.RandomState(seed)
    out = OrderedDict()
    for name, shape in param_shapes(d).items():
        if name == "V_init":
            p = rng.standard_normal(shape)
        elif name.endswith("/kernel"):
            p = _xavier(rng, shape)
        elif name.endswith("/bias"):
            if name.startswith("TSP/"):
                p = _xavier(rng, shape)  # graphnn.py:121 quirk
            else:
                p = np.zeros(shape)
            if perturb:
                p = p + 0.1 * rng.standard_normal(shape)
        elif name.endswith("/gamma"):
            p = np.ones(shape)
            if perturb:
                p = p + 0.2 * rng.standard_normal(shape)
        elif name.endswith("/beta"):
            p = np.zeros(shape)
            if perturb:
                p = p + 0.2 * rng.standard_normal(shape)
        else:
            raise KeyError(name)
        p = np.ascontiguousarray(p, dtype=np.float64)
        if round_f32:
            p = p.astype(np.float32).astype(np.float64)
        out[name] = p
    return out
