"""Inputs of the full-size float64 anchors (TEST INFRASTRUCTURE): which batch and which weights
``tests/golden/anchor_*.npz`` were computed on, and which rows of the state arrays they keep.  Shared by the
generator (oracle/gen_golden.py, build container) and the GPU test that replays them (tests/test_gpu_anchors.py);
touches nothing outside the repository."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ANCHORS = {
    # name: (sizes, T)  -- BASELINE.json configs[0], [1], [3]
    "c1": (lambda: [20] * 32, 8),
    "c2": (lambda: [40] * 128, 32),
    "c4": (lambda: [int(x) for x in np.random.RandomState(0).randint(20, 81, size=512)], 32),   # (T = 2, then 8, until round 6: now the depth `bench.py --workload c4` runs)
}
ANCHOR_ROWS = 512


# BASELINE config 5 at one GPU's graph size and depth (n=200, d=128, T=64, bf16 embeddings / fp32 accumulate) on 4 of the 32
# graphs of a shard: oracle = torch_oracle.forward(..., bf16=True), the float64 restatement with the build's rounding points
# "c5shard": ALL 32 graphs of one GPU's shard of config 5 (M = 636 800 edges, the size `bench.py --workload c5` runs) at T = 8
# "c5full" (round 6): the whole shard at config 5's own depth, T = 64 -- exactly what `bench.py --workload c5` runs (40 minutes of
# float64 on the build host)
BF16_ANCHORS = {"c5": (lambda: [200] * 4, 128, 64), "c5shard": (lambda: [200] * 32, 128, 8),   # (c5shard: T = 2 until round 6)
                "c5full": (lambda: [200] * 32, 128, 64)}


def bf16_anchor_inputs(name):
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    import tspgnn
    from oracle import params as P
    import zlib
    sizes, d, T = BF16_ANCHORS[name]
    batch = tspgnn.synthetic_batch(sizes(), seed=7)
    params = P.init_params(d, seed=3)
    EV, W, C = batch[0], batch[1], batch[2]
    finger = np.array([float(EV.shape[0]), float(EV.shape[1]), float(zlib.crc32(np.ascontiguousarray(EV.uv).view(np.uint8).reshape(-1))),
                       float(np.sum(W, dtype=np.float64)), float(np.sum(C, dtype=np.float64)),
                       float(sum(np.sum(np.asarray(v, dtype=np.float64)) for v in params.values())),
                       float(sum(np.sum(np.abs(np.asarray(v, dtype=np.float64))) for v in params.values()))])
    return batch, params, d, T, finger


def anchor_inputs(name):
    """The batch and the weights an anchor was computed on (also called by tests/test_gpu_anchors.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    import tspgnn
    from oracle import params as P
    import zlib
    sizes, T = ANCHORS[name]
    batch = tspgnn.synthetic_batch(sizes(), seed=1234)
    params = P.init_params(64, seed=0)
    EV, W, C = batch[0], batch[1], batch[2]
    finger = np.array([float(EV.shape[0]), float(EV.shape[1]), float(zlib.crc32(np.ascontiguousarray(EV.uv).view(np.uint8).reshape(-1))),
                       float(np.sum(W, dtype=np.float64)), float(np.sum(C, dtype=np.float64)),
                       float(sum(np.sum(np.asarray(v, dtype=np.float64)) for v in params.values())),
                       float(sum(np.sum(np.abs(np.asarray(v, dtype=np.float64))) for v in params.values()))])
    return batch, params, T, finger


def anchor_rows(n_rows):
    return np.unique(np.linspace(0, n_rows - 1, ANCHOR_ROWS).astype(np.int64))


# Gradient anchors (round 5): tf.gradients(loss + 1e-10 * l2) (model.py:163-166, unclipped) of the float64 oracle at FULL
# size -- C2 at T = 2, C1 at its own T = 8 -- for the same batches and weights as the forward anchors, but perturbed
# LayerNorm parameters / biases (init_params(perturb=True): with gamma = 1, beta = 0 and zero biases whole gradient
# blocks would sit at their symmetric points).  Kept per variable: its 2-norm, its largest entry, 64 sampled entries, and
# what the op-for-op float32 autograd restatement loses on the same variable (the error budget of any fp32 backward).
# Round 6: "c2t8" = C2 four times as deep (T = 8: ~16 GB of float64 autograd in the build container), and with every anchor
# the CONDITIONING of each variable's gradient: the largest change of the float64 gradient (sampled entries, 2-norm) when
# every variable entry moves by ONE fp32 ulp (w * (1 +- 2^-23), random signs, GRAD_PERTURBATIONS draws).  An fp32-class
# backward cannot be asked for less than that -- it computes the exact gradient of a network whose weights are a rounding
# away -- and tests/test_gpu_anchors.py asks every arithmetic for 2x that, with no arithmetic-specific allowance.
GRAD_ANCHORS = {"c2": (lambda: [40] * 128, 2), "c1": (lambda: [20] * 32, 8), "c2t8": (lambda: [40] * 128, 8),
                "c4": (lambda: [int(x) for x in np.random.RandomState(0).randint(20, 81, size=512)], 2)}   # (ragged, M = 695 849)
GRAD_SAMPLES = 64
GRAD_PERTURBATIONS = 3


def ulp_perturbed(params, draw):
    """Every entry of every variable moved by one fp32 ulp (relative 2^-23), signs from a generator seeded by ``draw``."""
    rng = np.random.RandomState(7919 + draw)
    out = {}
    for k, v in params.items():
        a = np.asarray(v, dtype=np.float64)
        out[k] = a * (1.0 + (rng.randint(0, 2, size=a.shape) * 2 - 1) * 2.0 ** -23)
    return out

# "Far from init" weights (round 5): the float64 oracle trained for 2 000 Adam steps at lr 1e-3 (model.py:160-167 with a
# larger step) on synthetic batches -- LayerNorm gains, biases and kernels with the statistics of a trained network, not
# of the initialisers.  tests/golden/trained_d64.npz holds the weights; the parity tests run the oracle next to the HIP path.
TRAINED = {"d": 64, "steps": 2000, "lr": 1e-3, "sizes": [12] * 16, "T": 8, "n_batches": 8}


def grad_anchor_inputs(name):
    sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
    import tspgnn
    from oracle import params as P
    import zlib
    sizes, T = GRAD_ANCHORS[name]
    batch = tspgnn.synthetic_batch(sizes(), seed=1234)
    params = P.init_params(64, seed=0, perturb=True)
    EV, W, C = batch[0], batch[1], batch[2]
    finger = np.array([float(EV.shape[0]), float(EV.shape[1]), float(zlib.crc32(np.ascontiguousarray(EV.uv).view(np.uint8).reshape(-1))),
                       float(np.sum(W, dtype=np.float64)), float(np.sum(C, dtype=np.float64)),
                       float(sum(np.sum(np.asarray(v, dtype=np.float64)) for v in params.values())),
                       float(sum(np.sum(np.abs(np.asarray(v, dtype=np.float64))) for v in params.values()))])
    return batch, params, T, finger


def grad_sample_index(name, size):
    """Flat indices of the sampled entries of variable ``name`` (deterministic in the name and the size)."""
    import zlib
    rng = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return np.sort(rng.choice(size, size=min(GRAD_SAMPLES, size), replace=False))
