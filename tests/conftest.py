import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pack(name, seed=0):
    """A tests/golden/pack_*.npz fixture (outputs of the reference's create_batch) as a dict with
    the instances re-assembled and ev_uv [M,2]."""
    z = np.load(os.path.join(GOLDEN, "pack_%s_seed%d.npz" % (name, seed)))
    n = int(z["n_instances"])
    M = int(z["ev_shape"][0])
    assert np.array_equal(z["ev_rows"], np.repeat(np.arange(M), 2))
    out = {k: z[k] for k in ("W", "C", "route_exists", "n_vertices", "n_edges", "ev_shape")}
    out["ev_uv"] = z["ev_cols"].reshape(M, 2)
    out["dev"] = float(z["dev"])
    tc = float(z["target_cost"])
    out["target_cost"] = None if np.isnan(tc) else tc
    out["instances"] = [(z["Ma_%d" % i].astype(int), z["Mw_%d" % i], [int(x) for x in z["route_%d" % i]])
                        for i in range(n)]
    if "EV_dense" in z.files:
        out["EV_dense"] = z["EV_dense"]
    return out


def batch_from_tuple(t):
    """create_batch 6-tuple (with SparseEV) -> oracle batch dict."""
    EV, W, C, route_exists, n_vertices, n_edges = t
    return {"ev_uv": EV.uv, "W": W, "C": C, "route_exists": route_exists, "n_vertices": n_vertices,
            "n_edges": n_edges}


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a HIP device")
    return torch.device("cuda:0")


def rel_err(a, b):
    """max |a-b| normalised by the scale of the reference tensor b."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / scale


def h2_zx_pack(Zx, scale):
    """Row-major projected messages [N, 4d] -> the f16x2 kernels' format (include/tspgnn.h): times 2^s, rows padded to a
    multiple of 16, blocked by 16 rows: float4 (columns 16t + 4g .. +3) of row v at (((v/16)*d/4 + t)*4 + g)*64 + (v%16)*4."""
    Zx = np.asarray(Zx, dtype=np.float32)
    n, w = Zx.shape
    pad = (n + 15) // 16 * 16
    flat = np.zeros((pad, w), dtype=np.float32)
    flat[:n] = scale * Zx
    return np.ascontiguousarray(flat.reshape(pad // 16, 16, w // 16, 4, 4).transpose(0, 2, 3, 1, 4)).reshape(pad, w)


def h2_zx_unpack(blocked, n, scale):
    """Inverse of h2_zx_pack: -> row-major [n, 4d], unscaled."""
    b = np.asarray(blocked, dtype=np.float32)
    pad, w = b.shape
    flat = b.reshape(pad // 16, w // 16, 4, 16, 4).transpose(0, 3, 1, 2, 4).reshape(pad, w)
    return flat[:n] / scale
