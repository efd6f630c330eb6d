"""Full-size parity against committed float64 anchors (tests/golden/anchor_*.npz, written by oracle/gen_golden.py in
the build container): BASELINE.json's C1 (n=20, B=32, T=8), C2 -- the headline configuration (n=40, B=128, T=32) --
and C4 (ragged n in 20..80, B=512, T=2), every GEMM arithmetic of the forward, at the 1e-5 relative tolerance
north_star states.  The oracle does not run here: the test regenerates the inputs, checks their fingerprints, runs
the HIP forward and compares predictions, loss, and the column sums / 512 sampled rows of E.h, E.c, V.h, V.c."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("name", ["c1", "c2", "c4"])
def test_full_size_forward_matches_float64_anchor(cuda_device, name, gemm):
    import tspgnn
    from oracle.anchors import anchor_inputs, anchor_rows
    z = np.load(os.path.join(GOLDEN, "anchor_%s.npz" % name))
    batch, params, T, finger = anchor_inputs(name)
    assert T == int(z["T"]) and np.array_equal(finger, z["fingerprint"]), "inputs differ from the anchor's"
    EV, W, C, route_exists, n_vertices, n_edges = batch
    model = tspgnn.build_network(64)
    model["gnn"].gemm = gemm
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
            model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, loss, acc, last = sess.run([model["predictions"], model["loss"], model["acc"], model["last_states"]],
                                     feed_dict=feed)
    worst = {}
    worst["predictions"] = np.abs(pred - z["predictions"]).max() / np.abs(z["predictions"]).max()
    assert abs(float(loss) - float(z["loss"])) < REL_TOL and abs(float(acc) - float(z["acc"])) < 1e-6
    for var in ("E", "V"):
        for part in ("h", "c"):
            a = np.asarray(getattr(last[var], part), dtype=np.float64)
            scale = float(z["%s%s_absmax" % (var, part)])
            rows = anchor_rows(a.shape[0])
            worst["%s.%s rows" % (var, part)] = np.abs(a[rows] - z["%s%s_rows" % (var, part)]).max() / scale
            # column sums over all rows: every row takes part; a sum of R rows may carry R * tol * scale
            worst["%s.%s colsum" % (var, part)] = np.abs(a.sum(0) - z["%s%s_colsum" % (var, part)]).max() / (scale * a.shape[0])
    print("anchor %s gemm=%s: %s" % (name, model["gnn"].gemm, "  ".join("%s %.1e" % kv for kv in worst.items())))
    for k, v in worst.items():
        assert v < REL_TOL, (k, v)
