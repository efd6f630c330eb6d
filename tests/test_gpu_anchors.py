"""Full-size parity against committed float64 anchors (tests/golden/anchor_*.npz, written by oracle/gen_golden.py in
the build container): BASELINE.json's C1 (n=20, B=32, T=8), C2 -- the headline configuration (n=40, B=128, T=32) --
and C4 (ragged n in 20..80, B=512, T=32: the depth `bench.py --workload c4` runs), every GEMM arithmetic of the forward, at the 1e-5 relative tolerance
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


@pytest.mark.parametrize("name", ["c5", "c5shard", "c5full"])
def test_bf16_storage_at_config5_depth(cuda_device, name):
    """BASELINE config 5's mode (bf16 embeddings, fp32 accumulate) at ITS graph size, width and depth -- n=200, d=128,
    T=64 -- on 4 graphs of a shard (M = 79 600 edges; "c5"), and on ALL 32 graphs of one GPU's shard -- M = 636 800 edges,
    the size `bench.py --workload c5` runs -- at T = 8 ("c5shard") and at config 5's own T = 64 ("c5full": exactly the benchmark's
    workload, 40 minutes of float64 on the build host), against committed outputs of the oracle that rounds to
    bf16 at the same points (oracle/torch_oracle.message_passing_bf16; minutes of float64 on the build host: anchors).
    Two kinds of bars: the bulk of the values (rms, relative to the tensor's largest entry) must agree to a small
    fraction of a bf16 ulp; single entries may land on the other side of a rounding boundary and then differ by whole
    ulps at the top of the range (2^-8 = 3.9e-3 each), so the max bar is a few ulps.  The fp32 semantics (plain float64
    oracle) are an approximation target: bf16 storage over 64 recurrent steps stays within 2e-2 of them."""
    import torch
    import tspgnn
    from oracle.anchors import anchor_rows, bf16_anchor_inputs
    z = np.load(os.path.join(GOLDEN, "anchor_bf16_%s.npz" % name))
    batch, params, d, T, finger = bf16_anchor_inputs(name)
    assert T == int(z["T"]) and d == int(z["d"]) and np.array_equal(finger, z["fingerprint"]), "inputs differ from the anchor's"
    EV, W, C, route_exists, n_vertices, n_edges = batch
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
            model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    pred, loss, last = sess.run([model["predictions"], model["loss"], model["last_states"]], feed_dict=feed)

    def errs(a, ref, scale):
        e = np.abs(np.asarray(a, dtype=np.float64) - ref) / scale
        return float(e.max()), float(np.sqrt((e ** 2).mean()))

    report = []
    for var in ("E", "V"):
        for part in ("h", "c"):
            a = np.asarray(getattr(last[var], part), dtype=np.float64)
            rows = anchor_rows(a.shape[0])
            mx, rms = errs(a[rows], z["bf16_%s%s_rows" % (var, part)], float(z["bf16_%s%s_absmax" % (var, part)]))
            fmx, frms = errs(a[rows], z["f64_%s%s_rows" % (var, part)], float(z["f64_%s%s_absmax" % (var, part)]))
            report.append("%s.%s max %.1e rms %.1e (fp32 semantics: %.1e / %.1e)" % (var, part, mx, rms, fmx, frms))
            assert mx < 2e-2 and rms < 1.5e-3, (var, part, mx, rms)
            assert fmx < 4e-2 and frms < 6e-3, (var, part, fmx, frms)
    e_pred = np.abs(pred - z["bf16_predictions"]).max() / np.abs(z["bf16_predictions"]).max()
    f_pred = np.abs(pred - z["f64_predictions"]).max() / np.abs(z["f64_predictions"]).max()
    print("bf16 @ config-5 %s: pred %.1e (fp32 semantics %.1e)  loss diff %.1e  %s"
          % (name, e_pred, f_pred, abs(float(loss) - float(z["bf16_loss"])), "  ".join(report)))
    assert e_pred < 5e-4 and abs(float(loss) - float(z["bf16_loss"])) < 2e-4
    assert f_pred < 1e-2


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("name", ["c2", "c2t8", "c1", "c4"])
def test_full_size_gradients_match_float64_anchor(cuda_device, name, gemm):
    """The training step's gradients at FULL size -- C2 (M = 99 840 edges) at T = 2 and at T = 8, C1 at its own T = 8, C4 (512
    ragged instances, M = 695 849 edges) at T = 2 --
    against committed float64 autograd gradients (tests/golden/anchor_grad_*.npz, oracle/gen_golden.py grads): per variable
    the 2-norm and 64 sampled entries.  ONE bar for every arithmetic (no arithmetic-specific slack), per
    variable, relative to max(its largest entry, 1e-3 of the largest gradient entry overall):
        max(1e-5,  2 x what the op-for-op float32 autograd restatement loses on that variable  (forward error of fp32
                   arithmetic on exact operands: sums of 10^5 near-cancelling rows),
                   2 x how far the variable's float64 gradient moves when every weight moves by ONE fp32 ulp  (the
                   anchor's `ulp_spread`: backward error -- an fp32-class kernel computes the exact gradient of a network
                   a rounding away; at C2 the LayerNorm shifts of the vertex cell move 8e-4 under that perturbation, which is
                   where the f16x2 path -- whose weight packing rounds to 22 bits -- sits, see profiles/r06_grad_anchor_report.txt)).
    The variable's 2-norm (the one check that sees every entry) gets the same bar built from the norm's own two figures.
    """
    import torch
    import tspgnn
    from oracle.anchors import grad_anchor_inputs, grad_sample_index
    z = np.load(os.path.join(GOLDEN, "anchor_grad_%s.npz" % name))
    batch, params, T, finger = grad_anchor_inputs(name)
    assert T == int(z["T"]) and np.array_equal(finger, z["fingerprint"]), "inputs differ from the anchor's"
    EV, W, C, route_exists, n_vertices, n_edges = batch
    model = tspgnn.build_network(64)
    model["gnn"].gemm = gemm
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: route_exists,
            model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    out = sess.loss_and_grads(feed)
    torch.cuda.synchronize()
    g = model.store.grad_dict()
    assert abs(float(out["stats"][0].item()) - float(z["loss"])) < REL_TOL
    gscale = float(z["grad_absmax"])
    worst = (0.0, None)
    for k in g:
        # the oracle's gradients include the L2 term 1e-10 * w (model.py:163-166); the HIP backward adds it in the optimiser
        ref = z["sample:" + k] - 1e-10 * np.asarray(params[k], dtype=np.float64).reshape(-1)[grad_sample_index(k, params[k].size)]
        got = np.asarray(g[k], dtype=np.float64).reshape(-1)[grad_sample_index(k, params[k].size)]
        scale = max(float(z["absmax:" + k]), 1e-3 * gscale)
        bar = max(1e-5, 2.0 * float(z["err32:" + k]) / scale, 2.0 * float(z["ulp_spread:" + k]) / scale)
        err = float(np.abs(got - ref).max()) / scale
        nscale = max(float(z["norm:" + k]), 1e-3 * gscale)
        nbar = max(1e-5, 2.0 * float(z["err32_norm:" + k]) / nscale, 2.0 * float(z["ulp_spread_norm:" + k]) / nscale)
        whole = np.asarray(g[k], dtype=np.float64) + 1e-10 * np.asarray(params[k], dtype=np.float64)   # (+ the L2 term, as above)
        nerr = abs(float(np.sqrt((whole ** 2).sum())) - float(z["norm:" + k])) / nscale
        if err / bar > worst[0]:
            worst = (err / bar, "%s: %.2e against a bar of %.2e" % (k, err, bar))
        assert err < bar, (k, err, bar)
        assert nerr < nbar, (k, "norm", nerr, nbar)
    print("gradient anchor %s %s (T=%d): worst variable %s" % (name, gemm, T, worst[1]))
