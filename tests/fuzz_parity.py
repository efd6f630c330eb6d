#!/usr/bin/env python
"""Randomised parity sweep (a script, not collected by pytest): random batch shapes -- 1..12 graphs of 3..45
vertices, complete or sparse, d in {32, 64}, T in 0..6 -- through the HIP path and the float64 oracle; forward for
both GEMM arithmetics, gradients on every third case.  Prints one line per case and the worst errors; exits
non-zero on the first violation of the 1e-5 bar (forward).  Gradients outside the budget of the suite (2e-5 per variable,
or 3x what the fp32 autograd restatement loses itself) are re-run through the other GEMM arithmetic and counted: a relu
whose pre-activation sits at the rounding level takes either branch, a step function of the inputs; only a gross error in
both arithmetics fails the sweep.  `tests/test_gpu_model.py::test_random_shapes_parity` runs a
fixed, short prefix of the same sequence in the suite.

    python tests/fuzz_parity.py [n_cases] [seed]
    BF16=1 python tests/fuzz_parity.py [n_cases] [seed]     # the bf16-storage mode against the bf16-rounding oracle
    DET=1 python tests/fuzz_parity.py [n_cases] [seed]      # eager vs captured training steps, bit for bit
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), "tsp-gnn_amd"), os.path.dirname(HERE), HERE]

import tspgnn  # noqa: E402
from oracle import params as P  # noqa: E402
from oracle import torch_oracle as TO  # noqa: E402

REL_TOL = 1e-5


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def draw_case(rng):
    B = int(rng.randint(1, 13))
    sizes = [int(x) for x in rng.randint(3, 46, size=B)]
    conn = float(rng.choice([1.0, 1.0, 0.7, 0.4]))
    d = int(rng.choice([32, 64, 64]))
    T = int(rng.randint(0, 7))
    return sizes, conn, d, T, int(rng.randint(0, 1 << 30))


def feed_of(model, t, T):
    EV, W, C, route_exists, n_vertices, n_edges = t
    return {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}


def run_case(idx, case, with_grads, strict=True):
    """-> dict of errors; raises AssertionError past the bar."""
    sizes, conn, d, T, seed = case
    t = tspgnn.synthetic_batch(sizes, seed=seed, connectivity=conn)
    params = P.init_params(d, seed=seed % 9973, perturb=True)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T)
    errs = {}
    for gemm in ("f16x2", "bf16x3", "f32"):
        model = tspgnn.build_network(d)
        model["gnn"].gemm = gemm
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        pred, loss, last = sess.run([model["predictions"], model["loss"], model["last_states"]], feed_dict=feed_of(model, t, T))
        e = max(rel_err(pred, ref["predictions"].numpy()),
                rel_err(last["E"].h, ref["last_states"]["E"][0].numpy()), rel_err(last["E"].c, ref["last_states"]["E"][1].numpy()),
                rel_err(last["V"].h, ref["last_states"]["V"][0].numpy()), rel_err(last["V"].c, ref["last_states"]["V"][1].numpy()))
        errs[gemm] = e
        if gemm == "f16x2":   # did the range guard send this batch to bf16x3?  (1: top of the range, 2: the variance floor)
            errs["guard_bits"] = float(sess.last_range_bits)
        assert e < REL_TOL, ("forward", gemm, case, e)
        assert abs(float(loss) - ref["loss"].item()) < REL_TOL, ("loss", gemm, case)
        if gemm == "f16x2" and with_grads:
            _, ref_g = TO.loss_and_grads(params, batch, T, dtype=torch.float64)
            _, f32_g = TO.loss_and_grads(params, batch, T, dtype=torch.float32, dense=True)
            l2 = {k: TO.L2NORM_SCALING * params[k] for k in params}
            gscale = max(np.abs(ref_g[k] - l2[k]).max() for k in ref_g)

            def grad_errors(session, mdl):
                session.loss_and_grads(feed_of(mdl, t, T))
                torch.cuda.synchronize()
                g = mdl.store.grad_dict()
                out = {}
                for k in ref_g:
                    r = ref_g[k] - l2[k]
                    scale = max(np.abs(r).max(), 1e-3 * gscale)
                    out[k] = (np.abs(g[k] - r).max() / scale, np.abs(f32_g[k] - ref_g[k]).max() / scale)
                return out

            # budget: 2e-5 per variable, or what the op-for-op fp32 autograd run loses itself (sums that nearly cancel)
            ge = grad_errors(sess, model)
            over = [k for k, (e, e32) in ge.items() if e >= max(2e-5, 3 * e32)]
            errs["grad"] = max(e for e, _ in ge.values())
            errs["grad_fp32_oracle"] = max(e32 for _, e32 in ge.values())
            if over:
                # A relu whose pre-activation sits at the rounding level (|z| ~ 1e-9) takes one branch in one fp32-class
                # arithmetic and the other branch in another -- a step function of the inputs (seed 77 case 129: the E
                # cell's gate; seed 4242 case 78: one vote-head unit of one edge of a 151-edge graph moves every gradient
                # by 1e-2).  Such a flip is specific to ONE arithmetic: the same gradients through the fp32-MFMA kernels
                # must then be within budget; off in both means a real defect.
                assert not strict, ("grad", over[0], case, ge[over[0]])
                m2 = tspgnn.build_network(d)
                m2["gnn"].gemm = "f32"
                s2 = tspgnn.Session(m2)
                s2.run(tspgnn.global_variables_initializer())
                m2.store.load(params)
                ge2 = grad_errors(s2, m2)
                over2 = [k for k, (e, e32) in ge2.items() if e >= max(2e-5, 3 * e32)]
                errs["grad_branch_flip_vars"] = float(len(over))
                if over2:
                    # (a flip inside a stage both arithmetics share -- E_init, the vote head -- shows in both: seed 4242
                    # case 177, one hidden unit of E_init for one edge of a 36-edge graph, 1.2e-4 on a [2,8] kernel)
                    worst2 = max(ge2[k][0] for k in over2)
                    errs["grad_over_budget_in_both"] = worst2
                    assert worst2 < 5e-2, ("grad off in BOTH arithmetics", over2[0], case, ge2[over2[0]], ge[over[0]])
    return errs


def run_case_bf16(case):
    """The bf16-storage mode (BASELINE config 5) on the same random shapes, d in {64, 128}: against the oracle
    that rounds at the same points (a handful of bf16 ulps) -- the suite's bars (tests/test_gpu_model.py)."""
    sizes, conn, d, T, seed = case
    d = 128 if d == 32 else d
    t = tspgnn.synthetic_batch(sizes, seed=seed, connectivity=conn)
    params = P.init_params(d, seed=seed % 9973, perturb=True)
    batch = {"ev_uv": t[0].uv, "W": t[1], "C": t[2], "route_exists": t[3], "n_vertices": t[4], "n_edges": t[5]}
    ref = TO.forward(TO.to_torch(params, torch.float64), batch, T, bf16=True)
    model = tspgnn.build_network(d, float_dtype=torch.bfloat16)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed_of(model, t, T))
    e_h = rel_err(last["E"].h, ref["last_states"]["E"][0].numpy())
    e_c = rel_err(last["V"].c, ref["last_states"]["V"][1].numpy())
    e_p = rel_err(pred, ref["predictions"].numpy())
    assert e_h < 3e-2 and e_c < 3e-2 and e_p < 1e-2, ("bf16", case, e_h, e_c, e_p)
    return {"bf16_h": e_h, "bf16_c": e_c, "bf16_pred": e_p}


def run_case_determinism(case):
    """Eager train_step vs the captured one (two HIP graphs), three steps each, on the same random shape: losses and
    final weights must be identical bit for bit, and a captured forward must replay bit-identically."""
    sizes, conn, d, T, seed = case
    t = tspgnn.synthetic_batch(sizes, seed=seed, connectivity=conn)
    params = P.init_params(d, seed=seed % 9973, perturb=True)
    finals = []
    for captured in (False, True):
        model = tspgnn.build_network(d)
        sess = tspgnn.Session(model)
        sess.run(tspgnn.global_variables_initializer())
        model.store.load(params)
        b = sess.prepare(feed_of(model, t, T))
        if captured:
            fwd = sess.capture_forward(b)
            p0 = fwd()["predictions"].clone()
            assert torch.equal(fwd()["predictions"], p0) and torch.equal(sess.forward_device(b)["predictions"], p0), ("replay", case)
        step = sess.capture_train_step(b) if captured else (lambda: sess.train_step(b))
        losses = [float(step()["stats"][0].item()) for _ in range(3)]
        torch.cuda.synchronize()
        finals.append((model.store.theta.clone(), losses))
    assert finals[0][1] == finals[1][1], ("losses", case, finals[0][1], finals[1][1])
    assert torch.equal(finals[0][0], finals[1][0]), ("weights", case)
    return {"det_ok": 1.0}


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    worst = {}
    flagged = 0
    for i in range(n_cases):
        case = draw_case(rng)
        if os.environ.get("BF16"):
            errs = run_case_bf16(case)
        elif os.environ.get("DET"):
            errs = run_case_determinism(case)
        else:
            errs = run_case(i, case, with_grads=(i % 3 == 0), strict=False)
        flagged += 1 if errs.get("guard_bits") else 0
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
        print("case %3d  B=%2d n=%s conn=%.1f d=%d T=%d  %s" % (
            i, len(case[0]), "%d..%d" % (min(case[0]), max(case[0])), case[1], case[2], case[3],
            "  ".join("%s %.2e" % kv for kv in sorted(errs.items()))), flush=True)
    print("worst over %d cases: %s" % (n_cases, "  ".join("%s %.2e" % kv for kv in sorted(worst.items()))))
    print("f16x2 batches the range guard sent to bf16x3: %d of %d" % (flagged, n_cases))


if __name__ == "__main__":
    main()
