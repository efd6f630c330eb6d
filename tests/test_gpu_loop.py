"""The whole T-step loop (graphnn.py:175-179) as one launch of resident workgroups with per-group synchronisation, in both
forms -- tspgnn_mp_loop_h2 (edge states in registers) and tspgnn_mp_resident_h2 (edge states through memory, work items by
LDS ticket) -- against the stepwise launches they replace (bit-identical: same arithmetic, same summation orders) and
against the float64 oracle (1e-5, BASELINE.json)."""
import numpy as np
import pytest
import torch

import tspgnn
from conftest import batch_from_tuple, rel_err
from oracle import params as P
from oracle import torch_oracle as TO
from test_gpu_model import pack_tuple

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5


@pytest.fixture(autouse=True, params=["loop", "resident"])
def kind(request, monkeypatch):
    """By default the register-resident loop only takes batches of <= 3 resident tiles per wavefront (where it is the
    faster path, loop_plan.max_edge_tiles) and the memory-resident one the larger ones; the tests exercise everything each
    kernel holds."""
    monkeypatch.setenv("TSPGNN_LOOP_MAX_TILES", "4")
    monkeypatch.setenv("TSPGNN_LOOP_KIND", request.param)
    return request.param


def forward(params, t, T, loop, d=64):
    model = tspgnn.build_network(d)
    model["gnn"].persistent_loop = loop
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    used = b.adj.loop_plan is not None and loop
    if used:
        import os
        assert b.adj.loop_plan[3] == os.environ["TSPGNN_LOOP_KIND"]
    pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
    return pred, last, used


def assert_bit_equal(a, b):
    (pa, la, _), (pb, lb, _) = a, b
    assert np.array_equal(np.asarray(pa), np.asarray(pb))
    for var in ("E", "V"):
        assert np.array_equal(np.asarray(la[var].h), np.asarray(lb[var].h)), var + ".h"
        assert np.array_equal(np.asarray(la[var].c), np.asarray(lb[var].c)), var + ".c"


@pytest.mark.parametrize("name,T", [("n20_B32", 8), ("n20_B32", 1), ("n20_B32", 2), ("ragged_B6", 5), ("sparse_B4", 7),
                                    ("target_B4", 3)])
def test_loop_equals_stepwise_launches_on_reference_fixtures(cuda_device, name, T):
    t = pack_tuple(name)
    params = P.init_params(64, seed=3, perturb=True)
    one = forward(params, t, T, True)
    steps = forward(params, t, T, False)
    assert one[2], "the one-launch loop did not take this batch"
    assert_bit_equal(one, steps)
    ref = TO.forward(TO.to_torch(params, torch.float64), batch_from_tuple(t), T)
    assert rel_err(one[0], ref["predictions"].numpy()) < REL_TOL
    assert rel_err(one[1]["E"].h, ref["last_states"]["E"][0].numpy()) < REL_TOL
    assert rel_err(one[1]["V"].c, ref["last_states"]["V"][1].numpy()) < REL_TOL


@pytest.mark.parametrize("sizes,T", [([40] * 128, 32), ([40] * 16, 6), ([7, 33, 12, 40, 40, 21, 5, 64, 3, 17] * 4, 9),
                                     ([40] * 131, 4)])
def test_loop_equals_stepwise_launches_at_size(cuda_device, sizes, T):
    """C2 itself (128 x n = 40, T = 32: every wavefront owns 3-4 tiles), a batch that leaves most workgroups idle, ragged
    instances down to n = 3 (several instances per group, groups that straddle wavefronts), and a batch one instance
    pair over what fits 7 tiles per SIMD."""
    t = tspgnn.synthetic_batch(sizes, seed=5)
    params = P.init_params(64, seed=9, perturb=True)
    one = forward(params, t, T, True)
    steps = forward(params, t, T, False)
    assert one[2], "the one-launch loop did not take this batch"
    assert_bit_equal(one, steps)


def test_loop_replays_are_bit_identical_and_serve_fresh_batches(cuda_device):
    """A captured forward (HIP graph) whose loop is the one launch: replays are bit-identical, and a different batch of the
    same block structure copied into the captured buffers (DeviceBatch.copy_from copies the work plan too) gives that
    batch's own results."""
    d, T = 64, 12
    params = P.init_params(d, seed=2, perturb=True)
    model = tspgnn.build_network(d)
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)

    def feed_of(t):
        EV, W, C, route_exists, n_vertices, n_edges = t
        return {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
                model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    ta, tb = tspgnn.synthetic_batch([40] * 32, seed=1), tspgnn.synthetic_batch([40] * 32, seed=2)
    ba, bb = sess.prepare(feed_of(ta)), sess.prepare(feed_of(tb))
    assert ba.adj.loop_plan is not None
    want_a = sess.forward_device(ba)["predictions"].clone()
    want_b = sess.forward_device(bb)["predictions"].clone()
    replay = sess.capture_forward(ba)
    for _ in range(3):
        assert torch.equal(replay()["predictions"], want_a)
    ba.copy_from(bb)
    assert torch.equal(replay()["predictions"], want_b)
    assert not torch.equal(want_a, want_b)
    assert not sess.range_exceeded()


def test_resident_loop_without_the_placement_assumption(cuda_device, kind, monkeypatch):
    """tspgnn_mp_resident_h2 checks inside the launch that a group's workgroups share an XCD (the one thing its L1-only
    invalidate leans on) and takes the agent-scope acquire when they do not; TSPGNN_RES_SAFE=1 raises the mismatch flag by
    hand: the results must be the same bits either way."""
    if kind != "resident":
        pytest.skip("the placement check belongs to the memory-resident form")
    t = tspgnn.synthetic_batch([40] * 48 + [17, 23, 31], seed=11)
    params = P.init_params(64, seed=4, perturb=True)
    fast = forward(params, t, 7, True)
    monkeypatch.setenv("TSPGNN_RES_SAFE", "1")
    safe = forward(params, t, 7, True)
    steps = forward(params, t, 7, False)
    assert fast[2] and safe[2]
    assert_bit_equal(fast, steps)
    assert_bit_equal(safe, steps)


def test_default_selector_takes_the_resident_loop_in_its_window(cuda_device, kind, monkeypatch):
    """TSPGNN_LOOP_KIND unset (the shipped default): a batch inside resident_plan.in_auto_window -- 192 instances of n = 40,
    149 760 edge rows -- goes to tspgnn_mp_resident_h2, C2 (99 840 rows) to the stepwise launches, 32 instances of n = 20 to
    the register-resident loop; the window batch matches the stepwise launches bit for bit, at 256 instances too (the size
    at which a plain load of the edge states was caught returning a stale line)."""
    if kind != "resident":
        pytest.skip("one run is enough")
    monkeypatch.delenv("TSPGNN_LOOP_KIND")
    monkeypatch.delenv("TSPGNN_LOOP_MAX_TILES")
    params = P.init_params(64, seed=12, perturb=True)
    model = tspgnn.build_network(64)
    sess = tspgnn.Session(model)

    def plan_kind(sizes):
        EV, W, C, route_exists, n_vertices, n_edges = tspgnn.synthetic_batch(sizes, seed=3)
        b = sess.prepare({model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: 2, model["route_exists"]: route_exists,
                          model["n_vertices"]: n_vertices, model["n_edges"]: n_edges})
        return None if b.adj.loop_plan is None else b.adj.loop_plan[3]
    assert plan_kind([40] * 192) == "resident" and plan_kind([40] * 128) is None and plan_kind([20] * 32) == "loop"
    for n_inst, T in ((192, 5), (256, 3)):
        t = tspgnn.synthetic_batch([40] * n_inst, seed=8)
        if n_inst == 256:
            monkeypatch.setenv("TSPGNN_LOOP_KIND", "resident")
        one = forward_any(params, t, T, True)
        steps = forward_any(params, t, T, False)
        assert one[2] == "resident" and steps[2] is None
        assert_bit_equal(one, steps)


def forward_any(params, t, T, loop):
    model = tspgnn.build_network(64)
    model["gnn"].persistent_loop = loop
    sess = tspgnn.Session(model)
    sess.run(tspgnn.global_variables_initializer())
    model.store.load(params)
    EV, W, C, route_exists, n_vertices, n_edges = t
    feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T,
            model["route_exists"]: route_exists, model["n_vertices"]: n_vertices, model["n_edges"]: n_edges}
    b = sess.prepare(feed)
    used = b.adj.loop_plan[3] if (b.adj.loop_plan is not None and loop) else None
    pred, last = sess.run([model["predictions"], model["last_states"]], feed_dict=feed)
    return pred, last, used
