"""Host logic of the one-launch T-step loop (tspgnn/loop_plan.py, csrc/mp_loop_h2.hip): the work plan covers every row
exactly once and respects the kernel's limits, and the synchronisation protocol the kernel runs over it -- three monotone
counters per group, buffers double-buffered by step parity -- is free of deadlock and of read/write hazards under
randomly interleaved schedules (a model of the protocol, phase by phase, not of the arithmetic)."""
import random

import numpy as np
import pytest

import tspgnn
from tspgnn import loop_plan as LP


def blocks_of(sizes, connectivity=1.0, seed=0):
    t = tspgnn.synthetic_batch(sizes, seed=seed, connectivity=connectivity)
    return t[0], t[0].blocks


def decode(plan, grid):
    p = np.asarray(plan).reshape(grid, LP.WAVES, LP.DESC)
    edge, vert = [], []
    for b in range(grid):
        for w in range(LP.WAVES):
            d = p[b, w]
            if d[0] == 1 and d[1] > 0:
                edge.append(dict(wg=b, wave=w, tiles=[(int(d[2 + i]), int(d[6 + i])) for i in range(d[1])], ga=int(d[10]),
                                 gb=int(d[11]), n_a=int(d[12]), n_b=int(d[13]), nvt_a=int(d[14]), nvt_b=int(d[15]),
                                 net_a=int(d[16]), share=(int(d[17]), int(d[18]))))
            elif d[0] == 2 and d[1] > 0:
                vert.append(dict(wg=b, wave=w, tiles=[(int(d[2 + i]), int(d[6 + i]), int(d[10 + i]), int(d[12 + i]))
                                                      for i in range(d[1])]))
    return p, edge, vert


@pytest.mark.parametrize("sizes,grid", [([40] * 128, 256), ([20] * 32, 256), ([7, 33, 12, 40, 40, 21, 5, 64, 3, 17] * 4, 256),
                                        ([40] * 131, 256), ([12] * 16, 16), ([3, 3, 3, 9], 16)])
def test_plan_covers_every_row_once_within_the_kernels_limits(sizes, grid):
    ev, (e_start, v_start) = blocks_of(sizes)
    built = LP.build(e_start, v_start, grid=grid, max_tiles=4)
    assert built is not None
    plan, G = built
    p, edge, vert = decode(plan, grid)
    M, N = ev.shape
    e_seen, v_seen, s_seen = np.zeros(M, int), np.zeros(N, int), np.zeros(N, int)
    for w in edge:
        assert 1 <= len(w["tiles"]) <= LP.EDGE_TILES
        for r, n in w["tiles"]:
            assert 1 <= n <= 16
            e_seen[r:r + n] += 1
        assert w["n_a"] + w["n_b"] == len(w["tiles"]) and (w["n_b"] == 0) == (w["ga"] == w["gb"])
        s_seen[w["share"][0]:w["share"][1]] += 1
    for w in vert:
        assert 1 <= len(w["tiles"]) <= LP.VERT_TILES
        for r, n, g, nvg in w["tiles"]:
            v_seen[r:r + n] += 1
    assert np.all(e_seen == 1) and np.all(v_seen == 1) and np.all(s_seen == 1)
    # roles are uniform over a workgroup, and a group's workgroups share one XCD (workgroup b -> XCD b mod 8)
    assert all(len(set(p[b, :, 0])) == 1 for b in range(grid))
    xcd_of_group = {}
    for w in edge:
        for g in (w["ga"], w["gb"]):
            assert xcd_of_group.setdefault(g, w["wg"] % 8) == w["wg"] % 8
    for w in vert:
        for _, _, g, _ in w["tiles"]:
            assert xcd_of_group.setdefault(g, w["wg"] % 8) == w["wg"] % 8
    assert len(xcd_of_group) == G
    # an edge never leaves its group's vertex range: the endpoints a tile gathers belong to a group it waits for
    uv = ev.uv
    vt_rows = {}
    for w in vert:
        for r, n, g, nvg in w["tiles"]:
            lo, hi = vt_rows.get(g, (10 ** 9, -1))
            vt_rows[g] = (min(lo, r), max(hi, r + n))
    for w in edge:
        groups = [w["ga"]] * w["n_a"] + [w["gb"]] * w["n_b"]
        for (r, n), g in zip(w["tiles"], groups):
            lo, hi = vt_rows[g]
            assert uv[r:r + n].min() >= lo and uv[r:r + n].max() < hi


def test_c2_plan_is_the_balanced_one():
    """C2 (128 x n = 40): 28 edge + 4 vertex workgroups per XCD, 784 tiles per XCD = 7 per SIMD (wavefronts w and w + 4
    share a SIMD: 4 + 3 tiles), groups of two instances = 5 full vertex tiles."""
    ev, (e_start, v_start) = blocks_of([40] * 128)
    assert LP.build(e_start, v_start, grid=256) is None      # (default: only batches of <= 3 tiles per wavefront)
    plan, G = LP.build(e_start, v_start, grid=256, max_tiles=4)
    assert G == 64 and LP.describe(plan, 256) == (224, 32, 4, 2)
    p = np.asarray(plan).reshape(256, LP.WAVES, LP.DESC)
    for b in range(256):
        if p[b, 0, 0] == 1:
            per_simd = [p[b, w, 1] + p[b, w + 4, 1] for w in range(4)]
            assert per_simd == [7, 7, 7, 7]


def test_oversize_and_empty_batches_are_declined():
    _, (e_start, v_start) = blocks_of([40] * 160)       # 124 800 edge rows: more than 4 tiles per wavefront
    assert LP.build(e_start, v_start, grid=256, max_tiles=4) is None
    assert LP.build(np.array([0, 0]), np.array([0, 5]), grid=256) is None
    _, (e_start, v_start) = blocks_of([40] * 96)        # 3 tiles per wavefront: taken by default
    assert LP.describe(LP.build(e_start, v_start, grid=256)[0], 256)[2] == 3


def test_block_structure_from_the_endpoint_list():
    ev, (e_start, v_start) = blocks_of([5, 9, 3, 12], seed=3)
    got = LP.block_structure(ev.uv, ev.shape[1])
    assert np.array_equal(got[0], e_start) and np.array_equal(got[1], v_start)
    rng = np.random.RandomState(0)     # edges shuffled inside their instances, endpoints swapped: same blocks
    perm = np.concatenate([e_start[i] + rng.permutation(int(e_start[i + 1] - e_start[i])) for i in range(4)])
    uv = ev.uv[perm][:, ::-1]
    got = LP.block_structure(uv, ev.shape[1])
    assert np.array_equal(got[0], e_start) and np.array_equal(got[1], v_start)


@pytest.mark.parametrize("sizes,grid,T", [([12] * 16, 16, 5), ([3, 3, 3, 9, 14, 6, 6, 20], 16, 4), ([20] * 32, 256, 3)])
def test_protocol_has_no_deadlock_and_no_buffer_hazard(sizes, grid, T):
    """Event model of mp_loop_h2_kernel's synchronisation.  Actors: every edge wavefront, every vertex workgroup (its
    wavefronts move in lock step).  Per step an edge wavefront (A) waits for its share group's message tiles, reads the
    message rows of its share's edges, writes its aggregated rows, arrives; (B) waits for the projected vertex tiles of its
    tiles' groups, reads them, writes its message rows of the next step, arrives.  A vertex workgroup waits for its tiles'
    aggregated rows, reads them, writes the projected rows of the next step, arrives.  Every buffer cell carries the step
    whose data it holds; a read checks it at the start AND at the end of its window, so a producer running ahead into a
    buffer still being read is caught.  Schedules are random; every actor must finish.  (The first version of the kernel
    had ONE counter per group and kind; this model found the flaw -- a vertex workgroup one step ahead of a sibling
    completes the count of the step the sibling has not stored yet -- which the GPU runs had not shown.  Counters are now
    split by step parity; a producer can lead a sibling by one step, never by two.)"""
    ev, (e_start, v_start) = blocks_of(sizes)
    plan, G = LP.build(e_start, v_start, grid=grid, max_tiles=4)
    p, edge, vert = decode(plan, grid)
    M, N = ev.shape
    uv = ev.uv
    rowptr, eid = ev.csr_by_vertex()
    for trial in range(4):
        rng = random.Random(trial)
        msg = [np.zeros(M, int), np.full(M, -9)]      # version held: messages of step 0 arrive from the launch before
        zx = [np.zeros(N, int), np.full(N, -9)]
        vagg = [np.full(N, -9), np.full(N, -9)]
        cnt = {(k, par): np.zeros(G, int) for k in ("msg", "vagg", "zx") for par in (0, 1)}   # by the parity of the step announced

        def edge_actor(w):
            rows = np.concatenate([np.arange(r, r + n) for r, n in w["tiles"]])
            ends = np.unique(uv[rows])
            s0, s1 = w["share"]
            share_edges = np.unique(np.concatenate([eid[rowptr[v]:rowptr[v + 1]] for v in range(s0, s1)])) if s1 > s0 \
                else np.zeros(0, int)
            for t in range(T):
                pz = t & 1
                if s1 > s0:
                    yield ("wait", ("msg", pz), w["ga"], (t + 1) // 2 * w["net_a"])
                    assert np.all(msg[pz][share_edges] == t), "row-sum reads messages of the wrong step"
                    yield ("run",)
                    assert np.all(msg[pz][share_edges] == t), "messages overwritten under the row-sum"
                    vagg[pz][s0:s1] = t
                    cnt["vagg", pz][w["ga"]] += s1 - s0
                yield ("wait", ("zx", pz), w["ga"], (t + 1) // 2 * w["nvt_a"])
                if w["gb"] != w["ga"]:
                    yield ("wait", ("zx", pz), w["gb"], (t + 1) // 2 * w["nvt_b"])
                assert np.all(zx[pz][ends] == t), "edge cell reads projected messages of the wrong step"
                yield ("run",)
                assert np.all(zx[pz][ends] == t), "projected messages overwritten under the edge cell"
                if t < T - 1:
                    msg[1 - pz][rows] = t + 1
                    cnt["msg", 1 - pz][w["ga"]] += w["n_a"]
                    if w["n_b"]:
                        cnt["msg", 1 - pz][w["gb"]] += w["n_b"]

        def vertex_actor(waves):
            for t in range(T):
                pz = t & 1
                for w in waves:
                    for r, n, g, nvg in w["tiles"]:
                        yield ("wait", ("vagg", pz), g, (t // 2 + 1) * nvg)
                        assert np.all(vagg[pz][r:r + n] == t), "vertex cell reads aggregates of the wrong step"
                yield ("run",)
                for w in waves:
                    for r, n, g, nvg in w["tiles"]:
                        assert np.all(vagg[pz][r:r + n] == t), "aggregates overwritten under the vertex cell"
                if t == T - 1:
                    break
                yield ("run",)
                for w in waves:
                    for r, n, g, nvg in w["tiles"]:
                        zx[1 - pz][r:r + n] = t + 1
                        cnt["zx", 1 - pz][g] += 1

        actors = [edge_actor(w) for w in edge]
        by_wg = {}
        for w in vert:
            by_wg.setdefault(w["wg"], []).append(w)
        actors += [vertex_actor(ws) for ws in by_wg.values()]
        pending = [next(a) for a in actors]
        alive = list(range(len(actors)))
        while alive:
            ready = [i for i in alive if pending[i][0] == "run" or cnt[pending[i][1]][pending[i][2]] >= pending[i][3]]
            assert ready, "deadlock: %d actors wait, none can run" % len(alive)
            i = rng.choice(ready)
            try:
                pending[i] = next(actors[i])
            except StopIteration:
                alive.remove(i)
