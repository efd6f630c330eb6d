"""Host logic of the memory-resident one-launch loop (tspgnn/resident_plan.py, csrc/mp_resident_h2.hip): the work plan
covers every row exactly once within the kernel's limits, and the protocol the kernel runs over it -- items by LDS ticket
through all steps, four parity-split monotone counters per group, buffers double-buffered by step parity, a tile's
publication deferred behind its successor's loads -- is free of deadlock and of read/write hazards under randomly
interleaved schedules (a model of the protocol, phase by phase, not of the arithmetic)."""
import random

import numpy as np
import pytest

import tspgnn
from tspgnn import resident_plan as RP

SIZES = [([40] * 128, 256), ([40] * 32, 256), ([20] * 32, 256), ([7, 33, 12, 40, 40, 21, 5, 64, 3, 17] * 4, 256),
         ([40] * 200, 256), ([12] * 16, 32), ([3, 3, 3, 9], 32)]


def blocks_of(sizes, seed=0):
    t = tspgnn.synthetic_batch(sizes, seed=seed)
    return t[0], t[0].blocks


@pytest.mark.parametrize("sizes,grid", SIZES)
def test_plan_covers_every_row_once_within_the_kernels_limits(sizes, grid):
    ev, (e_start, v_start) = blocks_of(sizes)
    built = RP.build(e_start, v_start, grid=grid)
    assert built is not None
    hdr, items = RP.decode(built)
    M, N = ev.shape
    e_seen, s_seen, c_seen, m_seen = np.zeros(M, int), np.zeros(N, int), np.zeros(N, int), np.zeros(N, int)
    slots = np.zeros(built["n_slots"], int)
    xcd_of_group = {}
    for b in range(grid):
        role, i0, n, slot0, g_first, n_local = (int(x) for x in hdr[b, :6])
        if role == 0 or n == 0:
            continue
        mine = items[i0:i0 + n]
        for it in mine:
            xcd_of_group.setdefault(int(it[2]), set()).add(b % RP.N_XCD)
        if role == 1:
            tiles = mine[mine[:, 3] >= 0]
            shares = mine[mine[:, 3] == -1]
            assert len(tiles) + len(shares) == n
            assert sorted(tiles[:, 3].tolist()) == list(range(n_local)) and n_local <= RP.LDS_WORD_LIMIT
            assert g_first == (tiles[:, 2].min() if len(tiles) else 0)
            for r, nv, g, local, cnt in tiles[:, :5]:
                assert 1 <= nv <= 16
                e_seen[r:r + nv] += 1
                slots[slot0 + local] += 1
            words = n_local
            for v0, v1, g, _, cnt, off in shares[:, :6]:
                assert 1 <= v1 - v0 <= RP.SHARE_ROWS
                s_seen[v0:v1] += 1
                if off >= 0:      # the share's edge-list block: behind the tile words, not overlapping its neighbours
                    assert off == words - n_local
                    words += RP.SHARE_BLOCK
            assert words <= built["lds_words"] <= RP.LDS_WORD_LIMIT
        else:
            assert n <= RP.WAVES * RP.VERT_TILES and np.all(mine[:, 3] == -2)
            for r, nv, g, _, cnt in mine[:, :5]:
                (c_seen if role == 2 else m_seen)[r:r + nv] += 1
    assert np.all(e_seen == 1) and np.all(s_seen == 1) and np.all(c_seen == 1) and np.all(m_seen == 1)
    assert np.all(slots[:int(slots.sum())] == 1)
    # everything a group needs runs on ONE XCD (workgroup b -> XCD b mod 8): the kernel's hand-offs stay within one L2
    assert all(len(x) == 1 for x in xcd_of_group.values())


def simulate(built, T, seed, n_waves=RP.WAVES):
    """Event model of mp_resident_h2_kernel.  Actors: every wavefront of every edge workgroup (tickets in order from the
    workgroup's counter), every wavefront of the vertex cell / message workgroups (their own tiles, step by step).  Data
    is modelled as VERSIONS: a buffer entry holds the step whose data it carries; every read asserts the version it needs
    -- a stale or an early-overwritten entry fails the assert -- and the run must end with every actor finished."""
    rng = random.Random(seed)
    grid = built["grid"]
    hdr, items = RP.decode(built)
    G = built["n_groups"]
    cnt = {k: np.zeros((G, 2), int) for k in ("msg", "vagg", "zx", "vh")}
    # group tables from the items
    et, vt, vrows = {}, {}, {}
    for it in items:
        if it[3] >= 0:
            vt[int(it[2])] = int(it[4])
        elif it[3] == -1:
            et[int(it[2])] = int(it[4])
    e_tiles_of = {}
    v_tiles_of = {}
    for b in range(grid):
        role, i0, n = int(hdr[b, 0]), int(hdr[b, 1]), int(hdr[b, 2])
        for it in items[i0:i0 + n] if role in (1, 2) else ():
            if role == 1 and it[3] >= 0:
                e_tiles_of.setdefault(int(it[2]), []).append(int(it[0]))
            if role == 2:
                v_tiles_of.setdefault(int(it[2]), []).append(int(it[0]))
                vrows[int(it[2])] = int(it[4])
    msg = [dict(), dict()]     # [parity][edge tile row0] -> version (messages OF step v); step 0's come from the pre-launch
    zx = [dict(), dict()]      # [parity][vertex tile row0] -> version
    vagg = [dict(), dict()]    # [parity][vertex row] -> version
    vh = [dict(), dict()]      # [parity][vertex tile row0] -> version of h (h after step v-1, i.e. input of step v)
    for g, rows in e_tiles_of.items():
        for r in rows:
            msg[0][r] = 0
    for g, rows in v_tiles_of.items():
        for r in rows:
            zx[0][r] = 0
    state = {}                 # (wg, local) -> version of the tile's state (input of step v)
    done = {}                  # (wg, local) -> steps completed, as PUBLISHED

    def edge_wave(b):
        role, i0, n = int(hdr[b, 0]), int(hdr[b, 1]), int(hdr[b, 2])
        pend = None
        while True:
            k = ticket[b]
            ticket[b] += 1
            if k >= n * T:
                break
            t, j = divmod(k, n)
            it = items[i0 + j]
            a0, a1, g, local, gc = (int(x) for x in it[:5])
            p = t & 1

            def flush():
                nonlocal pend
                if pend is not None:
                    pl, pt, pg = pend
                    done[(b, pl)] = pt + 1
                    cnt["msg"][pg, 1 - (pt & 1)] += 1
                    pend = None
            if local < 0:                                   # a share of rowsum(g, t)
                flush()
                while cnt["msg"][g, p] < ((t + 1) >> 1) * gc:
                    yield
                for r in e_tiles_of[g]:
                    assert msg[p][r] == t, "row-sum of step %d read messages of step %d" % (t, msg[p][r])
                yield
                for v in range(a0, a1):
                    vagg[p][v] = t
                yield
                cnt["vagg"][g, p] += a1 - a0
                continue
            if pend is not None and pend[1] < t:
                flush()
            if t > 0 and done.get((b, local), 0) < t:
                flush()
                while done.get((b, local), 0) < t:
                    yield
            tgt = ((t + 1) >> 1) * gc
            if cnt["zx"][g, p] < tgt:
                flush()
                while cnt["zx"][g, p] < tgt:
                    yield
            for r in v_tiles_of[g]:                          # the gathers reach any vertex of the group
                assert zx[p][r] == t, "edge step %d gathered projected messages of step %d" % (t, zx[p][r])
            assert state.get((b, local), 0) == t
            yield                                            # loads in flight; GEMM
            flush()                                          # (behind the GEMM: the previous tile's stores have drained)
            yield
            state[(b, local)] = t + 1
            if t < T - 1:
                msg[1 - p][a0] = t + 1
                pend = (local, t, g)
            yield
        if pend is not None:
            done[(b, pend[0])] = pend[1] + 1
            cnt["msg"][pend[2], 1 - (pend[1] & 1)] += 1

    def cell_wave(b, w):
        i0, n = int(hdr[b, 1]), int(hdr[b, 2])
        mine = [items[i0 + idx] for idx in (w, w + n_waves) if idx < n]
        for t in range(T):
            p = t & 1
            for it in mine:
                r0, _, g, _, rows = (int(x) for x in it[:5])
                if t > 0:
                    assert vh[p][r0] == t
                while cnt["vagg"][g, p] < ((t >> 1) + 1) * rows:
                    yield
                # (the tile's own 16 rows; every row of the group has arrived by then)
                for v in range(r0, min(r0 + 16, r0 + int(it[1]))):
                    assert vagg[p][v] == t, "vertex step %d read aggregates of step %d" % (t, vagg[p][v])
                yield
                if t < T - 1:
                    vh[1 - p][r0] = t + 1
                    yield
                    cnt["vh"][g, 1 - p] += 1

    def msg_wave(b, w):
        i0, n = int(hdr[b, 1]), int(hdr[b, 2])
        mine = [items[i0 + idx] for idx in (w, w + n_waves) if idx < n]
        for t in range(T - 1):
            p = t & 1
            for it in mine:
                r0, _, g, _, tiles = (int(x) for x in it[:5])
                while cnt["vh"][g, 1 - p] < ((t >> 1) + 1) * tiles:
                    yield
                assert vh[1 - p][r0] == t + 1
                yield
                zx[1 - p][r0] = t + 1
                yield
                cnt["zx"][g, 1 - p] += 1

    ticket = {}
    actors = []
    for b in range(grid):
        role, n = int(hdr[b, 0]), int(hdr[b, 2])
        if role == 0 or n == 0:
            continue
        if role == 1:
            ticket[b] = 0
            actors += [edge_wave(b) for _ in range(n_waves)]
        elif role == 2:
            actors += [cell_wave(b, w) for w in range(n_waves)]
        else:
            actors += [msg_wave(b, w) for w in range(n_waves)]
    budget = 4000 * len(actors) * T
    while actors:
        budget -= 1
        assert budget > 0, "no progress: deadlock (or livelock) in the protocol model"
        i = rng.randrange(len(actors))
        try:
            next(actors[i])
        except StopIteration:
            actors[i] = actors[-1]
            actors.pop()
    for g in e_tiles_of:
        assert vt[g] == len(v_tiles_of[g]) and et[g] == len(e_tiles_of[g])


@pytest.mark.parametrize("sizes,grid,T", [([12] * 16, 32, 5), ([3, 3, 3, 9], 32, 4), ([20] * 8, 32, 6),
                                          ([7, 33, 12, 40, 21, 5, 17] * 2, 32, 5)])
def test_protocol_model_has_no_deadlock_and_no_hazard(sizes, grid, T):
    ev, (e_start, v_start) = blocks_of(sizes)
    built = RP.build(e_start, v_start, grid=grid)
    assert built is not None
    for seed in range(4):
        simulate(built, T, seed)
    # wavefront counts that starve the model of free actors: one wavefront per workgroup still terminates (an item only
    # ever waits for items of earlier steps)
    simulate(built, T, 99, n_waves=1 if max(int(h[2]) for h in RP.decode(built)[0] if h[0] >= 2) <= 2 else RP.WAVES)


def test_a_single_counter_per_group_would_be_a_race():
    """The model is sharp enough to see what the parity split is for: fold the two parities of one counter into one and a
    fast producer completes the count of a slow sibling's step -- some schedule reads a stale buffer."""
    ev, (e_start, v_start) = blocks_of([12] * 16)
    built = RP.build(e_start, v_start, grid=32)

    class Folded(np.ndarray):
        pass
    failures = 0
    for seed in range(12):
        orig = np.zeros
        try:
            def zeros(shape, dtype=float):
                a = orig(shape, dtype)
                if shape == (built["n_groups"], 2):     # the counters: both parities alias one word
                    return _Aliased(a)
                return a
            np.zeros = zeros
            simulate(built, 6, seed)
        except (AssertionError, KeyError):    # a stale version, or an entry read before anybody wrote it
            failures += 1
        finally:
            np.zeros = orig
    assert failures > 0


class _Aliased(object):
    """A [G, 2] counter array whose two columns are one word."""

    def __init__(self, a):
        self.a = a

    def __getitem__(self, key):
        return self.a[key[0], 0]

    def __setitem__(self, key, value):
        self.a[key[0], 0] = value
