"""Host-side work plan of tspgnn_mp_loop_h2 (include/tspgnn.h, csrc/mp_loop_h2.hip): the whole T-step loop of
graphnn.py:175-179 as one launch of resident workgroups.

EV is block-diagonal by instance (instance_loader.py:56-66), so the plan cuts the batch into GROUPS of consecutive
instances -- the unit of synchronisation inside the launch -- and tells every wavefront which 16-row tiles it owns for
the whole loop:

  * a group's edge rows are cut into tiles of 16 starting at the group's first edge row (the last tile may be short);
    likewise its vertex rows;
  * workgroup b runs on XCD b mod 8 (observed; speed only), so groups are dealt to the 8 XCDs in contiguous, edge-balanced
    ranges and everything a group needs -- its edge wavefronts, its row-sum shares, its vertex tiles -- sits on one XCD;
  * per XCD a few workgroups take the vertex tiles (<= 2 per wavefront), the others the edge tiles (<= 4 per wavefront,
    resident in registers), consecutive tiles to consecutive wavefronts;
  * the V<-E row-sum of a group is shared, by vertex ranges, among the edge wavefronts whose FIRST tile lies in the group.

Descriptor of (workgroup, wavefront), TSPGNN_LOOP_DESC_INTS = 24 int32:
  [0] role 0 idle / 1 edge / 2 vertex   [1] tiles   [2..5] first row of tile i   [6..9] valid rows of tile i
  edge:   [10] group of the first tile (ga)  [11] group of the last tile (gb)  [12] tiles in ga  [13] tiles in gb (0 if
          gb == ga)  [14] vertex tiles of ga  [15] vertex tiles of gb  [16] edge tiles of ga  [17..18] row-sum share
          [v0, v1) (vertex rows of ga; empty if v0 == v1)
  vertex: [10..11] group of tile i  [12..13] vertex rows of that group
"""
import os

import numpy as np

WAVES = 8
DESC = 24
EDGE_TILES = 4      # resident tiles per edge wavefront (32 registers of state each)
VERT_TILES = 2
N_XCD = 8
MIN_GROUP_EDGE_TILES = 4   # a wavefront's <= 4 consecutive tiles then touch <= 2 groups, and every group holds a wavefront's first tile

_cache = {}


def block_structure(uv, n_total):
    """(e_start[B+1], v_start[B+1]) of the block-diagonal components of an incidence pattern whose blocks are contiguous
    in both orders (instance_loader.py:56-66), from the endpoint list alone; None if the pattern is not of that form."""
    uv = np.asarray(uv).reshape(-1, 2)
    M = uv.shape[0]
    if M == 0:
        return None
    lo, hi = uv.min(axis=1), uv.max(axis=1)
    pmax = np.maximum.accumulate(hi)
    smin = np.minimum.accumulate(lo[::-1])[::-1]
    cut = np.nonzero(smin[1:] > pmax[:-1])[0] + 1          # a block starts at edge e when nothing from e on reaches back
    e_start = np.concatenate(([0], cut, [M])).astype(np.int64)
    v_end = pmax[e_start[1:] - 1] + 1                       # block k owns vertices up to its largest endpoint
    v_start = np.concatenate(([0], v_end)).astype(np.int64)
    v_start[-1] = n_total                                   # isolated trailing vertices go with the last block
    if np.any(np.diff(v_start) <= 0) or np.any(lo < v_start[np.searchsorted(e_start, np.arange(M), side="right") - 1]):
        return None
    return e_start, v_start


def _groups(e_start, v_start):
    """Merge consecutive instances into groups: >= MIN_GROUP_EDGE_TILES edge tiles each, and vertex rows a multiple of 16
    where a few instances more achieve it (n = 40: two instances = 80 rows = 5 full tiles)."""
    B = len(e_start) - 1
    groups = []   # (e0, e1, v0, v1)
    i = 0
    while i < B:
        j = i + 1
        while j < B:
            et = (e_start[j] - e_start[i] + 15) // 16
            vr = v_start[j] - v_start[i]
            if et >= MIN_GROUP_EDGE_TILES and (vr % 16 == 0 or j - i >= 4 or vr >= 128):
                break
            j += 1
        groups.append([int(e_start[i]), int(e_start[j]), int(v_start[i]), int(v_start[j])])
        i = j
    # a short tail group joins its predecessor
    while len(groups) > 1 and (groups[-1][1] - groups[-1][0] + 15) // 16 < MIN_GROUP_EDGE_TILES:
        g = groups.pop()
        groups[-1][1], groups[-1][3] = g[1], g[3]
    return groups


def max_edge_tiles():
    """Largest number of resident tiles per edge wavefront a batch may need for the one-launch loop to be chosen.  The
    kernel holds up to EDGE_TILES = 4, but a wavefront works through its tiles one after the other at ~10 us each (two
    wavefronts per SIMD hide each other's stalls, not their own), so the loop's step time is set by the LONGEST wavefront:
    measured on MI355X at n = 40 (profiles/r05_loop_vs_steps.txt), <= 3 tiles per wavefront (<= 96 instances) beats the
    stepwise launches by 4-24 %, 4 tiles (C2's 128 instances) lose 10 %.  TSPGNN_LOOP_MAX_TILES=4 takes every batch that
    fits (tests, A/B runs)."""
    env = os.environ.get("TSPGNN_LOOP_MAX_TILES")
    return max(1, min(EDGE_TILES, int(env))) if env else 3


def build(e_start, v_start, grid=256, vertex_wgs=None, max_tiles=None):
    """-> (plan int32[grid * WAVES * DESC], n_groups) or None when the batch does not fit the resident design (more than
    ``max_tiles`` -- default max_edge_tiles() -- edge tiles per wavefront; 4 = ~115 k edge rows on 256 CUs is what the
    kernel holds) or has no edges."""
    e_start = np.asarray(e_start, dtype=np.int64)
    v_start = np.asarray(v_start, dtype=np.int64)
    if grid < N_XCD * 2 or grid % N_XCD != 0 or e_start[-1] == 0:
        return None
    if vertex_wgs is None:
        env = os.environ.get("TSPGNN_LOOP_VWG")
        vertex_wgs = int(env) if env else None
    if max_tiles is None:
        max_tiles = max_edge_tiles()
    key = (e_start.tobytes(), v_start.tobytes(), grid, vertex_wgs, max_tiles)
    if key in _cache:
        return _cache[key]
    out = _build(e_start, v_start, grid, vertex_wgs)
    if out is not None and describe(out[0], grid)[2] > max_tiles:
        out = None
    if len(_cache) > 64:
        _cache.clear()
    _cache[key] = out
    return out


def _build(e_start, v_start, grid, vertex_wgs):
    groups = _groups(e_start, v_start)
    G = len(groups)
    if (groups[0][1] - groups[0][0] + 15) // 16 < MIN_GROUP_EDGE_TILES and G > 0:
        if G == 1 and groups[0][1] - groups[0][0] == 0:
            return None
    et = np.array([(g[1] - g[0] + 15) // 16 for g in groups], dtype=np.int64)
    vt = np.array([(g[3] - g[2] + 15) // 16 for g in groups], dtype=np.int64)
    if np.any(et < 1):
        return None
    wg_per_xcd = grid // N_XCD
    # groups -> XCDs: contiguous ranges balanced by edge tiles
    cum = np.cumsum(et)
    mid = cum - et / 2.0
    xcd_of = np.minimum((mid * N_XCD / cum[-1]).astype(np.int64), N_XCD - 1)
    plan = np.zeros((grid, WAVES, DESC), dtype=np.int32)
    for x in range(N_XCD):
        gs = np.nonzero(xcd_of == x)[0]
        if len(gs) == 0:
            continue
        ET, VT = int(et[gs].sum()), int(vt[gs].sum())
        # vertex workgroups: enough that the vertex chain (a vertex tile ~2.3 edge tiles of matrix + vector work, spread
        # over the 4 SIMDs, plus two LDS stagings per step) keeps up with the edge wavefronts' tiles per SIMD
        best = None
        for nv in ([vertex_wgs] if vertex_wgs else range(1, wg_per_xcd)):
            ne = wg_per_xcd - nv
            if ne < 1 or nv < 1:
                continue
            per_wg_v = -(-VT // nv)
            per_wg_e = -(-ET // ne)
            if per_wg_v > WAVES * VERT_TILES or per_wg_e > WAVES * EDGE_TILES:
                continue
            edge_time = -(-per_wg_e // 4)
            vert_time = per_wg_v * 0.6 + 1.0
            cost = (max(edge_time, vert_time), nv)
            if best is None or cost < best[0]:
                best = (cost, nv)
        if best is None:
            return None
        nv = best[1]
        ne = wg_per_xcd - nv
        wgs = [x + N_XCD * s for s in range(wg_per_xcd)]
        edge_wgs, vert_wgs = wgs[:ne], wgs[ne:]
        # tiles of the XCD in group order
        etiles, vtiles = [], []
        for gi in gs:
            e0, e1, v0, v1 = groups[gi]
            for r in range(e0, e1, 16):
                etiles.append((r, min(16, e1 - r), int(gi)))
            for r in range(v0, v1, 16):
                vtiles.append((r, min(16, v1 - r), int(gi)))
        # edge tiles -> workgroups -> wavefronts, consecutive
        first_in_group = {}   # group -> [(wg, wave)] whose first tile lies in it
        for k, b in enumerate(edge_wgs):
            t0, t1 = len(etiles) * k // ne, len(etiles) * (k + 1) // ne
            n = t1 - t0
            base, extra = divmod(n, WAVES)
            pos = t0
            for w in range(WAVES):
                cnt = base + (1 if w < extra else 0)
                d = plan[b, w]
                d[0] = 1
                d[1] = cnt
                if cnt == 0:
                    continue
                mine = etiles[pos:pos + cnt]
                pos += cnt
                for i, (r, nvld, _) in enumerate(mine):
                    d[2 + i], d[6 + i] = r, nvld
                ga, gb = mine[0][2], mine[-1][2]
                if any(t[2] not in (ga, gb) for t in mine):
                    return None
                n_a = sum(1 for t in mine if t[2] == ga)
                d[10], d[11] = ga, gb
                d[12], d[13] = n_a, (cnt - n_a if gb != ga else 0)
                d[14], d[15], d[16] = vt[ga], vt[gb], et[ga]
                # (light = fewer tiles than the wavefront it shares a SIMD with: wavefronts w and w + 4)
                first_in_group.setdefault(ga, []).append((b, w, cnt < base + 1 or extra == 0))
        # row-sum shares
        for gi in gs:
            owners = first_in_group.get(int(gi))
            if not owners:
                return None
            # the row-sum goes to the LIGHT wavefronts of the group where there are any: a SIMD holds 7 tiles as 4 + 3, and
            # the 3-tile wavefront sums its share while its neighbour -- which then has none, and goes straight on to the
            # next step -- works on its fourth tile
            light = [(b, w) for b, w, is_light in owners if is_light]
            owners = light if light else [(b, w) for b, w, _ in owners]
            v0, v1 = groups[gi][2], groups[gi][3]
            n = v1 - v0
            for k, (b, w) in enumerate(owners):
                plan[b, w, 17] = v0 + n * k // len(owners)
                plan[b, w, 18] = v0 + n * (k + 1) // len(owners)
        # vertex tiles -> workgroups -> wavefronts (tile j of a workgroup: wavefront j mod 8, slot j div 8)
        for k, b in enumerate(vert_wgs):
            plan[b, :, 0] = 2
            t0, t1 = len(vtiles) * k // nv, len(vtiles) * (k + 1) // nv
            for j, (r, nvld, gi) in enumerate(vtiles[t0:t1]):
                w, s = j % WAVES, j // WAVES
                d = plan[b, w]
                d[1] = s + 1
                d[2 + s], d[6 + s] = r, nvld
                d[10 + s] = gi
                d[12 + s] = groups[gi][3] - groups[gi][2]
    return plan.reshape(-1), G


def describe(plan, grid):
    """Counts for logs / tests: (edge workgroups, vertex workgroups, max edge tiles per wavefront, max vertex tiles)."""
    p = np.asarray(plan).reshape(grid, WAVES, DESC)
    role = p[:, 0, 0]
    return (int((role == 1).sum()), int((role == 2).sum()), int(p[role == 1][:, :, 1].max(initial=0)),
            int(p[role == 2][:, :, 1].max(initial=0)))
