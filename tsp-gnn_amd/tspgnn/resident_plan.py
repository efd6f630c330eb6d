"""Host-side work plan of tspgnn_mp_resident_h2 (include/tspgnn.h, csrc/mp_resident_h2.hip): the whole T-step loop of
graphnn.py:175-179 as one launch of resident workgroups, edge states through memory, work handed out by an LDS ticket.

EV is block-diagonal by instance (instance_loader.py:56-66).  As for tspgnn_mp_loop_h2 (loop_plan.py, whose grouping this
plan shares) the batch is cut into GROUPS of consecutive instances -- the unit of synchronisation -- whose edge / vertex
rows are cut into 16-row tiles starting at the group's first row.  On top of that:

  * groups are dealt to the 8 XCDs in contiguous, edge-balanced ranges (workgroup b runs on XCD b mod 8: speed only);
  * the groups of an XCD are cut into CLASSES (two contiguous halves by edge tiles).  An edge workgroup holds a slice of
    EVERY class, and its items of one step are ordered [shares of class 0][tiles of class 0][shares of class 1][tiles of
    class 1]: the vertex chain of a class (row-sum -> vertex cells -> message MLP -> projection, which the class's own
    tiles of the next step but one wait for) runs while the workgroup's wavefronts are busy with the other class;
  * the vertex tiles of a class go to that class's CELL workgroups (role 2) and, once more, to its MESSAGE workgroups
    (role 3), <= 2 tiles per wavefront;
  * the V<-E row-sum of a group is cut into SHARES of <= SHARE_ROWS vertex rows, dealt to the edge workgroups that hold the
    group's tiles in proportion to the tiles they hold.  (Dealt to the class's VERTEX workgroups instead -- their wavefronts
    idle through most of a step -- the step got LONGER, 59 vs 42 us at C2: 1.7 shares of ~13 us + 1.7 tiles per vertex
    wavefront are a serial 40 us, and rounds of shares and tiles of different groups can wait for each other in a cycle.)

Layout (int32): grid headers of HDR ints, then items of ITEM ints -- documented at the top of csrc/mp_resident_h2.hip.
"""
import os

import numpy as np

from . import loop_plan

WAVES = 12
HDR = 8
ITEM = 8
N_XCD = 8
VERT_TILES = 2        # per wavefront
SHARE_ROWS = int(os.environ.get("TSPGNN_RES_SHARE_ROWS", "8"))   # vertex rows per row-sum item (development override: needs a library built to match)
SHARE_CAP = 48        # edge ids per vertex row of a share kept in LDS (TSPGNN_RESIDENT_SHARE_CAP)
SHARE_BLOCK = SHARE_ROWS + SHARE_ROWS * SHARE_CAP // 2   # LDS words of a share's block: counts, then 16-bit edge offsets
LDS_WORD_LIMIT = 11000  # LDS words per edge workgroup behind 113 KB of weights: one per tile + the shares' blocks

_cache = {}


def n_classes():
    env = os.environ.get("TSPGNN_RES_CLASSES")
    return max(1, min(4, int(env))) if env else 2


def cell_tiles_per_wg():
    """Vertex tiles a vertex CELL workgroup takes where the XCD can afford it (a wavefront's second tile doubles its part of
    the vertex chain; more workgroups are taken from the edge side).  TSPGNN_RES_CELL_TILES; default: one per wavefront."""
    env = os.environ.get("TSPGNN_RES_CELL_TILES")
    return max(1, min(WAVES * VERT_TILES, int(env))) if env else WAVES * VERT_TILES


def msg_tiles_per_wg():
    """The same for the vertex MESSAGE workgroups (TSPGNN_RES_MSG_TILES; default: two per wavefront -- a message tile is
    the shorter chain)."""
    env = os.environ.get("TSPGNN_RES_MSG_TILES")
    return max(1, min(WAVES * VERT_TILES, int(env))) if env else WAVES * VERT_TILES


def in_auto_window(edge_rows):
    """Does the default selector (graphnn.choose_loop_plan, TSPGNN_LOOP_KIND unset) hand a batch of this many edge rows to
    tspgnn_mp_resident_h2?  Measured on MI355X at n = 40, T = 32 against the stepwise launches (profiles/r06_resident_vs_steps.txt,
    one box, three alternating runs each): 32 / 64 instances +13 % / +10 % (the register-resident loop's ground), 96: equal,
    128 (C2, 99 840 rows): -3 %..+2 % -- inside the box-to-box spread, so C2 stays on the launches whose row-sum the roofline
    line is quoted on --, 160: -1.5 %, 192 (149 760 rows): -16 %, 224: -16 %, 256 (199 680 rows): -6 %.  The window is the
    stretch where the gain is outside the noise; TSPGNN_LOOP_KIND=resident takes every batch the kernel holds."""
    return 140000 <= edge_rows <= 205000


def share_lead():
    """Items by which a class's row-sum shares are moved AHEAD of the class's first tile (into the tail of the previous
    class's segment): earlier shares shorten the vertex chain's start, at the price of wavefronts that wait for the last
    message tiles.  Development switch (TSPGNN_RES_SHARE_LEAD); default 0."""
    env = os.environ.get("TSPGNN_RES_SHARE_LEAD")
    return max(0, int(env)) if env else 0


def build(e_start, v_start, grid=256, classes=None, lead=None):
    """-> dict(plan=int32[...], n_groups, n_slots, lds_words, edge_wgs, vertex_wgs, items_per_step_max) or None when the
    batch does not fit the design (no edges, fewer than 16 workgroups, more vertex tiles than the vertex workgroups of an
    XCD hold)."""
    e_start = np.asarray(e_start, dtype=np.int64)
    v_start = np.asarray(v_start, dtype=np.int64)
    if grid < N_XCD * 2 or grid % N_XCD != 0 or e_start[-1] == 0:
        return None
    if classes is None:
        classes = n_classes()
    if lead is None:
        lead = share_lead()
    key = (e_start.tobytes(), v_start.tobytes(), grid, classes, lead, cell_tiles_per_wg(), msg_tiles_per_wg())
    if key in _cache:
        return _cache[key]
    out = _build(e_start, v_start, grid, classes, lead)
    if len(_cache) > 64:
        _cache.clear()
    _cache[key] = out
    return out


def _split_classes(gs, et, k):
    """Cut the group list gs into <= k contiguous, non-empty parts balanced by edge tiles."""
    k = min(k, len(gs))
    if k <= 1:
        return [list(gs)]
    cum = np.cumsum(et[gs])
    parts, start = [], 0
    for c in range(1, k):
        cut = int(np.searchsorted(cum, cum[-1] * c / k, side="left")) + 1
        cut = max(cut, start + 1)
        cut = min(cut, len(gs) - (k - c))
        parts.append(list(gs[start:cut]))
        start = cut
    parts.append(list(gs[start:]))
    return [p for p in parts if p]


def _build(e_start, v_start, grid, classes, lead):
    groups = loop_plan._groups(e_start, v_start)   # (e0, e1, v0, v1)
    G = len(groups)
    et = np.array([(g[1] - g[0] + 15) // 16 for g in groups], dtype=np.int64)
    vt = np.array([(g[3] - g[2] + 15) // 16 for g in groups], dtype=np.int64)
    if G == 0 or np.any(et < 1):
        return None
    wg_per_xcd = grid // N_XCD
    cum = np.cumsum(et)
    mid = cum - et / 2.0
    xcd_of = np.minimum((mid * N_XCD / cum[-1]).astype(np.int64), N_XCD - 1)
    hdr = np.zeros((grid, HDR), dtype=np.int32)
    items = []          # rows of ITEM ints
    slot = 0
    lds_words = 0
    n_edge = n_vert = 0
    w_max = 0
    for x in range(N_XCD):
        gs = np.nonzero(xcd_of == x)[0]
        if len(gs) == 0:
            continue
        # classes, and vertex workgroups per class: cell workgroups (K resident) and message workgroups (message MLP +
        # projection resident), each wavefront <= 2 tiles; fewer tiles per workgroup where the XCD can afford it, fewer
        # classes where it cannot
        nv = None
        for k_cls in range(classes, 0, -1):
            parts = _split_classes(gs, et, k_cls)
            for per_cell, per_msg in ((cell_tiles_per_wg(), msg_tiles_per_wg()), (WAVES * VERT_TILES, WAVES * VERT_TILES)):
                nc = [max(1, -(-int(vt[p].sum()) // per_cell)) for p in parts]
                nm = [max(1, -(-int(vt[p].sum()) // per_msg)) for p in parts]
                if sum(nc) + sum(nm) <= max(2, wg_per_xcd // 3) and sum(nc) + sum(nm) < wg_per_xcd:
                    nv = [x_ + y_ for x_, y_ in zip(nc, nm)]
                    break
            if nv is not None:
                break
        if nv is None:
            return None
        ne = wg_per_xcd - sum(nv)
        if ne < 1:
            return None
        wgs = [x + N_XCD * s for s in range(wg_per_xcd)]
        edge_wgs, rest = wgs[:ne], wgs[ne:]
        # tiles per class, in group order
        ctiles = []
        for p in parts:
            tl = []
            for gi in p:
                e0, e1, _, _ = groups[gi]
                for r in range(e0, e1, 16):
                    tl.append((r, min(16, e1 - r), int(gi)))
            ctiles.append(tl)
        # slices of every class per edge workgroup
        slices = [[ctiles[c][len(ctiles[c]) * k // ne: len(ctiles[c]) * (k + 1) // ne] for c in range(len(parts))]
                  for k in range(ne)]
        # row-sum shares: group -> [(workgroup index k, tiles held)]
        holders = {}
        for k in range(ne):
            for c in range(len(parts)):
                for (_, _, gi) in slices[k][c]:
                    holders.setdefault(gi, {}).setdefault(k, 0)
                    holders[gi][k] += 1
        shares = [[[] for _ in parts] for _ in range(ne)]   # [k][c] -> [(v0, v1, gi)]
        for c, p in enumerate(parts):
            for gi in p:
                hs = sorted(holders[int(gi)].items())
                tot = sum(n for _, n in hs)
                v0, v1 = groups[gi][2], groups[gi][3]
                acc = 0
                for k, n in hs:
                    a = v0 + (v1 - v0) * acc // tot
                    acc += n
                    b = v0 + (v1 - v0) * acc // tot
                    for s in range(a, b, SHARE_ROWS):
                        shares[k][c].append((s, min(b, s + SHARE_ROWS), int(gi)))
        for k, b in enumerate(edge_wgs):
            seq = []   # the step's items in ticket order
            local = 0
            g_first = None
            for c in range(len(parts)):
                sh = [(s0, s1, gi, -1, int(et[gi]), 0, int(groups[gi][0])) for (s0, s1, gi) in shares[k][c]]
                tl = []
                for (r, nvld, gi) in slices[k][c]:
                    tl.append((r, nvld, gi, local, int(vt[gi])))
                    local += 1
                    g_first = gi if g_first is None else min(g_first, gi)
                # shares `lead` items ahead of the class's first tile
                cut = max(0, len(seq) - lead) if lead > 0 else len(seq)
                seq = seq[:cut] + sh + seq[cut:] + tl
            if not seq:
                continue
            if local > LDS_WORD_LIMIT:
                return None
            n_edge += 1
            hdr[b] = [1, len(items), len(seq), slot, g_first if g_first is not None else 0, local, 0, 0]
            words = local
            for it in seq:
                row = list(it) + [0] * (ITEM - len(it))
                if it[3] == -1:    # a share: its edge lists wait in LDS while there is room (else the kernel's general loop)
                    if words + SHARE_BLOCK <= LDS_WORD_LIMIT and groups[it[2]][1] - groups[it[2]][0] < 65536:
                        row[5] = words - local
                        words += SHARE_BLOCK
                    else:
                        row[5] = -1
                items.append(row)
            slot += local
            lds_words = max(lds_words, words)
            w_max = max(w_max, len(seq))
        # vertex tiles per class -> that class's cell workgroups, and again -> its message workgroups
        pos = 0
        for c, p in enumerate(parts):
            vl = []
            for gi in p:
                _, _, v0, v1 = groups[gi]
                for r in range(v0, v1, 16):
                    vl.append((r, min(16, v1 - r), int(gi), -2, 0))
            for role, n_wg in ((2, nc[c]), (3, nm[c])):
                mine = rest[pos:pos + n_wg]
                pos += n_wg
                for k, b in enumerate(mine):
                    tl = vl[len(vl) * k // len(mine): len(vl) * (k + 1) // len(mine)]
                    if len(tl) > WAVES * VERT_TILES:
                        return None
                    if not tl:
                        continue
                    n_vert += 1
                    hdr[b] = [role, len(items), len(tl), 0, 0, 0, c, 0]
                    for (r, nvld, gi, tag, _) in tl:
                        # [4]: what the tile waits for -- a cell tile for its group's vertex ROWS (aggregated), a message tile
                        # for its group's vertex TILES (updated)
                        cnt = groups[gi][3] - groups[gi][2] if role == 2 else int(vt[gi])
                        items.append([r, nvld, gi, tag, cnt, 0, 0, 0])
    plan = np.concatenate([hdr.reshape(-1), np.asarray(items, dtype=np.int32).reshape(-1)]).astype(np.int32)
    return dict(plan=plan, n_groups=G, n_slots=max(slot, 1), lds_words=lds_words, edge_wgs=n_edge, vertex_wgs=n_vert,
                items_per_step_max=w_max, grid=grid)


def decode(built):
    """(headers [grid, HDR], items [n, ITEM]) of a built plan (tests, logs)."""
    grid = built["grid"]
    p = np.asarray(built["plan"])
    return p[:grid * HDR].reshape(grid, HDR), p[grid * HDR:].reshape(-1, ITEM)
