"""Development experiment (CPU): end-to-end error of the split-operand GEMM arithmetics against the float64 oracle.

  f32    plain float32 matmul (what the reference's fp32 graph does)
  x3     three bf16 pieces per operand, six products (dense_x3.hip)
  h2     two fp16 pieces per operand, three products  hi*hi + hi*lo + lo*hi  (dense_h2.hip)
  h2s    h2 with the weight operand scaled by 2^s before the split (lo piece out of the fp16 subnormals)

Usage: python tools/h2_numerics.py [n] [B] [T]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
from oracle import np_oracle as NO  # noqa: E402
from oracle import params as P  # noqa: E402


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16(x):
    a = to_bf16(x)
    r = (x - a).astype(np.float32)
    b = to_bf16(r)
    c = to_bf16((r - b).astype(np.float32))
    return a, b, c


def split_f16(x, flush=False):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    if flush:  # what an MFMA that flushed fp16 subnormal inputs would see
        tiny = np.float16(6.103515625e-05)
        hi = np.where(np.abs(hi) < tiny, np.float16(0), hi)
        lo = np.where(np.abs(lo) < tiny, np.float16(0), lo)
    return hi.astype(np.float32), lo.astype(np.float32)


class Arith(object):
    def __init__(self, kind, s=0, flush=False):
        self.kind, self.s, self.flush = kind, s, flush

    def mm(self, x, W):
        x = np.ascontiguousarray(x, dtype=np.float32)
        W = np.ascontiguousarray(W, dtype=np.float32)
        if self.kind == "f32":
            return x @ W
        if self.kind == "x3":
            a1, a2, a3 = split_bf16(x)
            b1, b2, b3 = split_bf16(W)
            return ((a1 @ b3 + a2 @ b2 + a3 @ b1) + (a1 @ b2 + a2 @ b1) + a1 @ b1).astype(np.float32)
        sc = np.float32(2.0 ** self.s)
        a1, a2 = split_f16(x, self.flush)
        b1, b2 = split_f16(W * sc, self.flush)
        y = (a1 @ b2 + a2 @ b1) + a1 @ b1
        return (y / sc).astype(np.float32)


def forward_with(arith, params, batch, T):
    """np_oracle.forward in float32 with every dense product routed through `arith`."""
    saved = (NO.dense, NO.lnlstm)

    def dense(x, W, b, act):
        y = arith.mm(x, W) + b
        return NO.relu(y) if act else y

    def lnlstm(x, h, c, K, ln, z0=None):
        d = h.shape[1]
        z = arith.mm(np.concatenate([x, h], axis=1), K)
        i, j, f, o = z[:, :d], z[:, d:2 * d], z[:, 2 * d:3 * d], z[:, 3 * d:]
        i = NO.layer_norm(i, *ln["input"])
        j = NO.layer_norm(j, *ln["transform"])
        f = NO.layer_norm(f, *ln["forget"])
        o = NO.layer_norm(o, *ln["output"])
        nc = c * NO.sigmoid(f + np.float32(1.0)) + NO.sigmoid(i) * NO.relu(j)
        nc = NO.layer_norm(nc, *ln["state"])
        return NO.relu(nc) * NO.sigmoid(o), nc

    NO.dense, NO.lnlstm = dense, lnlstm
    try:
        return NO.forward(params, batch, T, dtype=np.float32)
    finally:
        NO.dense, NO.lnlstm = saved


def synthetic(n, B, seed):
    rng = np.random.RandomState(seed)
    uv, W, C, ne, nv, lab = [], [], [], [], [], []
    off = 0
    for b in range(B):
        pts = rng.rand(n, 2)
        iu = np.triu_indices(n, 1)
        uv.append(np.stack(iu, axis=1) + off)
        W.append(np.sqrt(((pts[iu[0]] - pts[iu[1]]) ** 2).sum(-1)))
        cost = 0.7124 * np.sqrt(n) / n
        C.append(np.full(len(iu[0]), cost * (0.98 if b % 2 == 0 else 1.02)))
        ne.append(len(iu[0]))
        nv.append(n)
        lab.append(b % 2)
        off += n
    return {"ev_uv": np.concatenate(uv).astype(np.int32), "W": np.concatenate(W), "C": np.concatenate(C),
            "n_edges": np.array(ne), "n_vertices": np.array(nv), "route_exists": np.array(lab, dtype=np.float64)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    batch = synthetic(n, B, 0)
    for seed, perturb in ((0, False), (1, True)):
        params = P.init_params(64, seed=seed, perturb=perturb)
        ref = NO.forward(params, batch, T, dtype=np.float64)

        def rel(a, b):
            return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())

        print("params seed=%d perturb=%s  n=%d B=%d T=%d" % (seed, perturb, n, B, T))
        for name, ar in (("f32", Arith("f32")), ("x3", Arith("x3")), ("h2", Arith("h2")), ("h2 flush", Arith("h2", 0, True)),
                         ("h2s s=3", Arith("h2", 3)), ("h2s s=6", Arith("h2", 6)), ("h2s s=6 flush", Arith("h2", 6, True))):
            out = forward_with(ar, params, batch, T)
            print("  %-14s pred %.2e  E.h %.2e  E.c %.2e  V.h %.2e  V.c %.2e" % (
                name, rel(out["predictions"], ref["predictions"]), rel(out["E"][0], ref["E"][0]),
                rel(out["E"][1], ref["E"][1]), rel(out["V"][0], ref["V"][0]), rel(out["V"][1], ref["V"][1])))
