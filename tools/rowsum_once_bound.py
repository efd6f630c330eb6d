"""Upper bound on what an edge-once V<-E row-sum could gain at n >= 64: the shipped CSR kernel timed
on (a) the full incidence lists (every edge row read by both endpoints: what ships), (b) the ROW half only (vertex u
lists its edges (u, v > u): every edge row read exactly once, contiguous runs) and (c) the COLUMN half only (strided
single rows).  (b) is the floor of any formulation that reads each row once -- without the partial sums such a kernel
has to write and combine on top.  X is rewritten before every launch (as the message MLP does in the loop).
python tools/rowsum_once_bound.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
from tspgnn import _lib  # noqa: E402


def lists(sizes):
    full, upper, lower = [], [], []
    e0 = 0
    for n in sizes:
        iu, iv = np.triu_indices(n, 1)
        order_u, order_v = np.argsort(iu, kind="stable"), np.argsort(iv, kind="stable")
        cu, cv = np.searchsorted(iu[order_u], np.arange(n + 1)), np.searchsorted(iv[order_v], np.arange(n + 1))
        for v in range(n):
            row = order_u[cu[v]:cu[v + 1]] + e0
            col = order_v[cv[v]:cv[v + 1]] + e0
            upper.append(row)
            lower.append(col)
            full.append(np.sort(np.concatenate([col, row])))
        e0 += iu.size
    return e0, full, upper, lower


def csr(lst, dev):
    rowptr = np.zeros(len(lst) + 1, np.int32)
    rowptr[1:] = np.cumsum([len(x) for x in lst])
    eid = np.concatenate(lst).astype(np.int32)
    return torch.from_numpy(rowptr).to(dev), torch.from_numpy(eid).to(dev)


def run(name, sizes, d, dtype, fn):
    dev = torch.device("cuda:0")
    M, full, upper, lower = lists(sizes)
    N = int(sum(sizes))
    X0 = torch.randn(M, d, device=dev).to(dtype)
    X = torch.empty_like(X0)
    Y = torch.empty(N, d, device=dev, dtype=dtype)
    out = []
    for label, lst in (("full (ships)", full), ("row half: each edge once", upper), ("column half: strided", lower)):
        rp, ei = csr(lst, dev)
        ts = []
        for it in range(30):
            X.copy_(X0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call(fn, _lib.ptr(rp), _lib.ptr(ei), _lib.ptr(X), _lib.ptr(Y), N, M, d, _lib.current_stream())
            e1.record()
            torch.cuda.synchronize()
            if it >= 5:
                ts.append(1e3 * e0.elapsed_time(e1))
        out.append((label, float(np.median(ts)), int(ei.numel())))
    row_bytes = d * X0.element_size()
    print("== %s: %d graphs, n %d..%d, d=%d %s, %d edge rows of %d B (%.0f MB)" % (name, len(sizes), min(sizes), max(sizes), d, str(dtype)[6:], M, row_bytes, M * row_bytes / 1e6))
    for label, us, reads in out:
        print("  %-28s %7.1f us   %d row reads = %.0f MB -> %.2f TB/s" % (label, us, reads, reads * row_bytes / 1e6,
                                                                           reads * row_bytes / us / 1e6))


if __name__ == "__main__":
    run("C4", list(np.random.RandomState(0).randint(20, 81, size=512)), 64, torch.float32, "tspgnn_csr_rowsum_f32")
    run("C5 shard", [200] * 32, 128, torch.bfloat16, "tspgnn_csr_rowsum_bf16")
    run("C2", [40] * 128, 64, torch.float32, "tspgnn_csr_rowsum_f32")
