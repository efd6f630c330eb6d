#!/usr/bin/env python
"""Weight-gradient reductions at the training shapes (development aid, gpurun): us per call and the HBM rate of the
compulsory bytes (X once + dY once).   python tools/wgrad_bench.py [c2|c5]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tsp-gnn_amd"))
from tspgnn import _lib  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
shapes = {"c2": [(99840 * 32, 64, 256, False), (99840 * 32, 64, 64, False), (5120 * 32, 128, 256, False)],
          "c5": [(636800 * 8, 128, 512, True), (636800 * 8, 128, 128, True), (636800 * 8, 128, 512, False),
                 (6400 * 64, 256, 512, True)]}[which]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for rows, kin, nout, xb in shapes:
    X = torch.randn(rows, kin, device=dev)
    if xb:
        X = X.to(torch.bfloat16)
    dY = torch.randn(rows, nout, device=dev)
    gW = torch.zeros(kin, nout, device=dev)
    gb = torch.zeros(nout, device=dev)
    ws = _lib.workspace("tspgnn_wgrad_workspace_floats", rows, kin, nout, device=dev)
    name = "tspgnn_wgrad_bf16x_f32" if xb else "tspgnn_wgrad_f32"
    t = timeit(lambda: _lib.call(name, _lib.ptr(X), _lib.ptr(dY), rows, kin, nout, _lib.ptr(gW), _lib.ptr(gb), _lib.ptr(ws), None))
    nbytes = rows * (kin * (2 if xb else 4) + nout * 4)
    ref = X[:65536].float().T.double() @ dY[:65536].double()
    gW.zero_()
    _lib.call(name, _lib.ptr(X), _lib.ptr(dY), 65536, kin, nout, _lib.ptr(gW), _lib.ptr(gb), _lib.ptr(ws), None)
    err = float((gW.double() - ref).abs().max() / ref.abs().max())
    print("%-24s rows=%-9d %3dx%-3d  %8.1f us  %.2f TB/s   (err %.1e)" % (name, rows, kin, nout, t, nbytes / t / 1e6, err))
    del X, dY, ws
