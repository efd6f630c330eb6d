#!/usr/bin/env python
"""Times tspgnn_mlp_bwd_rc_h2 alone at the C2 shard's shape (99 840 edge rows of 128 complete graphs on 40 vertices,
three pushed layers of width 64): python tools/rc_bench.py [reps]   (TSPGNN_LIB=tools/variants/X.so for A/B builds)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tsp-gnn_amd"))
from tspgnn import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
B, n, d, L = 128, 40, 64, 3
iu = np.triu_indices(n, 1)
uv = np.concatenate([np.stack([iu[0] + b * n, iu[1] + b * n], 1) for b in range(B)]).astype(np.int32)
rows, n_src = uv.shape[0], B * n
g = torch.Generator().manual_seed(0)
X = torch.randn((rows, d), generator=g).to(dev)
dYs = (1e-4 * torch.randn((n_src, d), generator=g)).to(dev)
uvd = torch.from_numpy(uv).to(dev)
scale = _lib.lib.tspgnn_h2_weight_scale()
Ws = [(torch.randn((d, d), generator=g) / 8.0).to(dev) for _ in range(L)]
bs = [(0.1 * torch.randn(d, generator=g)).to(dev) for _ in range(L)]
per = 4 * d * d + 4 * d
wb = torch.empty(L * per, dtype=torch.uint8, device=dev)
wt = torch.empty(L * 4 * d * d, dtype=torch.uint8, device=dev)
for j in range(L):
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(Ws[j]), _lib.ptr(wb[j * per:]), d, d, None, None)
    wb[j * per + 4 * d * d:(j + 1) * per].copy_((scale * bs[j]).view(torch.uint8))
    _lib.call("tspgnn_pack_weights_h2", _lib.ptr(Ws[j].t().contiguous()), _lib.ptr(wt[j * 4 * d * d:]), d, d, None, None)
Y = torch.empty((rows, d), device=dev)
task = _lib.MlpTask(_lib.ptr(X), _lib.ptr(wb), _lib.ptr(Y), None, 0, rows, L, 7, None, None, None)
_lib.call_multi("tspgnn_mlp_fwd_multi_h2", [task], d)
part = torch.zeros(int(_lib.lib.tspgnn_mlp_bwd_rc_partial_floats(d, L)), device=dev)
dX = torch.zeros((rows, d), device=dev)
t = _lib.MlpBwdRcTask(_lib.ptr(X), _lib.ptr(wb), _lib.ptr(wt), _lib.ptr(Y), _lib.ptr(dYs), _lib.ptr(uvd), _lib.ptr(dX), 1, rows, L, 7,
                      None, 0, None, 0, _lib.ptr(part))     # (weight gradients in the launch: the one form kept, round 6)
tp = ctypes.cast(ctypes.pointer(t), ctypes.c_void_p)
for _ in range(5):
    _lib.call("tspgnn_mlp_bwd_rc_h2", tp, d, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.call("tspgnn_mlp_bwd_rc_h2", tp, d, None)
e1.record()
torch.cuda.synchronize()
print("mlp_bwd_rc_h2 %s: %.2f us per launch (rows %d, %d reps)" % (os.environ.get("TSPGNN_LIB", "in-tree"), 1e3 * e0.elapsed_time(e1) / reps, rows, reps))
