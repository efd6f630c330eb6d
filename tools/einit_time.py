import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tsp-gnn_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import tspgnn
from tspgnn import _lib
for d, M in ((64, 99840), (64, 695849), (128, 636800)):
    rng = np.random.RandomState(0)
    dims = [2, d // 8, d // 4, d // 2, d]
    wb = np.concatenate([np.concatenate([rng.randn(a, b).astype(np.float32).reshape(-1), (0.3 * rng.randn(b)).astype(np.float32)]) for a, b in zip(dims[:-1], dims[1:])])
    dev = torch.device("cuda")
    WC = torch.rand((M, 2), device=dev); dE0 = torch.randn((M, d), device=dev); wbd = torch.from_numpy(wb).to(dev)
    dwb = torch.zeros(wb.size, device=dev)
    ws = _lib.workspace("tspgnn_einit_bwd_workspace_floats", M, d, device=dev)
    for _ in range(3):
        _lib.call("tspgnn_einit_bwd_f32", _lib.ptr(WC), _lib.ptr(wbd), _lib.ptr(dE0), _lib.ptr(dwb), _lib.ptr(ws), M, d, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        _lib.call("tspgnn_einit_bwd_f32", _lib.ptr(WC), _lib.ptr(wbd), _lib.ptr(dE0), _lib.ptr(dwb), _lib.ptr(ws), M, d, None)
    torch.cuda.synchronize(); print("einit_bwd d=%d M=%d: %.1f us" % (d, M, 1e5 * (time.perf_counter() - t0)))
