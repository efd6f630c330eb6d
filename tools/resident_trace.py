"""Where the memory-resident one-launch loop (tspgnn_mp_resident_h2) spends its time: per-wavefront sums of s_memrealtime
ticks (100 MHz) per phase, averaged per step.  python tools/resident_trace.py [graphs=128] [n=40] [T=32]
edge wavefronts:   0 set-up (once)  1 ticket + item fetch  2 share: wait for the group's message tiles  3 share: row-sum, drain, arrive
                   4 tile: wait for its own previous step (LDS)  5 tile: wait for the projected messages (+ L1 invalidate)
                   6 tile: gather, h Kh, gates, state stores  7 tile: message MLP, stores, drain, publish
cell wavefronts:   0 set-up (once)  1 h fetch + wait for the aggregates + their fetch  2 GEMM, gates, stores  3 drain + arrive
message wavefronts: 0 set-up (once)  1 wait for the updated tiles  2 h' fetch, MLP, projection, stores  3 drain + arrive"""
import os
import sys

os.environ["TSPGNN_LOOP_TRACE"] = "1"
os.environ["TSPGNN_LOOP_KIND"] = "resident"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tsp-gnn_amd")):
    sys.path.insert(0, p)
import tspgnn  # noqa: E402
from oracle import params as P  # noqa: E402
from tspgnn import resident_plan as RP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
t = tspgnn.synthetic_batch([n] * B, seed=0)
params = P.init_params(64, seed=1, perturb=True)
model = tspgnn.build_network(64)
sess = tspgnn.Session(model)
sess.run(tspgnn.global_variables_initializer())
model.store.load(params)
EV, W, C, r, nv, ne = t
feed = {model["EV"]: EV, model["W"]: W, model["C"]: C, model["time_steps"]: T, model["route_exists"]: r,
        model["n_vertices"]: nv, model["n_edges"]: ne}
b = sess.prepare(feed)
for _ in range(3):
    sess.forward_device(b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sess.forward_device(b)
e1.record()
torch.cuda.synchronize()
print("eager forward with trace: %.3f ms" % e0.elapsed_time(e1))
full = model["gnn"].loop_trace.cpu().numpy().astype(np.float64)
tr = full[:, :, :16] * 0.01 / T     # us per step
plan, G, grid, kind, n_slots, lds_words, n_active = b.adj.loop_plan
assert kind == "resident"
hdr = plan.cpu().numpy()[:grid * RP.HDR].reshape(grid, RP.HDR)
print("\n".join(__doc__.split("\n")[2:]))
for role, name in ((1, "edge"), (2, "cell"), (3, "message")):
    sel = (hdr[:, 0] == role) & (hdr[:, 2] > 0)
    x = tr[sel].reshape(-1, 16)
    if len(x) == 0:
        continue
    print("%s workgroups: %d, items per step %s" % (name, int(sel.sum()), sorted(set(hdr[sel, 2].tolist()))))
    print("  phase      " + " ".join("%7d" % i for i in range(8)) + "    total")
    for label, v in (("mean", x.mean(0)), ("p10", np.percentile(x, 10, axis=0)), ("p90", np.percentile(x, 90, axis=0)),
                     ("max", x.max(0))):
        print("  %-9s  " % label + " ".join("%7.2f" % a for a in v[:8]) + "  %7.2f" % v.sum())

# timeline of step T/2 (absolute 100 MHz stamps in slots 8..15), relative to the earliest stamp
raw = full[:, :, 8:16] * 0.01
t0 = raw[raw > 0].min()
names_e = ["item taken", "own step ready", "Zx ready", "tile done", "share taken", "share: msgs ready", "share done", "-"]
names_c = ["aggregates ready", "-", "h' handed over", "-", "-", "-", "-", "-"]
names_m = ["-", "-", "-", "h' ready", "-", "Zx arrived", "-", "-"]
for role, names in ((1, names_e), (2, names_c), (3, names_m)):
    sel = (hdr[:, 0] == role) & (hdr[:, 2] > 0)
    x = raw[sel].reshape(-1, 8)
    print("step %d timeline (%s), us after the first stamp: min / median / max over wavefronts" % (T // 2, {1: "edge", 2: "cell", 3: "message"}[role]))
    for i, nm in enumerate(names):
        v = x[:, i][x[:, i] > 0] - t0
        if len(v):
            print("  %-20s %7.2f %7.2f %7.2f   (%d)" % (nm, v.min(), np.median(v), v.max(), len(v)))

# one XCD, workgroup by workgroup: the stamps of step T/2 (us after the XCD's first stamp), min..max over the workgroup's wavefronts
if os.environ.get("RES_TRACE_XCD") is not None:
    x = int(os.environ["RES_TRACE_XCD"])
    rawx = raw.reshape(grid, -1, 8)
    sel = [b for b in range(grid) if b % 8 == x and hdr[b, 0] > 0 and hdr[b, 2] > 0]
    tx = min(rawx[b][rawx[b] > 0].min() for b in sel)
    for b in sel:
        role = int(hdr[b, 0])
        names = {1: names_e, 2: names_c, 3: names_m}[role]
        parts = []
        for i, nm in enumerate(names):
            v = rawx[b][:, i]
            v = v[v > 0] - tx
            if len(v) and nm != "-":
                parts.append("%s %.1f..%.1f" % (nm, v.min(), v.max()))
        print("wg %3d role %d class %d items %2d | " % (b, role, hdr[b, 6], hdr[b, 2]) + " | ".join(parts))


# inside ONE tile (the last tile a wavefront ran in step T/2): stamps 16..24 = item decoded, own previous step ready, Zx ready,
# loads issued, loads landed (an extra vmcnt(0) in the trace build), K GEMM done, gates done, state stores issued, MLP done + message stores issued
fine = full[:, :, 16:25] * 0.01
sel = (hdr[:, 0] == 1) & (hdr[:, 2] > 0)
f = fine[sel].reshape(-1, 9)
f = f[(f > 0).all(1)]
names = ["own step ready", "Zx ready", "loads issued", "loads landed", "K GEMM done", "gates done", "state stored", "MLP + msg stores"]
print("inside one edge tile of step %d, us between consecutive stamps (median / p10 / p90 over %d wavefronts):" % (T // 2, len(f)))
for i, nm in enumerate(names):
    dlt = f[:, i + 1] - f[:, i]
    print("  -> %-18s %6.2f %6.2f %6.2f" % (nm, np.median(dlt), np.percentile(dlt, 10), np.percentile(dlt, 90)))
tot = f[:, 8] - f[:, 0]
print("  whole tile          %6.2f %6.2f %6.2f" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
