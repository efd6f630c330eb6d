#!/usr/bin/env python
"""Bytes-and-roof column for the kernels of the C2 training step: reads the committed rocprofv3 summary
profiles/<round>_c2_train_kernel_stats.txt and prints, per kernel that matters, the ALGORITHMIC bytes of one launch (formula
stated per line; C2: M = 99 840 edge rows, N = 5 120 vertex rows, d = 64, T = 32; fp32 = 4 B), the rate at the summary's average
duration and the fraction of the 8 TB/s roof.  python tools/train_roofs.py [r06] > profiles/r06_c2_train_kernel_roofs.txt"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
M, N, d, T = 99840, 5120, 64, 32
R = d * 4                      # one fp32 row of d
rows = []
for line in open(os.path.join(ROOT, "profiles", "%s_c2_train_kernel_stats.txt" % rnd)):
    if line.startswith("#") or line.startswith("kernel"):
        if line.startswith("# csrc_sha16"):
            sha = line.strip()
        continue
    f = line[80:].split()
    if len(f) >= 9:
        rows.append((line[:80].strip(), int(f[0]), int(f[4]), float(f[5]), float(f[6]), float(f[7]), float(f[8])))
steps = 8      # training steps in the profiled run (5 timed + 2 warm-up + 1 capture)

# (pattern, grid or None, bytes per launch, formula)
SPEC = [
    ("lnlstm_bwd_h2_kernel", None, M * R * (4 + 4 + 2) + N * 4 * R + M * 8 + N * R * (5 + 4 + 1 + 4 + 2),
     "E cell: h, c, dh', dc' read + dz [M,4d] + dc, dh written (10 rows of d per edge row) + Zx [N,4d] + uv; V cell: x, h, c, dh', dc' + dz + dc, and its second phase: dz back in, [d(agg) | dh] out"),
    ("tspgnn::mlp_bwd_h2_kernel", None, M * R * (3 + 3 + 2) + N * R * (4 + 3 + 4 + 2),
     "E_msg_V (pushed: 3 layers on M rows): 3 saved activations read, 3 dpre written, dh read + written; d(agg) gathered from [N,d]; V_msg_E on N rows: dZx [N,4d] in (projection head), 3 activations, 4 dpre, dh"),
    ("lnlstm_mlp_fwd_h2_kernel<64, false", None, M * R * (2 + 2 + 3) + N * 4 * R + M * 8 + N * R * (3 + 2 + 3),
     "training forward cell + next messages: h, c read, h', c' written, 3 taped activations written per edge row; Zx gather; vertex side alike"),
    ("csr_rowsum_kernel<16, false, 4>", None, M * 4 * R + N * 4 * R + (2 * M + N + 1) * 4,
     "EV^T dz: dz [M,4d] read once, [N,4d] written, CSR"),
    ("csr_rowsum_kernel<16, false, 1>", None, M * R + N * R + (2 * M + N + 1) * 4, "V<-E row-sum of a [M,d] array"),
    ("gather2_sum_kernel", None, M * R + N * R + M * 8, "E<-V gather (adjoint of the row-sum) into a [M,d] array"),
    ("tspgnn::mlp_fwd_h2_kernel", None, M * R * (1 + 3) + N * R * (1 + 4) + N * 4 * R,
     "message MLPs of a step, taping: h read, 3 activations written (E side, pushed); V side 4 layers + the Kx projection [N,4d]"),
    ("wgrad_x3_kernel<false, 1>", 131072, None, "per call: X [rows,kin] + dY [rows,nout] read once; the four calls per step are four DIFFERENT reductions"),
    ("wgrad_x3_kernel<false, 1>", 130560, T * M * R * 2, "dW of one message-MLP layer on the edge rows: a_l [T M, d] and dpre_l [T M, d]"),
]
print(sha)
print("# python tools/train_roofs.py %s: algorithmic bytes per launch, rate at the rocprofv3 average duration, fraction of 8 TB/s" % rnd)
print("# (profiles/%s_c2_train_kernel_stats.txt; C2: M = %d, N = %d, d = %d, T = %d)" % (rnd, M, N, d, T))
print("%-44s %9s %8s %10s %8s %6s  %s" % ("kernel", "calls/step", "avg_us", "MB/launch", "TB/s", "frac", "bytes counted"))
for pat, grid, nbytes, what in SPEC:
    for name, g, calls, avg, mn, mx, pct in rows:
        if pat in name and (grid is None or g == grid):
            if nbytes is None:       # the shared-grid wgrad calls: list the two extremes
                big = T * M * R + T * M * 4 * R          # h^T dz of the edge cell
                small = T * N * 2 * R + T * N * 4 * R    # [x | h]^T dz of the vertex cell
                print("%-44s %9.1f %8.1f %10s %8s %6s  %s" % (name.replace('void ', '').replace('tspgnn::', '')[:44], calls / steps, avg, "-", "-", "-", what))
                print("%-44s %9s %8.1f %10.1f %8.2f %6.2f  %s" % ("  its longest call (max_us)", "1", mx, big / 1e6, big / mx / 1e6, big / mx / 1e6 / 8,
                                                                 "h^T dz of the edge cell: h [T M, d] + dz [T M, 4d] = 3.19 M rows"))
                print("%-44s %9s %8.1f %10.1f %8.2f %6.2f  %s" % ("  its shortest call (min_us)", "1", mn, small / 1e6, small / mn / 1e6, small / mn / 1e6 / 8,
                                                                 "[x | h]^T dz of the vertex cell: 164 k rows"))
            else:
                print("%-44s %9.1f %8.1f %10.1f %8.2f %6.2f  %s" % (name.replace('void ', '').replace('tspgnn::', '')[:44], calls / steps, avg, nbytes / 1e6, nbytes / avg / 1e6,
                                                                   nbytes / avg / 1e6 / 8, what))
            break
tot = sum(c * a for _, _, c, a, _, _, _ in rows) / steps / 1e3
print("# sum of all kernels: %.2f ms per training step under the profiler (bench line of the same set: profiles/%s_c2_train_bench.json)" % (tot, rnd))
