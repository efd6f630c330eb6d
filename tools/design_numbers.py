#!/usr/bin/env python
"""Prints the numbers table of DESIGN.md section 9 from the committed round-6 bench lines and kernel summaries
(profiles/r06_*): python tools/design_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def bench(w):
    return json.load(open(os.path.join(P, "r06_%s_bench.json" % w)))


def kernel_us(w, pat, fname="forward_kernel_stats"):
    for line in open(os.path.join(P, "r06_%s_%s.txt" % (w, fname))):
        if re.search(pat, line):
            f = line[80:].split()
            return float(f[5]), int(f[4])
    return None, None


rows = []
for w, what in (("c1", "C1: 32 x n=20, d=64, T=8 (persistent loop)"), ("c2", "**C2: 128 x n=40, d=64, T=32** (headline)"),
                ("c4", "C4: 512 ragged n=20..80, d=64, T=32"), ("c5", "C5 shard: 32 x n=200, d=128, T=64, bf16 storage")):
    j = bench(w)
    r = j["roofline"]
    rc = j.get("roofline_cell") or {}
    t = j.get("train") or {}
    sv = j.get("serve") or {}
    rs, _ = kernel_us(w, "csr_rowsum")
    cell, _ = kernel_us(w, "lnlstm_mlp_fwd_h2_kernel|lnlstm_fwd_bf16_kernel|mp_loop_h2_kernel")
    alg = (r.get("algorithmic_bytes_per_launch") or {}).get("csr_rowsum")
    frac = (alg / (rs * 1e-6) / 8e12) if (rs and alg) else None
    rows.append("| %s | %.3f | %.0f | %s | %s | %s | %s | %s |" % (
        what, j["ms_per_step"], j["value"],
        "%.2f us = %.2f" % (rs, frac) if frac else "-",
        "%.0f us" % cell if cell else "-",
        ("%.2fx" % r["traffic_over_algorithmic"]) if r.get("traffic_over_algorithmic") else "-",
        ("%.2f" % t["ms_per_step"]) if t.get("ms_per_step") else "-",
        ("%.3f (%.2fx)" % (sv["ms_per_batch"], sv["vs_resident"])) if sv.get("ms_per_batch") else "-"))
print("| workload (one MI355X) | forward ms | mp-steps/s | V<-E row-sum launch (rocprof avg, frac of 8 TB/s on its algorithmic bytes) | "
      "cell (or loop) launch | row-sum PMC traffic / algorithmic | training step ms | fresh batches ms (vs resident) |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
c2t = json.load(open(os.path.join(P, "r06_c2_train_bench.json")))
c5t = json.load(open(os.path.join(P, "r06_c5_train_bench.json")))
print("\nTraining, dedicated runs: C2 %.2f ms per step (`r06_c2_train_bench.json`), C5 shard %.1f ms (`r06_c5_train_bench.json`)."
      % (c2t["ms_per_step"], c5t["ms_per_step"]))
j = bench("c2")
g = j["gemm"]["alternatives"]
print("C2 forward in the stricter arithmetics: bf16x3 %.3f ms, fp32 MFMA %.3f ms.  CPU restatement of the TF graph on the box's %d host "
      "threads: %.3f mp-steps/s." % (g["bf16x3"]["ms_per_step"], g["f32"]["ms_per_step"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["value"]))
