#!/usr/bin/env python
"""Copies the summaries tools/profile_r06_all.sh left in gpurun_out/r06/ (merged back by gpurun) to profiles/r06_* and
builds profiles/r06_spmm_pmc_traffic.json from the PMC passes.  python tools/collect_r06.py"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
DST = os.path.join(ROOT, "profiles")
KEEP = ["*_bench.json", "*_forward_kernel_stats.txt", "*_forward_timeline.txt", "*_loop_kernel_stats.txt", "*_resident_kernel_stats.txt", "c2_train_kernel_stats.txt",
        "c5_train_kernel_stats.txt", "train_variants.txt", "fuzz_parity.txt",
        "loop_vs_steps.txt", "resident_vs_steps.txt", "resident_trace.txt", "soak.txt", "stager_breakdown.txt", "grad_anchor_report.txt", "rowsum_once_bound.txt"]
n = 0
for pat in KEEP:
    for path in sorted(glob.glob(os.path.join(SRC, pat))):
        if os.path.getsize(path) == 0:
            print("EMPTY", path)
            continue
        shutil.copy(path, os.path.join(DST, "r06_" + os.path.basename(path)))
        n += 1
cell = os.path.join(ROOT, "gpurun_out", "r06cell")
for name in ("c2_pmc_mfma.txt", "c2_pmc_lds.txt"):
    if os.path.exists(os.path.join(cell, name)):
        shutil.copy(os.path.join(cell, name), os.path.join(DST, "r06_c2_forward_" + name[3:]))
        n += 1
subprocess.check_call([sys.executable, os.path.join(DST, "make_traffic_json.py"), SRC])
print("copied", n, "files")
