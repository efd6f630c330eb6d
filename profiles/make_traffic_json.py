#!/usr/bin/env python
"""Builds profiles/<round>_spmm_pmc_traffic.json (round = name of the source directory) from the per-workload PMC summaries
tools/profile_r02.sh / tools/profile_r03_all.sh leave in gpurun_out/<round>/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; summarize_pmc.py text).  FETCH_SIZE is doubled
(gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md); the counters sit on the L2<->fabric side.
Usage: python profiles/make_traffic_json.py [gpurun_out/r02]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02")


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            m = re.match(r"(.*) grid=(\d+) calls=(\d+)", line.strip())
            cur = (m.group(1), int(m.group(2)), int(m.group(3))) if m else None
            if cur:
                out[cur] = {}
        elif cur:
            k, v = line.split()
            out[cur][k] = float(v)
    return out


def grab(tag, sub, F, Wr, dst):
    for (name, grid, calls), v in F.items():
        if sub in name:
            wv = [x for (n2, g2, _), x in Wr.items() if n2 == name and g2 == grid]
            if not wv:
                continue
            f, wb = v["FETCH_SIZE"] * 1024 * 2, wv[0]["WRITE_SIZE"] * 1024
            dst.setdefault(tag, {})[str(grid)] = {
                "dispatches": calls, "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": wv[0]["WRITE_SIZE"],
                "fetch_bytes_corrected": int(f), "write_bytes": int(wb), "traffic_bytes": int(f + wb),
                "avg_us_under_profiler": v["duration_us"]}


res = {}
for w in ("c2", "c4", "c5"):
    p = lambda name: os.path.join(SRC, "%s_%s.txt" % (w, name))
    if not os.path.exists(p("pmc_FETCH_SIZE")):
        continue
    F, Wr = parse(p("pmc_FETCH_SIZE")), parse(p("pmc_WRITE_SIZE"))
    FF, WF = parse(p("forward_pmc_FETCH_SIZE")), parse(p("forward_pmc_WRITE_SIZE"))
    ent = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py "
                  "--workload %s --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --no-graph (kernels) and -- python "
                  "tools/forward_only.py %s 3 (in_forward); per dispatch means; FETCH_SIZE doubled (gfx950 reports half of "
                  "a wide coalesced read); L2<->fabric side, Infinity-Cache hits included" % (w, w),
           "kernels": {}, "in_forward": {}}
    for tag, sub in (("spmm_pair", "spmm_pair_kernel"), ("gather2_sum", "gather2_sum"), ("csr_rowsum", "csr_rowsum")):
        grab(tag, sub, F, Wr, ent["kernels"])
    for tag, sub in (("csr_rowsum", "csr_rowsum"), ("cell_launch", "lnlstm_mlp_fwd_h2_kernel"),
                     ("cell_launch_bf16", "lnlstm_fwd_bf16"), ("mlp_bf16", "mlp_fwd_bf16")):
        grab(tag, sub, FF, WF, ent["in_forward"])
    k = ent["kernels"]

    def one(tag):
        d = k.get(tag, {})
        return max(d.values(), key=lambda x: x["dispatches"])["traffic_bytes"] if d else None
    ent["pair_traffic_bytes"] = one("spmm_pair") or ((one("gather2_sum") or 0) + (one("csr_rowsum") or 0))
    res[w] = ent
    print(w, "pair", ent["pair_traffic_bytes"], {t: {g: (x["traffic_bytes"], x["avg_us_under_profiler"]) for g, x in d.items()}
                                                  for t, d in ent["in_forward"].items()})
tag = os.path.basename(os.path.normpath(os.path.abspath(SRC)))
if not re.fullmatch(r"r\d\d\w*", tag):   # (round 3 ran with SRC='.' and committed profiles/._spmm_pmc_traffic.json)
    raise SystemExit("make_traffic_json: source directory %r must be named after its round (r04, r04b, ...)" % SRC)
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_spmm_pmc_traffic.json" % tag), "w"), indent=1)
