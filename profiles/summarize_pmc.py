#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc rocpd database: per kernel & grid, mean of each counter (summed
over its instances per dispatch) next to the mean dispatch duration.
Usage: python profiles/summarize_pmc.py <results.db> [filter-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    c = sqlite3.connect(path)
    per = defaultdict(lambda: defaultdict(float))
    dur = {}
    names = {}
    for did, kname, gx, cname, val, d in c.execute(
            "select dispatch_id, kernel_name, grid_size_x, counter_name, value, duration from counters_collection"):
        if filt and filt not in kname:
            continue
        per[did][cname] += val
        dur[did] = d
        names[did] = (kname.split("(")[0][-60:], gx)
    agg = defaultdict(lambda: defaultdict(list))
    for did, cs in per.items():
        key = names[did]
        for k, v in cs.items():
            agg[key][k].append(v)
        agg[key]["duration_us"].append(dur[did] / 1e3)
    for key, cs in sorted(agg.items()):
        print("%s grid=%d calls=%d" % (key[0], key[1], len(cs["duration_us"])))
        for k, v in sorted(cs.items()):
            print("    %-32s %16.1f" % (k, sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
