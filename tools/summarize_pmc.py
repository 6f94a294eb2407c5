#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel medians (KB per launch).
usage: summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, sys


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return d


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


f, w = load(sys.argv[1]), load(sys.argv[2])
out = {k: {"FETCH_SIZE_KB": med(f[k]), "WRITE_SIZE_KB": med(w.get(k, [0.0])), "launches": len(f[k])} for k in f}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%-45s FETCH %12.1f KB  WRITE %12.1f KB" % (k[-45:], v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"]))
