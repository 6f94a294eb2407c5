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


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # csrc_sha(): bench.py only quotes a profile for the kernel sources it was collected on

f, w = load(sys.argv[1]), load(sys.argv[2])
out = {k: {"FETCH_SIZE_KB": med(f[k]), "WRITE_SIZE_KB": med(w.get(k, [0.0])), "launches": len(f[k])} for k in f}
out["_meta"] = {
    "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline --no-extras",
    "units": "KB per launch, median over launches", "csrc_sha": bench.csrc_sha(),
    "frames_per_launch": {"default": 500, "void mslam::k_match_knn2_fp4<4>": 1000, "mslam::k_ratio_compact": 1000,
                          "mslam::k_backproject": 1000},
    "note": "the detector runs as 2 chunks of 500 frames per 1000-frame step; k_resize_col is the median of its 7 "
            "per-level launches"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if k != "_meta":
        print("%-45s FETCH %12.1f KB  WRITE %12.1f KB" % (k[-45:], v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"]))
