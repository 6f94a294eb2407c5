#!/usr/bin/env python3
"""Per-kernel medians of every counter in a rocprofv3 --pmc counter_collection.csv.  usage: summarize_sq.py <csv> [out.json]"""
import collections, csv, json, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in d.items()}
if len(sys.argv) > 2:
    full = dict(out)
    full["_meta"] = {"csrc_sha": bench.csrc_sha(), "units": "counter value per launch, median over launches",
                     "frames_per_launch": {"default": 500, "void mslam::k_match_knn2_fp4<4>": 1000,
                                           "mslam::k_ratio_compact": 1000, "mslam::k_backproject": 1000}}
    json.dump(full, open(sys.argv[2], "w"), indent=1)
for k, cs in out.items():
    print(k[-40:])
    for c, v in sorted(cs.items()):
        print("    %-28s %16.0f" % (c, v))
