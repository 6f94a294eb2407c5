#!/usr/bin/env python3
"""per-dispatch durations of the level kernels from a rocprofv3 kernel trace (serialized stage pass): which level costs what.
usage: python tools/level_trace.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_resize_blur" in n or "k_gray_blur" in n or "k_fast_cells" in n or "k_describe" in n or "k_quadtree" in n:
        key = (n.split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print("%-60s grid %s x %s  n=%d  median %.1f us  min %.1f" % (k[0][-60:], k[1], k[2], len(v), v[len(v) // 2], v[0]))
