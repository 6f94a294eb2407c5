#!/usr/bin/env python3
"""GPU time line of ONE synchronous detect call from a rocprofv3 --kernel-trace CSV of tools/latency.py: kernels in start
order with their duration and the gap in front of each, median over the calls of the trace.
usage: python tools/call_timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mslam::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # a call starts with the carry kernel (or the gray kernel when there is none) once the previous call has described its frame
    # (k_pack_results followed k_describe until round 5: k_describe now writes the mapped result block itself)
    if ("k_gray" in n or "k_carry_prev" in n) and cur and any("k_describe" in x[0] for x in cur):
        calls.append(cur)
        cur = []
    cur.append((n, s, e))
calls = [c for c in calls[5:] if any("k_describe" in x[0] for x in c)]
if not calls:
    raise SystemExit("no complete call in the trace")
sig = collections.Counter(tuple(x[0] for x in c) for c in calls).most_common(1)[0][0]
same = [c for c in calls if tuple(x[0] for x in c) == sig]
med = lambda v: sorted(v)[len(v) // 2]
print("%d calls with the common kernel sequence (%d kernels)" % (len(same), len(sig)))
tot = 0.0
for i, name in enumerate(sig):
    dur = med([(c[i][2] - c[i][1]) / 1e3 for c in same])
    gap = med([(c[i][1] - c[i - 1][2]) / 1e3 for c in same]) if i else 0.0
    tot += dur + gap
    print("%-44s gap %6.1f us   run %6.1f us" % (name[-44:], gap, dur))
print("first kernel start -> last kernel end: %.1f us (median %.1f)" % (tot, med([(c[-1][2] - c[0][1]) / 1e3 for c in same])))
