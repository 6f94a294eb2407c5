#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: every interval with NO kernel in flight (steady state only), grouped by
the kernel that ended before it and the kernel that started after it; plus one step's time line (kernel, queue, start, end).
usage: trace_gaps.py <kernel_trace.csv> [skip_first_fraction] [--timeline N_ROWS] [--until FRACTION]
(--until: ignore what starts after that fraction of the trace — the tail of a bench run is result copies, not steps)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 0.85
n_tl = int(sys.argv[sys.argv.index("--timeline") + 1]) if "--timeline" in sys.argv else 0
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * skip
until = float(sys.argv[sys.argv.index("--until") + 1]) if "--until" in sys.argv else 1.0
hi = t0 + (t1 - t0) * until
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mslam::", "")[:34],
              r.get("Queue_Id", "?")) for r in rows if lo <= int(r["Start_Timestamp"]) <= hi), key=lambda k: k[0])
span = ks[-1][1] - ks[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
idle = 0
cur_end, cur_name = ks[0][1], ks[0][2]
for s, e, name, q in ks[1:]:
    if s > cur_end:
        g = s - cur_end
        idle += g
        key = (cur_name, name)
        gaps[key][0] += g
        gaps[key][1] += 1
    if e > cur_end:
        cur_end, cur_name = e, name
print("span %.3f ms, idle (no kernel in flight) %.3f ms = %.1f %%" % (span / 1e6, idle / 1e6, 100.0 * idle / span))
for (a, b), (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %7.1f us in %3d gaps (%5.1f us each)  after %-34s before %s" % (g / 1e3, n, g / 1e3 / n, a, b))
if n_tl:
    base = ks[0][0]
    for s, e, name, q in ks[:n_tl]:
        print("  q%-3s %9.1f .. %9.1f us (%7.1f)  %s" % (q, (s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, name))
