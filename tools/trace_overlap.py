#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace CSV: per kernel name, the summed duration, and how much of the
traced span had 1, 2, 3.. kernels in flight.  usage: trace_overlap.py <kernel_trace.csv> [skip_first_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
dur = collections.Counter()
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < lo:
        continue
    name = r["Kernel_Name"].split("(")[0][-40:]
    dur[name] += e - s
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t; depth += d
span = ev[-1][0] - ev[0][0]
print("span %.3f ms, sum of kernel durations %.3f ms" % (span / 1e6, sum(dur.values()) / 1e6))
for k in sorted(hist):
    print("  %d kernels in flight: %.3f ms (%.1f%%)" % (k, hist[k] / 1e6, 100.0 * hist[k] / span))
for k, v in dur.most_common():
    print("  %-42s %.3f ms" % (k, v / 1e6))
