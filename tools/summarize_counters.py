#!/usr/bin/env python3
"""Per-STEP totals of rocprofv3 --pmc counters, per kernel and per pipeline stage.
usage: summarize_counters.py <n_steps> <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
A stage of the step is one or more kernels and several launches (the detector runs as two 500-frame chunks per
1000-frame step; the pyramid is one launch per level), so the unit that can be compared with `ms_per_step` and with
the algorithmic bytes of a step is the SUM over all launches of the profiled run divided by the number of steps the
profiled command ran (bench.py --steps S --warmup W --no-extras runs max(W, 1) + S steps).  bench.py reads the JSON
(`profiles/r05_pmc_per_step.json`) for `roofline.traffic` = (2 x FETCH_SIZE + WRITE_SIZE) KB (the gfx950 factor 2 of
MI355X_MICROARCH.md §HBM) and for `roofline.step_valu_issue`."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # csrc_sha() and the kernel -> stage map

n_steps = int(sys.argv[1])
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        per_kernel[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
kernels = {k: {c: v / n_steps for c, v in cs.items()} for k, cs in per_kernel.items()}
for k in kernels:
    kernels[k]["launches_per_step"] = max(launches[k].values()) / float(n_steps)
stages = collections.defaultdict(lambda: collections.defaultdict(float))
for k, cs in kernels.items():
    st = bench.stage_of_kernel(k)
    if st is None:
        continue
    for c, v in cs.items():
        stages[st][c] += v
out = {"_meta": {"csrc_sha": bench.csrc_sha(), "frames_per_step": 1000, "steps_in_profiled_run": n_steps,
                 "units": "counter totals per 1000-frame step (sum over all launches of the run / steps); FETCH_SIZE and "
                          "WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE x 2 = bytes on gfx950)",
                 "command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "
                            "(one pass per counter group; FETCH_SIZE and WRITE_SIZE in separate passes)"},
       "stages": {s: dict(v) for s, v in stages.items()}, "kernels": kernels}
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
tot = collections.defaultdict(float)
for s, cs in sorted(stages.items()):
    print("%-14s %s" % (s, "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(cs.items()))))
    for c, v in cs.items():
        tot[c] += v
print("%-14s %s" % ("TOTAL", "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(tot.items()))))
if "FETCH_SIZE" in tot:
    print("memory-side MB per frame: %.2f" % ((2 * tot["FETCH_SIZE"] + tot.get("WRITE_SIZE", 0.0)) * 1024 / 1000 / 1e6))
if "SQ_INSTS_VALU" in tot:
    print("vector-ALU issue time per step at 4 cycles per instruction: %.3f ms" % (tot["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9) * 1e3))
