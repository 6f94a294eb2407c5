#!/usr/bin/env python3
"""Per-stage kernel times of the cfg2 step, every stage alone on the GPU (one stream, HIP events around each stage
launch), median and minimum over --reps repetitions.  The iteration tool behind the numbers in DESIGN.md §5:
    python tools/stage_times.py [--reps 12] [--batch 1000] [--width 640 --height 480 --levels 8]
Environment switches of the library (MSLAM_*) apply as usual, so `VAR=1 python tools/stage_times.py` is an A/B arm."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--unique", type=int, default=0, help="distinct frames (default: = batch)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--min-area", type=int, default=1000)
    ap.add_argument("--detector", default="distributed", choices=["distributed", "cvorb"])
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    import torch
    import synth
    import __graft_entry__ as graft
    pkg = graft.load_package()
    B = a.batch
    n_unique = max(B, a.unique)
    frames = synth.make_stream(n_unique, a.width, a.height, seed=1234)
    d_frames = torch.from_numpy(frames).cuda()
    depth = synth.make_depth(1, a.width, a.height, seed=1234)[0]
    d_depth = torch.from_numpy(np.ascontiguousarray(np.stack([depth] * B)).view(np.int16)).cuda()
    area = max(1, -(-a.width * a.height // (640 * 480)))
    k_scale = area * max(1, 1000 // max(a.min_area, 1))
    cv = a.detector == "cvorb"
    ts = torch.cuda.Stream()
    ctx = pkg.Context(width=a.width, height=a.height, max_batch=B, n_levels=a.levels, min_node_area=a.min_area,
                      max_keypoints=min(32736, 4096 * k_scale), max_candidates=16384 * area, stream=ts.cuda_stream,
                      detector=pkg.DETECTOR_CV_ORB if cv else pkg.DETECTOR_DISTRIBUTED)
    fb = a.width * a.height * 3

    def step(i):
        off = (i % (n_unique // B)) * B
        ctx.detect_batch_dev(d_frames.data_ptr() + off * fb, B)
        ctx.match_batch_dev(0.7, True)
        ctx.backproject_batch_dev(d_depth.data_ptr())

    for i in range(2):
        step(i)
    ctx.sync()
    ctx.set_profiling(1)
    rows = {}
    for i in range(a.reps):
        step(i)
        for name, ms in ctx.stage_times():
            rows.setdefault(name, []).append(ms)
    ctx.set_profiling(0)
    ctx.sync()
    med = {k: float(np.median(v)) for k, v in rows.items()}
    mn = {k: float(np.min(v)) for k, v in rows.items()}
    print("[%s] median ms per %d frames: %s | sum %.3f" % (a.label, B, " ".join("%s %.3f" % kv for kv in med.items()), sum(med.values())))
    print("[%s] min    ms per %d frames: %s | sum %.3f" % (a.label, B, " ".join("%s %.3f" % kv for kv in mn.items()), sum(mn.values())))
    print(json.dumps({"label": a.label, "median": med, "min": mn}))
    ctx.close()


if __name__ == "__main__":
    main()
