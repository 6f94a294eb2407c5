#!/usr/bin/env python3
"""Single-frame latency of the synchronous C-ABI calls (host buffers in, host buffers out): what a caller that
feeds one frame at a time, like the reference's frontend, sees.  usage: python tools/latency.py [--reps 200]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    pkg = graft.load_package()
    frames = synth.make_stream(8, 640, 480, seed=1234)
    ctx = pkg.Context(width=640, height=480, max_batch=1)
    dets = [ctx.detect(f) for f in frames]
    t = []
    for i in range(a.reps):
        t0 = time.perf_counter(); ctx.detect(frames[i % 8]); t.append(time.perf_counter() - t0)
    t.sort()
    print("detect 640x480 (H2D 0.92 MB + 8-level ORB + D2H of %d keypoints): median %.0f us, p10 %.0f us" % (
        len(dets[0]["xy"]), t[len(t) // 2] * 1e6, t[len(t) // 10] * 1e6))
    m = []
    for i in range(a.reps):
        t0 = time.perf_counter(); ctx.match(dets[(i + 1) % 8]["desc"], dets[i % 8]["desc"]); m.append(time.perf_counter() - t0)
    m.sort()
    print("match %d x %d (H2D + knn-2 + ratio + D2H): median %.0f us, p10 %.0f us" % (
        len(dets[1]["desc"]), len(dets[0]["desc"]), m[len(m) // 2] * 1e6, m[len(m) // 10] * 1e6))


if __name__ == "__main__":
    main()
