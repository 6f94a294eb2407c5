#!/usr/bin/env python3
"""Per-stage device times of one batched step (HIP events on the context's stream), for kernel work.
usage: python tools/stage_bench.py [--batch 100] [--reps 10] [--bow]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import synth  # noqa: E402
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--min-area", type=int, default=1000)
    ap.add_argument("--bow", action="store_true")
    a = ap.parse_args()
    pkg = graft.load_package()
    B = a.batch
    frames = synth.make_stream(min(B, 50), a.width, a.height, seed=1234)
    frames = np.concatenate([frames] * ((B + len(frames) - 1) // len(frames)))[:B]
    d = torch.from_numpy(frames).cuda()
    big = a.width > 700
    ctx = pkg.Context(width=a.width, height=a.height, max_batch=B, n_levels=a.levels, min_node_area=a.min_area,
                      max_keypoints=16384 if big else 4096, max_candidates=65536 if big else 16384)
    if a.bow:
        ctx.bow_load(synth.make_vocabulary(10, 6, seed=77))

    def step():
        ctx.detect_batch_dev(d.data_ptr(), B)
        ctx.match_batch_dev(0.7, True)
        if a.bow:
            ctx.bow_batch_dev(True)

    for _ in range(3):
        step()
    ctx.sync()
    ctx.set_profiling(True)
    acc = {}
    for _ in range(a.reps):
        step()
        for n, ms in ctx.stage_times():
            acc.setdefault(n, []).append(ms)
    ctx.set_profiling(False)
    tot = 0
    for n, v in acc.items():
        print("%-14s median %.4f ms   min %.4f" % (n, float(np.median(v)), min(v)))
        tot += float(np.median(v))
    cnt = pkg.read_device(ctx, ctx.batch_view().count, (B,), np.int32)
    print("total %.4f ms / %d frames; keypoints/frame %.1f -> %.1f M keypoints/s" % (tot, B, cnt.mean(),
                                                                                  cnt.sum() / tot / 1e3))


if __name__ == "__main__":
    main()
