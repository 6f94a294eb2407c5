#!/usr/bin/env python3
"""PCIe-inclusive throughput: frames start in pinned HOST memory; each 250-frame batch is copied H2D on a
copy stream (double-buffered) while the previous batch is extracted + matched.  Reported in DESIGN.md §5; the
headline bench value is the HBM-resident rate."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, synth
import __graft_entry__ as graft

pkg = graft.load_package()
B, steps = 250, 12
frames = synth.make_stream(50, 640, 480, seed=1234)
host = torch.from_numpy(np.concatenate([frames] * 5)[:B]).pin_memory()
dev = [torch.empty_like(host, device="cuda") for _ in range(2)]
ctx = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=4096)
copy_stream = torch.cuda.Stream()
ev = [torch.cuda.Event() for _ in range(2)]


def copy(i):
    with torch.cuda.stream(copy_stream):
        dev[i % 2].copy_(host, non_blocking=True)
        ev[i % 2].record(copy_stream)


for warm in range(2):
    copy(warm); ev[warm % 2].synchronize()
    ctx.detect_batch_dev(dev[warm % 2].data_ptr(), B); ctx.match_batch_dev(0.7, True)
ctx.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
copy(0)
for i in range(steps):
    ev[i % 2].synchronize()              # batch i is in HBM
    if i + 1 < steps:
        if i >= 1:
            ctx.sync()                   # batch i-1 (which used the buffer batch i+1 goes into) is done
        copy(i + 1)                      # overlaps the compute of batch i
    ctx.detect_batch_dev(dev[i % 2].data_ptr(), B)
    ctx.match_batch_dev(0.7, True)
ctx.sync(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
gb = steps * host.numel() / 1e9
print("PCIe-inclusive: %.1f k frames/s, H2D %.1f GB/s, %.3f ms per %d-frame batch" % (steps * B / dt / 1e3, gb / dt, dt / steps * 1e3, B))
