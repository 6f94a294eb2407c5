#!/usr/bin/env python3
"""every detector stage alone on the GPU (serialised timing), 500-frame batches: quick A/B of kernel changes"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, synth, __graft_entry__ as g
pkg = g.load_package()
B = 500
d = torch.from_numpy(synth.make_stream(B, 640, 480, seed=1234)).cuda()
c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=4096)
c.set_profiling(1)
acc = {}
for rep in range(8):
    c.detect_batch_dev(d.data_ptr(), B)
    c.match_batch_dev(0.7, True)
    if rep >= 3:
        for k, v in c.stage_times():
            acc[k] = acc.get(k, 0.0) + v / 5
print({k: round(v, 4) for k, v in acc.items()}, "sum", round(sum(acc.values()), 4))
