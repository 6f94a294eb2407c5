#!/usr/bin/env python3
"""debug aid: a B-frame batch against the oracle, reports which frames / keypoints / fields differ"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, synth, __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = 4096
src = synth.make_stream(6, 640, 480, seed=1234)
idx = np.arange(B) % 6
frames = torch.from_numpy(np.ascontiguousarray(src[idx])).cuda()
c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K)
refs = [orc.detect(f, orc.params()) for f in src]
for rep in range(2):
    c.detect_batch_dev(frames.data_ptr(), B); c.sync()
    v = c.batch_view()
    cnt = pkg.read_device(c, v.count, (B,), np.int32)
    desc = pkg.read_device(c, v.desc, (B, K, 32), np.uint8)
    ang = pkg.read_device(c, v.angle, (B, K), np.float32)
    octv = pkg.read_device(c, v.octave, (B, K), np.int32)
    bad_frames = 0
    for t in range(B):
        r = refs[idx[t]]; n = len(r["xy"])
        if cnt[t] != n:
            print("rep", rep, "frame", t, "count", cnt[t], "!=", n); bad_frames += 1; continue
        bd = np.nonzero((desc[t, :n] != r["desc"]).any(1))[0]
        ba = np.nonzero(ang[t, :n].view(np.uint32) != r["angle"].view(np.uint32))[0]
        if len(bd) or len(ba):
            bad_frames += 1
            if bad_frames <= 8:
                print("rep", rep, "frame", t, "bad desc", len(bd), bd[:8], "levels", octv[t, bd[:8]], "bad angle", len(ba), ba[:8],
                      "bits", [int(np.unpackbits(desc[t, i] ^ r["desc"][i]).sum()) for i in bd[:8]])
    print("rep", rep, "bad frames", bad_frames, "of", B)
c.close()
