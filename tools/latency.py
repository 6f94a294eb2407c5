#!/usr/bin/env python3
"""Single-frame latency through the drop-in boundary: what `IFeatureDetector::detect` (rgbd_feature_frontend.cpp:187) and
`IFeatureMatcher::match` (:237) of the plugin cost per call.  Calls `mslam_hip_detect` / `mslam_hip_match` of the C ABI
directly (ctypes, preallocated outputs): host BGR frame in, host keypoints / descriptors / matches out, synchronous.
    python tools/latency.py [--calls 300] [--pinned]      (--pinned: the frame sits in page-locked host memory)
Prints the median / p95 in microseconds and one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--detector", default="distributed", choices=["distributed", "cvorb"])
    ap.add_argument("--pinned", action="store_true")
    a = ap.parse_args()
    import torch
    import synth
    import __graft_entry__ as graft
    pkg = graft.load_package()
    frames = synth.make_stream(16, a.width, a.height, seed=1234)
    if a.pinned:
        keep = [torch.from_numpy(f.copy()).pin_memory() for f in frames]
        frames = [k.numpy() for k in keep]
    cv = a.detector == "cvorb"
    area = max(1, -(-a.width * a.height // (640 * 480)))    # capacities scale with the frame area, as in bench.py
    c = pkg.Context(width=a.width, height=a.height, max_batch=1, max_keypoints=min(32736, 4096 * area), max_candidates=16384 * area,
                    detector=pkg.DETECTOR_CV_ORB if cv else pkg.DETECTOR_DISTRIBUTED)
    L, K = c.L, c.params.max_keypoints
    xy = np.empty((K, 2), np.float32)
    de = np.empty((2, K, 32), np.uint8)
    oc, an, rs = np.empty(K, np.int32), np.empty(K, np.float32), np.empty(K, np.float32)
    fi, ti = np.empty(K, np.int32), np.empty(K, np.int32)
    n, m = C.c_int(0), C.c_int(0)
    pp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    t_det, t_mat, counts = [], [], [0, 0]
    for i in range(a.calls + 20):
        fr = frames[i % 16]
        t0 = time.perf_counter()
        rc = L.mslam_hip_detect(c._h, pp(fr), a.width, a.height, K, pp(xy), pp(de[i & 1]), pp(oc), pp(an), pp(rs), C.byref(n))
        t1 = time.perf_counter()
        counts[i & 1] = n.value
        if rc == 0 and i > 0:
            rc = L.mslam_hip_match(c._h, pp(de[i & 1]), counts[i & 1], pp(de[(i & 1) ^ 1]), counts[(i & 1) ^ 1], C.c_double(0.7),
                                   pp(fi), pp(ti), C.byref(m))
        t2 = time.perf_counter()
        if rc != 0:
            raise SystemExit("call failed: %d %s" % (rc, L.mslam_hip_last_error(c._h)))
        if i >= 20:
            t_det.append(t1 - t0)
            t_mat.append(t2 - t1)
    c.close()
    out = {"detect_us": round(float(np.median(t_det)) * 1e6, 1), "detect_p95_us": round(float(np.percentile(t_det, 95)) * 1e6, 1),
           "match_us": round(float(np.median(t_mat)) * 1e6, 1), "match_p95_us": round(float(np.percentile(t_mat, 95)) * 1e6, 1),
           "keypoints": counts[0], "matches": m.value, "calls": a.calls, "frame": "%dx%d" % (a.width, a.height),
           "detector": a.detector, "pinned_input": bool(a.pinned)}
    print("detect %.1f us (p95 %.1f), match %.1f us (p95 %.1f), %d keypoints, %d matches" % (
        out["detect_us"], out["detect_p95_us"], out["match_us"], out["match_p95_us"], counts[0], m.value))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
