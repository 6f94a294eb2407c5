#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (not part of the pytest suites: minutes of GPU + oracle time).

    python tools/fuzz_parity.py [--seconds 120] [--seed 1]

Draws frame sizes, pyramid shapes, thresholds and stop areas at random, for both detector modes, on synthetic
texture / noise / low-contrast frames, and compares every output array of mslam_hip_detect with the oracle bit for
bit.  Prints one line per mismatch with the parameters that reproduce it and exits non-zero if there was any."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth, __graft_entry__ as g
pkg = g.load_package()
import mslam_oracle as orc

KEYS = ("xy", "desc", "octave", "angle", "response")


def frame(rng, W, H, kind):
    if kind == 0:
        return synth.make_stream(1, W, H, seed=int(rng.integers(1 << 30)))[0]
    if kind == 1:
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    f = synth.make_stream(1, W, H, seed=int(rng.integers(1 << 30)))[0].astype(np.int32)
    return np.clip(128 + (f - 128) // 4 + rng.integers(-3, 4, f.shape), 0, 255).astype(np.uint8)  # low contrast


def fuzz_matcher(rng, seconds):
    """random descriptor sets (dense / tie-heavy: few distinct bits), both matcher kernels, ratio test on"""
    c = pkg.Context(width=0, height=0, max_keypoints=6000)
    t0, n_ok, n_bad = time.time(), 0, 0
    while time.time() - t0 < seconds:
        n_from, n_to = int(rng.integers(0, 5000)), int(rng.integers(0, 5000))
        mode = int(rng.integers(3))
        if mode == 0:
            f = rng.integers(0, 256, (n_from, 32), dtype=np.uint8)
            t = rng.integers(0, 256, (n_to, 32), dtype=np.uint8)
        else:  # tie-heavy: descriptors drawn from a small pool with a few flipped bits
            pool = rng.integers(0, 256, (max(2, int(rng.integers(2, 40))), 32), dtype=np.uint8)

            def draw(n):
                d = pool[rng.integers(0, len(pool), n)].copy()
                if mode == 2 and n:
                    d[np.arange(n), rng.integers(0, 32, n)] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
                return d
            f, t = draw(n_from), draw(n_to)
        ratio = float(rng.choice([0.5, 0.7, 0.9, 1.0]))
        rf, rt = orc.match(f, t, ratio)
        for kind in (pkg.MATCHER_AUTO, pkg.MATCHER_POPCOUNT):
            c.set_matcher(kind)
            gf, gt = c.match(f, t, ratio)
            if not (np.array_equal(gf, rf) and np.array_equal(gt, rt)):
                n_bad += 1
                print("MISMATCH matcher", dict(n_from=n_from, n_to=n_to, mode=mode, ratio=ratio, kind=kind), flush=True)
            else:
                n_ok += 1
    c.close()
    print("fuzz matcher: %d ok, %d mismatches in %.0f s" % (n_ok, n_bad, time.time() - t0))
    return n_bad


def fuzz_batch(rng, seconds):
    """the batched device path (what bench.py times): random batch sizes and frame orders at random (small) frame sizes,
    several batches chained per context; every frame's keypoints / descriptors / matches vs the oracle"""
    import torch
    t0, n_ok, n_bad = time.time(), 0, 0
    while time.time() - t0 < seconds:
        W, H = int(rng.integers(120, 400)), int(rng.integers(120, 300))
        cv = bool(rng.integers(2))
        B = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 64]))
        uniq = synth.make_stream(5, W, H, seed=int(rng.integers(1 << 30)))
        K = 4096
        kw = dict(detector=pkg.DETECTOR_CV_ORB, n_features=300) if cv else dict(min_node_area=300)
        c = pkg.Context(width=W, height=H, max_batch=B, max_keypoints=K, n_levels=4, **kw)
        refs = [orc.cvorb_detect(f, orc.cvorb_params(n_features=300, n_levels=4)) if cv else
                orc.detect(f, orc.params(n_levels=4, min_size=300)) for f in uniq]
        prev = None
        bad = []
        for rep in range(3):
            n = int(rng.integers(1, B + 1))
            idx = rng.integers(0, 5, n)
            frames = torch.from_numpy(np.ascontiguousarray(uniq[idx])).cuda()
            c.detect_batch_dev(frames.data_ptr(), n)
            c.match_batch_dev(0.7, True)
            c.sync()
            v = c.batch_view()
            cnt = pkg.read_device(c, v.count, (n,), np.int32)
            desc = pkg.read_device(c, v.desc, (n, K, 32), np.uint8)
            xy = pkg.read_device(c, v.xy, (n, K, 2), np.float32)
            ang = pkg.read_device(c, v.angle, (n, K), np.float32)
            mc = pkg.read_device(c, v.match_count, (n,), np.int32)
            mf = pkg.read_device(c, v.match_from, (n, K), np.int32)
            mt = pkg.read_device(c, v.match_to, (n, K), np.int32)
            for i in range(n):
                r = refs[idx[i]]
                m = len(r["xy"])
                if cnt[i] != m or not (np.array_equal(desc[i, :m], r["desc"]) and np.array_equal(xy[i, :m], r["xy"])
                                       and np.array_equal(ang[i, :m], r["angle"])):
                    bad.append(("detect", rep, i))
                p = prev if i == 0 else refs[idx[i - 1]]
                if p is not None:
                    rf, rt = orc.match(r["desc"], p["desc"])
                    if mc[i] != len(rf) or not (np.array_equal(mf[i, :mc[i]], rf) and np.array_equal(mt[i, :mc[i]], rt)):
                        bad.append(("match", rep, i))
                elif mc[i] != 0:
                    bad.append(("match0", rep, i))
            prev = refs[idx[n - 1]]
        c.close()
        if bad:
            n_bad += 1
            print("MISMATCH batch", dict(W=W, H=H, cv=cv, B=B), bad[:4], flush=True)
        else:
            n_ok += 1
    print("fuzz batch: %d ok, %d mismatches in %.0f s" % (n_ok, n_bad, time.time() - t0))
    return n_bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--matcher", action="store_true", help="fuzz the matcher instead of the detectors")
    ap.add_argument("--batch", action="store_true", help="fuzz the batched device path (detect_batch_dev + match_batch_dev)")
    ap.add_argument("--max-width", type=int, default=900)
    ap.add_argument("--max-height", type=int, default=700)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    if a.matcher:
        sys.exit(1 if fuzz_matcher(rng, a.seconds) else 0)
    if a.batch:
        sys.exit(1 if fuzz_batch(rng, a.seconds) else 0)
    t0, n_ok, n_bad, n_skip = time.time(), 0, 0, 0
    while time.time() - t0 < a.seconds:
        cv = bool(rng.integers(2))
        W, H = int(rng.integers(96, a.max_width)), int(rng.integers(96, a.max_height))
        scale = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.25, 1.5, 2.0])))
        levels = int(rng.integers(1, 9))
        while min(W, H) / scale ** (levels - 1) < 80:
            levels -= 1
        kind = int(rng.integers(3))
        thr = int(rng.integers(5, 40))
        f = frame(rng, W, H, kind)
        desc = dict(cv=cv, W=W, H=H, scale=scale, levels=levels, kind=kind, thr=thr)
        try:
            if cv:
                n = int(rng.choice([0, 50, 500, 1000, 3000]))
                edge = int(rng.integers(19, 40))
                order = int(rng.integers(2))   # 0: libstdc++'s retainBest order (the default), 1: raster
                desc.update(n=n, edge=edge, order=order)
                c = pkg.Context(width=W, height=H, detector=pkg.DETECTOR_CV_ORB, n_features=n, n_levels=levels,
                                scale_factor=scale, ini_fast_thr=thr, edge_threshold=edge, max_keypoints=65535,
                                max_candidates=262144)
                c.set_cv_keypoint_order(order)
                ref = orc.cvorb_detect(f, orc.cvorb_params(n_features=n, n_levels=levels, scale_factor=scale,
                                                           fast_threshold=thr, edge_threshold=edge, order=order))
            else:
                min_thr = int(rng.integers(2, thr + 1))
                area = int(rng.choice([40, 150, 400, 1000, 4000]))
                desc.update(min_thr=min_thr, area=area)
                c = pkg.Context(width=W, height=H, n_levels=levels, scale_factor=scale, ini_fast_thr=thr,
                                min_fast_thr=min_thr, min_node_area=area, max_keypoints=65535, max_candidates=262144)
                ref = orc.detect(f, orc.params(n_levels=levels, scale_factor=scale, ini_fast_thr=thr, min_fast_thr=min_thr,
                                               min_size=area))
        except (pkg.MslamHipError, RuntimeError) as e:
            n_skip += 1
            print("skip", desc, str(e)[:80])
            continue
        try:
            got = c.detect(f, max_out=65535)
            bad = [k for k in KEYS if len(got[k]) != len(ref[k]) or not np.array_equal(got[k], ref[k])]
            if len(ref["xy"]) > 65535:
                bad = []
        except pkg.MslamHipError as e:
            bad = [] if e.code == pkg.E_CAPACITY else ["error: %s" % e]
            if not bad:
                n_skip += 1
        c.close()
        if bad:
            n_bad += 1
            print("MISMATCH", desc, bad, len(ref["xy"]), flush=True)
        else:
            n_ok += 1
    print("fuzz: %d ok, %d mismatches, %d skipped in %.0f s" % (n_ok, n_bad, n_skip, time.time() - t0))
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
