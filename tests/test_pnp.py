"""RANSAC PnP (row f-3; reference call site cv_ransac_pnp.cpp:56-57).  CPU: the numpy oracle recovers ground-truth poses.
GPU: the HIP solver against ground truth and against the oracle (same samples by construction: same splitmix64 rule)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mslam_pnp_oracle as po  # noqa: E402

CAM = (525.0, 525.0, 319.5, 239.5)      # TUM intrinsics, rgbd_file_provider.cpp:136-145


def scene(seed, n=400, outliers=0.3, noise=0.5):
    rng = np.random.default_rng(seed)
    R = po.rodrigues(rng.normal(size=3) * 0.4)
    t = rng.normal(size=3) * 0.3 + np.array([0.1, -0.2, 0.5])
    obj = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 7, n)], 1)
    obj = ((obj - t) @ R).astype(np.float32)            # world points whose camera coordinates are the box above
    img, ok = po.project(R, t, obj.astype(np.float64), CAM)
    assert ok.all()
    img += rng.normal(size=img.shape) * noise
    bad = rng.random(n) < outliers
    img[bad] = rng.uniform(0, [640, 480], (int(bad.sum()), 2))
    return obj, img.astype(np.float32), R, t, ~bad


def rot_err(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_oracle_recovers_ground_truth():
    for seed in range(4):
        obj, img, R, t, good = scene(seed)
        res = po.pnp_ransac(obj, img, CAM, seed=seed)
        assert res is not None and rot_err(res["R"], R) < 0.1 and np.linalg.norm(res["t"] - t) < 0.02
        assert (res["mask"] & ~good).sum() <= 5 and res["mask"].sum() > 0.9 * good.sum()
    # the extrinsic guess is the refinement's starting point (useExtrinsicGuess = true): a nearby guess converges to the
    # same optimum
    obj, img, R, t, good = scene(7)
    res = po.pnp_ransac(obj, img, CAM, seed=1, guess=(po.rodrigues([0.02, -0.01, 0.03]) @ R, t + 0.05))
    assert rot_err(res["R"], R) < 0.1 and np.linalg.norm(res["t"] - t) < 0.02
    # no consensus: pure garbage yields no model
    rng = np.random.default_rng(5)
    assert po.pnp_ransac(rng.normal(size=(50, 3)).astype(np.float32) + [0, 0, 5],
                         rng.uniform(0, 480, (50, 2)).astype(np.float32), CAM, thr=0.5) is None


def test_oracle_confidence_bound_ends_clean_scenes_early():
    """cv_ransac_pnp.cpp:57 passes confidence 0.99: RANSACUpdateNumIters on every new best hypothesis.  A clean scene needs a
    handful of hypotheses, a 60 %-outlier scene (w^5 = 1 %: 447 samples for 0.99) runs all 100; without a confidence the
    loop never ends early; and the winner is the first maximum of the hypotheses looked at.  The model-point count of the
    formula is the call site's (5: solvePnPRansac with default flags), pinned here: 25 hypotheses at 30 % outliers, not 17."""
    assert po.MODEL_POINTS == 5
    assert po.update_num_iters(0.99, 0.0, 5, 100) == 0 and po.update_num_iters(0.99, 0.6, 5, 100) == 100
    assert po.update_num_iters(0.99, 0.1, 5, 100) == 5 and po.update_num_iters(0.99, 0.3, 5, 100) == 25
    assert po.update_num_iters(0.99, 0.3, 4, 100) == 17
    assert po.update_num_iters(1.0, 0.1, 5, 100) == 100 and po.update_num_iters(0.0, 0.1, 5, 100) == 100
    obj, img, R, t, good = scene(21, outliers=0.0)
    clean = po.pnp_ransac(obj, img, CAM, seed=2)
    assert clean["looked_at"] <= 8 and rot_err(clean["R"], R) < 0.1
    obj, img, R, t, good = scene(22, outliers=0.6)
    hard = po.pnp_ransac(obj, img, CAM, seed=2)
    assert hard["looked_at"] == 100 and rot_err(hard["R"], R) < 0.2
    full = po.pnp_ransac(obj, img, CAM, seed=2, confidence=1.0)
    assert full["looked_at"] == 100 and full["best"] == hard["best"]
    c = hard["counts"]
    assert hard["best"] == int(np.argmax(c)) and c[hard["best"]] >= 4


def test_p3p_solutions_contain_the_true_pose():
    rng = np.random.default_rng(11)
    for _ in range(50):
        obj, img, R, t, _ = scene(int(rng.integers(1 << 30)), n=8, outliers=0.0, noise=0.0)
        sols = po.p3p(obj[:3].astype(np.float64), img[:3].astype(np.float64), CAM)
        assert any(rot_err(Rs, R) < 1e-3 and np.linalg.norm(ts - t) < 1e-4 for Rs, ts in sols)


@pytest.mark.gpu
def test_gpu_pnp_matches_ground_truth_and_oracle(pkg):
    c = pkg.Context(width=0, height=0)
    for seed in range(6):
        obj, img, R, t, good = scene(seed, n=300 + 200 * seed, outliers=0.1 * (seed % 4), noise=0.4)
        got = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], seed=seed)
        assert got is not None
        r, tv, mask = got
        Rg = po.rodrigues(r)
        assert rot_err(Rg, R) < 0.1 and np.linalg.norm(tv - t) < 0.02                 # ground truth
        ref = po.pnp_ransac(obj, img, CAM, seed=seed)
        assert np.array_equal(mask, ref["mask"]), (seed, mask.sum(), ref["mask"].sum())   # same best hypothesis
        assert rot_err(Rg, ref["R"]) < 1e-6 and np.linalg.norm(tv - ref["t"]) < 1e-7   # same optimum
    # with an extrinsic guess (what the reference passes, cv_ransac_pnp.cpp:56): same optimum from a nearby start
    obj, img, R, t, good = scene(9)
    R0 = po.rodrigues([0.02, -0.01, 0.03]) @ R
    th = np.arccos((np.trace(R0) - 1) / 2)
    r0 = th / (2 * np.sin(th)) * np.array([R0[2, 1] - R0[1, 2], R0[0, 2] - R0[2, 0], R0[1, 0] - R0[0, 1]])
    r, tv, mask = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], rvec=r0, tvec=t + 0.05, seed=3)
    ref = po.pnp_ransac(obj, img, CAM, seed=3, guess=(R0, t + 0.05))
    assert np.array_equal(mask, ref["mask"]) and rot_err(po.rodrigues(r), ref["R"]) < 1e-6
    assert np.linalg.norm(tv - ref["t"]) < 1e-7 and rot_err(po.rodrigues(r), R) < 0.1
    # the confidence bound (cv_ransac_pnp.cpp:57): a clean scene stops after a few hypotheses, a 60 %-outlier scene does
    # not; both agree with the oracle's sequential loop, and switching the bound off gives the all-hypotheses result
    for sd, outl in ((21, 0.0), (22, 0.6), (23, 0.3)):
        obj, img, R, t, good = scene(sd, outliers=outl)
        ref = po.pnp_ransac(obj, img, CAM, seed=2)
        r, tv, mask = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], seed=2)
        assert np.array_equal(mask, ref["mask"]) and rot_err(po.rodrigues(r), ref["R"]) < 1e-6, (sd, ref["looked_at"])
        assert (ref["looked_at"] <= 8) == (outl == 0.0) and (ref["looked_at"] == 100) == (outl == 0.6)
        c.pnp_set_confidence(1.0)
        full = po.pnp_ransac(obj, img, CAM, seed=2, confidence=1.0)
        r, tv, mask = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], seed=2)
        assert full["looked_at"] == 100 and np.array_equal(mask, full["mask"]) and rot_err(po.rodrigues(r), full["R"]) < 1e-6
        c.pnp_set_confidence(0.99)
    # no model
    rng = np.random.default_rng(5)
    assert c.pnp_ransac(rng.normal(size=(50, 3)).astype(np.float32) + [0, 0, 5],
                        rng.uniform(0, 480, (50, 2)).astype(np.float32), CAM[:2], CAM[2:], reprojection_error=0.5) is None
    with pytest.raises(pkg.MslamHipError):
        c.pnp_ransac(obj[:3], img[:3], CAM[:2], CAM[2:])
    c.close()


@pytest.mark.gpu
def test_gpu_pnp_against_the_opencv_algorithm_oracle(pkg):
    """the HIP solver against oracle/mslam_cv_pnp_oracle.py — cv::solvePnPRansac restated from OpenCV's published algorithm
    (cv::RNG 5-point subsets, EPnP, float reprojection errors, RANSACUpdateNumIters, LM from the last hypothesis), which shares
    neither sampler nor minimal solver nor refinement with the kernel (P3P on splitmix64 samples, damped Gauss-Newton).  What
    must agree all the same is what the call site consumes (cv_ransac_pnp.cpp:59-83): success, the consensus set and the pose
    refined on it.  Noise-free scenes with far outliers (every all-inlier hypothesis of either solver explains exactly the true
    inliers): masks equal, rvec / tvec within 1e-6.  Noisy scenes: masks may differ in a few borderline points, poses within
    0.02 degrees / 2 mm of each other."""
    import mslam_cv_pnp_oracle as cvo
    from test_oracle_cv_pnp import scene as cv_scene
    c = pkg.Context(width=0, height=0)
    for seed, outl in ((31, 0.0), (32, 0.2), (33, 0.45), (34, 0.6)):
        obj, img, rvec, t, good = cv_scene(seed, n=500, outliers=outl, noise=0.0)
        ref = cvo.solve_pnp_ransac(obj, img, CAM)
        got = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], seed=seed)
        assert ref["ok"] and got is not None
        r, tv, mask = got
        assert np.array_equal(ref["mask"], good) and np.array_equal(mask, good), (seed, mask.sum(), ref["mask"].sum(), good.sum())
        assert np.abs(r - ref["rvec"]).max() < 1e-6 and np.abs(tv - ref["tvec"]).max() < 1e-6, (seed, r - ref["rvec"], tv - ref["tvec"])
    for seed, outl in ((41, 0.1), (42, 0.4)):
        obj, img, rvec, t, good = cv_scene(seed, n=500, outliers=outl, noise=0.4)
        # the call site passes the previous pose as the guess: both entry points take it, both end in the consensus optimum
        ref = cvo.solve_pnp_ransac(obj, img, CAM, rvec0=rvec + 0.01, tvec0=t + 0.02)
        r, tv, mask = c.pnp_ransac(obj, img, CAM[:2], CAM[2:], rvec=rvec + 0.01, tvec=t + 0.02, seed=seed)
        assert (mask != ref["mask"]).sum() <= 0.02 * len(mask) and not (mask & ~good).any() and not (ref["mask"] & ~good).any()
        assert rot_err(po.rodrigues(r), po.rodrigues(ref["rvec"])) < 0.02 and np.linalg.norm(tv - ref["tvec"]) < 2e-3
    rng = np.random.default_rng(5)
    o, i2 = rng.normal(size=(50, 3)).astype(np.float32) + [0, 0, 5], rng.uniform(0, 480, (50, 2)).astype(np.float32)
    assert not cvo.solve_pnp_ransac(o, i2, CAM, thr=0.05)["ok"] and c.pnp_ransac(o, i2, CAM[:2], CAM[2:], reprojection_error=0.05) is None
    c.close()


@pytest.mark.gpu
def test_gpu_pnp_batch_on_device_results(pkg):
    """the batched device form: correspondences of every frame gathered on the device from the matches and the previous
    frame's back-projected points; one PnP per frame.  A fronto-parallel plane at 2 m seen by a camera that moves
    parallel to it: frame t is frame t-1 shifted by whole pixels, so the pose between consecutive frames is
    R = I, t = (dx Z / fx, dy Z / fy, 0).  Checked: the gathered correspondences against a host-side gather, every
    pose against the single-problem entry point on the same data and seed, the oracle on two frames, ground truth."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    B, K, Z = 6, 4096, 2.0
    base = synth.make_stream(1, 640 + 64, 480 + 48, seed=11)[0]
    shifts = [(3 * t, 2 * t) for t in range(B)]                      # (dx, dy): the window moves right/down = camera moves
    frames = np.stack([np.ascontiguousarray(base[dy:dy + 480, dx:dx + 640]) for dx, dy in shifts])
    depth = np.full((B, 480, 640), int(Z * 5000), np.uint16)
    c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K)
    c.detect_batch_dev(torch.from_numpy(frames).cuda().data_ptr(), B)
    c.match_batch_dev(0.7, False)
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    c.backproject_batch_dev(d_depth.data_ptr(), focal=CAM[:2], principal=CAM[2:])
    c.pnp_batch_dev(CAM[:2], CAM[2:], seed=40)
    c.sync()
    v, pv, nv = c.batch_view(), c.points_view(), c.pnp_view()
    xy = pkg.read_device(c, v.xy, (B, K, 2), np.float32)
    mc = pkg.read_device(c, v.match_count, (B,), np.int32)
    mf = pkg.read_device(c, v.match_from, (B, K), np.int32)
    mt = pkg.read_device(c, v.match_to, (B, K), np.int32)
    xyz = pkg.read_device(c, pv.xyz, (B, K, 3), np.float64)
    ok = pkg.read_device(c, pv.valid, (B, K), np.uint8)
    pose = pkg.read_device(c, nv.pose, (B, 16), np.float64)
    npts = pkg.read_device(c, nv.n_points, (B,), np.int32)
    obj = pkg.read_device(c, nv.object_points, (B, K, 3), np.float32)
    img = pkg.read_device(c, nv.image_points, (B, K, 2), np.float32)
    inl = pkg.read_device(c, nv.inliers, (B, K), np.uint8)
    assert npts[0] == 0 and pose[0, 14] == 0.0                       # no predecessor inside the batch
    single = pkg.Context(width=0, height=0)
    for t in range(1, B):
        keep = ok[t - 1, mt[t, :mc[t]]] != 0
        ref_obj = xyz[t - 1, mt[t, :mc[t]][keep]].astype(np.float32)
        ref_img = xy[t, mf[t, :mc[t]][keep]]
        n = int(npts[t])
        assert n == len(ref_obj) and n > 300
        assert np.array_equal(obj[t, :n], ref_obj) and np.array_equal(img[t, :n], ref_img)
        assert pose[t, 14] == 1.0
        R, tv = pose[t, :9].reshape(3, 3), pose[t, 9:12]
        r1, t1, m1 = single.pnp_ransac(ref_obj, ref_img, CAM[:2], CAM[2:], seed=40 + t)
        assert np.array_equal(m1, inl[t, :n]) and int(pose[t, 12]) == int(m1.sum())
        # (the single-problem call returns a Rodrigues vector: compare the matrices, not arccos of a trace next to 3)
        assert np.abs(po.rodrigues(r1) - R).max() < 1e-9 and np.array_equal(t1, tv)
        dx, dy = shifts[t][0] - shifts[t - 1][0], shifts[t][1] - shifts[t - 1][1]
        # a point at pixel x in frame t-1 is at x - dx in frame t: X_t = X_{t-1} - (dx Z / fx, dy Z / fy, 0)
        assert rot_err(R, np.eye(3)) < 0.2 and np.linalg.norm(tv - [-dx * Z / CAM[0], -dy * Z / CAM[1], 0]) < 0.01
        assert m1.sum() > 0.9 * n
        if t <= 2:
            ref = po.pnp_ransac(ref_obj, ref_img, CAM, seed=40 + t)
            # (rot_err goes through arccos near 1: ~1e-6 degrees is its own resolution there)
            assert np.array_equal(ref["mask"], m1) and np.abs(R - ref["R"]).max() < 1e-9 and np.linalg.norm(tv - ref["t"]) < 1e-7
    single.close()
    c.close()
