"""oracle/mslam_cv_pnp_oracle.py — the numpy restatement of cv::solvePnPRansac as OpenCV 4.8.1 runs it for the reference's call
(cv_ransac_pnp.cpp:56-57) — checked on CPU: the pieces against closed forms and ground truth.  PARITY UNPINNED (no OpenCV in this
image): these are properties of the published algorithm, not comparisons with a real build."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mslam_cv_pnp_oracle as cvo  # noqa: E402

CAM = (525.0, 525.0, 319.5, 239.5)


def scene(seed, n=400, outliers=0.3, noise=0.3, far=25.0):
    """ground-truth pose, n correspondences; outliers are moved at least `far` pixels away, so that no point sits near the
    5-pixel threshold of any reasonable hypothesis"""
    rng = np.random.default_rng(seed)
    rvec = rng.normal(size=3) * 0.4
    R = cvo.rodrigues(rvec)
    t = rng.normal(size=3) * 0.3 + np.array([0.1, -0.2, 0.5])
    cam_pts = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 7, n)], 1)
    obj = ((cam_pts - t) @ R).astype(np.float32)
    img = cvo.project_points(obj, rvec, t, CAM)
    img += np.clip(rng.normal(size=img.shape) * noise, -2.5 * noise, 2.5 * noise)
    bad = rng.random(n) < outliers
    ang = rng.uniform(0, 2 * np.pi, n)
    img[bad] += (far + rng.uniform(0, 200, n))[bad, None] * np.stack([np.cos(ang), np.sin(ang)], 1)[bad]
    return obj, img.astype(np.float32), rvec, t, ~bad


def rot_err(ra, rb):
    Ra, Rb = cvo.rodrigues(ra), cvo.rodrigues(rb)
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_cv_rng_is_the_multiply_with_carry_generator():
    r = cvo.CvRNG(cvo.M64)
    s = cvo.M64
    for _ in range(1000):
        s = (s & 0xFFFFFFFF) * 4164903690 + (s >> 32)       # exact in Python integers: never exceeds 64 bits
        assert s < 1 << 64
        assert r.next() == (s & 0xFFFFFFFF) and r.state == s
    assert cvo.CvRNG(0).state == 0xFFFFFFFF                  # RNG(0) falls back to 0xffffffff
    r = cvo.CvRNG(cvo.M64)
    draws = [r.uniform(0, 37) for _ in range(2000)]
    assert min(draws) == 0 and max(draws) == 36 and r.uniform(5, 5) == 5
    # getSubset: distinct indices, reproducible, and the generator is shared by consecutive subsets (one stream per call)
    a, b = cvo.CvRNG(cvo.M64), cvo.CvRNG(cvo.M64)
    s1, s2 = cvo.get_subset(a, 9, 5), cvo.get_subset(a, 9, 5)
    assert len(set(s1)) == 5 and len(set(s2)) == 5 and s1 != s2 and cvo.get_subset(b, 9, 5) == s1


def test_update_num_iters_is_ransacupdatenumiters():
    assert cvo.update_num_iters(0.99, 0.0, 5, 100) == 0 and cvo.update_num_iters(0.99, 0.6, 5, 100) == 100
    assert cvo.update_num_iters(0.99, 0.1, 5, 100) == 5 and cvo.update_num_iters(0.99, 0.3, 5, 100) == 25
    assert cvo.update_num_iters(0.99, 0.5, 5, 1000) == int(np.rint(np.log(0.01) / np.log(1 - 0.5 ** 5)))


def test_rodrigues_round_trip_and_jacobian():
    rng = np.random.default_rng(3)
    for _ in range(50):
        r = rng.normal(size=3)
        r *= rng.uniform(0.01, 3.0) / np.linalg.norm(r)      # rotation angles below pi: the vector is unique
        assert np.allclose(cvo.rodrigues_inv(cvo.rodrigues(r)), r, atol=1e-10)
    assert np.allclose(cvo.rodrigues_inv(np.eye(3)), 0)
    P = rng.normal(size=(7, 3)) + [0, 0, 5]
    r, t = np.array([0.2, -0.3, 0.1]), np.array([0.1, 0.2, 0.3])
    uv, J = cvo.project_points(P, r, t, CAM, jac=True)
    p0 = np.concatenate([r, t])
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        num = (cvo.project_points(P, (p0 + d)[:3], (p0 + d)[3:], CAM) - cvo.project_points(P, (p0 - d)[:3], (p0 - d)[3:], CAM)) / 2e-6
        assert np.allclose(J[:, k], num.reshape(-1), rtol=1e-5, atol=1e-4)


def test_epnp_is_exact_on_noise_free_minimal_samples():
    rng = np.random.default_rng(4)
    worst = 0.0
    for i in range(40):
        obj, img, rvec, t, _ = scene(100 + i, n=5 if i % 2 else 12, outliers=0.0, noise=0.0)
        us = cvo.project_points(obj, rvec, t, CAM)                       # exact pixels (no float rounding)
        R, tt = cvo.epnp(obj, us, CAM)
        worst = max(worst, rot_err(cvo.rodrigues_inv(R), rvec), float(np.linalg.norm(tt - t)))
        assert rot_err(cvo.rodrigues_inv(R), rvec) < 1e-3 and np.linalg.norm(tt - t) < 1e-4, (i, worst)


def test_iterative_refinement_converges_from_a_rough_guess():
    obj, img, rvec, t, _ = scene(9, n=60, outliers=0.0, noise=0.0)
    r, tt = cvo.refine_iterative(obj, img, CAM, rvec + [0.05, -0.04, 0.03], t + [0.1, -0.1, 0.2])
    assert rot_err(r, rvec) < 1e-3 and np.linalg.norm(tt - t) < 1e-4


def test_solve_pnp_ransac_recovers_pose_and_consensus_set():
    for seed, outl in ((1, 0.0), (2, 0.3), (3, 0.5), (4, 0.6)):
        obj, img, rvec, t, good = scene(seed, outliers=outl)
        res = cvo.solve_pnp_ransac(obj, img, CAM)
        assert res["ok"] and np.array_equal(res["mask"], good), (seed, res["mask"].sum(), good.sum())
        assert rot_err(res["rvec"], rvec) < 0.05 and np.linalg.norm(res["tvec"] - t) < 0.01
        assert (res["looked_at"] <= 8) == (outl == 0.0) and (res["looked_at"] == 100) == (outl >= 0.5)   # 145 samples for 0.99 at 50 %
        # the library's quirk: the final solve starts from the last hypothesis evaluated, wherever that is — it still lands in
        # the optimum of the consensus set
        last = [h for h in res["hypotheses"] if h is not None][-1]
        r2, t2 = cvo.refine_iterative(obj[res["inliers"]], img[res["inliers"]], CAM, last[0], last[1])
        assert np.array_equal(r2, res["rvec"]) and np.array_equal(t2, res["tvec"])
    rng = np.random.default_rng(5)
    none = cvo.solve_pnp_ransac(rng.normal(size=(50, 3)).astype(np.float32) + [0, 0, 5],
                                rng.uniform(0, 480, (50, 2)).astype(np.float32), CAM, thr=0.05)
    assert not none["ok"]
