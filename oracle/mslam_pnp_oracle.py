"""numpy oracle of the RANSAC PnP step (reference call site cv_ransac_pnp.cpp:56-57) — the KERNEL's
hypothesis sequence (same sampler, P3P); the oracle that follows OpenCV's own algorithm instead (cv::RNG, EPnP, LM) is
oracle/mslam_cv_pnp_oracle.py, and tests/test_pnp.py compares the HIP solver with both.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: cv::solvePnPRansac's internals (EPnP on 5-point samples, cv::RNG, LM of
cvFindExtrinsicCameraParams2) live in OpenCV, which is not in the reference tree; what IS fixed by the call site is the
contract — pin-hole camera, <= 100 hypotheses, 5 px inlier threshold, refinement on the consensus set from the
extrinsic guess.  This module restates the library's specification of that contract independently of the HIP code:
same splitmix64 sampling, P3P through numpy's polynomial root finder (not Ferrari), the fourth sample point picks the
P3P branch, most inliers wins (first on ties), Gauss-Newton refinement with numpy's linear algebra."""
import numpy as np

M64 = (1 << 64) - 1


def _splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return state, z ^ (z >> 31)


def sample_indices(seed, h, n):
    x = (seed + 0x9E3779B97F4A7C15 * (h + 1)) & M64
    idx = []
    for _ in range(4):
        tries = 0
        while True:
            x, r = _splitmix(x)
            k = r % n
            if k not in idx:
                idx.append(k)
                break
            tries += 1
            if tries > 64:
                return None
    return idx


def rodrigues(r):
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def project(R, t, P, cam):
    X = P @ R.T + t
    ok = X[:, 2] > 1e-9
    z = np.where(ok, X[:, 2], 1.0)
    return np.stack([cam[0] * X[:, 0] / z + cam[2], cam[1] * X[:, 1] / z + cam[3]], 1), ok


def _frame(a, b, c):
    e1 = (b - a) / np.linalg.norm(b - a)
    n = np.cross(e1, c - a)
    e3 = n / np.linalg.norm(n)
    return np.stack([e1, np.cross(e3, e1), e3], 1)      # columns


def p3p(P, uv, cam):
    """all (R, t) with X_i = R P_i + t on the rays of uv_i, i = 0..2 (Grunert's quartic via numpy.roots)"""
    f = np.stack([(uv[:, 0] - cam[2]) / cam[0], (uv[:, 1] - cam[3]) / cam[1], np.ones(3)], 1)
    f /= np.linalg.norm(f, axis=1)[:, None]
    a2, b2, c2 = np.sum((P[1] - P[2]) ** 2), np.sum((P[0] - P[2]) ** 2), np.sum((P[0] - P[1]) ** 2)
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    k1, k2 = (a2 - c2) / b2, (a2 + c2) / b2
    A = [(k1 - 1) ** 2 - 4 * c2 / b2 * ca ** 2,
         4 * (k1 * (1 - k1) * cb - (1 - k2) * ca * cg + 2 * c2 / b2 * ca ** 2 * cb),
         2 * (k1 ** 2 - 1 + 2 * k1 ** 2 * cb ** 2 + 2 * (b2 - c2) / b2 * ca ** 2 - 4 * k2 * ca * cb * cg + 2 * (b2 - a2) / b2 * cg ** 2),
         4 * (-k1 * (1 + k1) * cb + 2 * a2 / b2 * cg ** 2 * cb - (1 - k2) * ca * cg),
         (1 + k1) ** 2 - 4 * a2 / b2 * cg ** 2]
    out = []
    for v in np.roots(A):
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = 2 * (cg - v * ca)
        if abs(den) < 1e-14:
            continue
        u = ((k1 - 1) * v * v - 2 * k1 * cb * v + 1 + k1) / den
        s1d = 1 + v * v - 2 * v * cb
        if u <= 0 or s1d <= 1e-14:
            continue
        s1 = np.sqrt(b2 / s1d)
        X = np.stack([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
        R = _frame(*X) @ _frame(*P).T
        out.append((R, X[0] - R @ P[0]))
    return out


def hypothesis(obj, img, cam, idx):
    P, uv = obj[idx].astype(np.float64), img[idx].astype(np.float64)
    best = None
    try:
        sols = p3p(P[:3], uv[:3], cam)
    except (np.linalg.LinAlgError, FloatingPointError, ZeroDivisionError):
        return None
    for R, t in sols:
        pr, ok = project(R, t, P[3:4], cam)
        if not ok[0] or not np.isfinite(R).all():
            continue
        e = float(np.sum((pr[0] - uv[3]) ** 2))
        if best is None or e < best[0]:
            best = (e, R, t)
    return None if best is None else best[1:]


def inliers_of(R, t, obj, img, cam, thr):
    """reprojection error <= thr, in the division-free form the kernel evaluates (k_pnp.hip: is_inlier):
    (fx X + (cx - u) Z)^2 + (fy Y + (cy - v) Z)^2 <= thr^2 Z^2 for Z > 1e-9"""
    X = obj.astype(np.float64) @ R.T + t
    uv = img.astype(np.float64)
    eu = cam[0] * X[:, 0] + (cam[2] - uv[:, 0]) * X[:, 2]
    ev = cam[1] * X[:, 1] + (cam[3] - uv[:, 1]) * X[:, 2]
    return (X[:, 2] > 1e-9) & (eu * eu + ev * ev <= (thr * thr) * (X[:, 2] * X[:, 2]))


def refine(R, t, obj, img, cam, mask, iters=50):
    """Gauss-Newton on the reprojection error of the consensus set: X = exp(w) R P + t + dt"""
    P, uv = obj[mask].astype(np.float64), img[mask].astype(np.float64)

    def cost(R, t):
        pr, ok = project(R, t, P, cam)
        return np.sum((pr - uv) ** 2) + 1e12 * np.count_nonzero(~ok)
    c, lam = cost(R, t), 1e-3
    for _ in range(iters):
        Y = P @ R.T
        X = Y + t
        iz = 1.0 / X[:, 2]
        r = np.concatenate([cam[0] * X[:, 0] * iz + cam[2] - uv[:, 0], cam[1] * X[:, 1] * iz + cam[3] - uv[:, 1]])
        ux, uz = cam[0] * iz, -cam[0] * X[:, 0] * iz * iz
        vy, vz = cam[1] * iz, -cam[1] * X[:, 1] * iz * iz
        z0 = np.zeros(len(P))
        Ju = np.stack([uz * Y[:, 1], ux * Y[:, 2] - uz * Y[:, 0], -ux * Y[:, 1], ux, z0, uz], 1)
        Jv = np.stack([-vy * Y[:, 2] + vz * Y[:, 1], -vz * Y[:, 0], vy * Y[:, 0], z0, vy, vz], 1)
        J = np.concatenate([Ju, Jv])
        H, g = J.T @ J, J.T @ r
        d = np.linalg.solve(H + lam * np.diag(np.diag(H)) + 1e-12 * np.eye(6), -g)
        Rn, tn = rodrigues(d[:3]) @ R, t + d[3:]
        cn = cost(Rn, tn)
        if cn < c:
            done = np.linalg.norm(d) < 1e-12 or c - cn <= 1e-14 * c
            R, t, c, lam = Rn, tn, cn, max(lam * 0.1, 1e-12)
            if done:
                break
        else:
            lam *= 10
            if lam > 1e12 or np.linalg.norm(d) < 1e-9:   # rejected step below rounding level: converged
                break
    return R, t, c


# modelPoints of RANSACUpdateNumIters as cv::solvePnPRansac sets it for the reference's call (default flags, more than 4
# points: 5, calib3d/src/solvepnp.cpp) — the library's own minimal sample is 3 + 1 points, its stopping rule is the call site's
MODEL_POINTS = 5


def update_num_iters(p, ep, model_points, max_iters):
    """OpenCV's RANSACUpdateNumIters (calib3d/src/ptsetreg.cpp): samples needed to have drawn an all-inlier one with
    probability p when a share ep of the points are outliers; p outside (0, 1): no early exit"""
    if not (0.0 < p < 1.0):
        return max_iters
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))                  # cvRound


def pnp_ransac(obj, img, cam, iterations=100, thr=5.0, seed=0, guess=None, confidence=0.99):
    """-> dict(R, t, mask, best, counts, hyps, looked_at) or None.  cam = (fx, fy, cx, cy); guess = (R0, t0) or None.
    The loop is RANSACPointSetRegistrator::run's: hypotheses in order, a new best one (more inliers than the best so far
    and at least 4) lowers the iteration count for `confidence` (cv_ransac_pnp.cpp:57 passes 0.99)."""
    obj, img = np.asarray(obj, np.float32), np.asarray(img, np.float32)
    n = len(obj)
    hyps, counts = [], []
    best, bc, niters, h = -1, -1, iterations, 0
    while h < niters:
        idx = sample_indices(seed, h, n)
        hy = hypothesis(obj, img, cam, idx) if idx is not None else None
        hyps.append(hy)
        counts.append(-1 if hy is None else int(inliers_of(hy[0], hy[1], obj, img, cam, thr).sum()))
        if counts[h] > max(bc, 3):
            bc, best = counts[h], h
            niters = update_num_iters(confidence, (n - bc) / n, MODEL_POINTS, niters)
        h += 1
    if best < 0:
        return None
    mask = inliers_of(hyps[best][0], hyps[best][1], obj, img, cam, thr)
    R0, t0 = guess if guess is not None else hyps[best]
    R, t, c = refine(np.array(R0, np.float64), np.array(t0, np.float64), obj, img, cam, mask)
    return dict(R=R, t=t, mask=mask, best=best, counts=counts, hyps=hyps, cost=c, looked_at=h)
