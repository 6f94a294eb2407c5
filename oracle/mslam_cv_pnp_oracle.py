"""numpy restatement of cv::solvePnPRansac AS OPENCV 4.8.1 RUNS IT for the reference's call
(cv_ransac_pnp.cpp:56-57: useExtrinsicGuess = true, 100 iterations, 5 px, confidence 0.99, no distortion, default flags).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: OpenCV is not in /root/reference (conanfile.py:28 pins 4.8.1) and not in this
image, so nothing here has been compared with a real OpenCV; it is written from OpenCV's published algorithm — NOT from
csrc/k_pnp.hip, with which it shares neither the sampler, nor the minimal solver, nor the refinement:

  modules/calib3d/src/solvepnp.cpp   solvePnPRansac: 5 model points + SOLVEPNP_EPNP as the RANSAC kernel (more than 4 points,
                                     default flags), PnPRansacCallback::computeError (projectPoints into CV_32F, squared error
                                     in float), final solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess) on the inliers of the best
                                     hypothesis; solvePnPGeneric's EPnP branch (undistortPoints into CV_32FC2, then epnp)
  modules/calib3d/src/ptsetreg.cpp   RANSACPointSetRegistrator::run / getSubset / findInliers, RANSACUpdateNumIters;
                                     RNG rng((uint64)-1)
  modules/core/include/.../core.hpp  cv::RNG: multiply-with-carry, state = (unsigned)state * 4164903690 + (state >> 32),
                                     uniform(a, b) = next() % (b - a) + a
  modules/calib3d/src/epnp.cpp       control points by PCA, barycentric coordinates, M (2n x 12), null space of M^T M, the
                                     three beta approximations, five Gauss-Newton steps each, Horn's absolute orientation,
                                     the approximation with the smallest mean reprojection error wins
  modules/calib3d/src/calibration.cpp cvFindExtrinsicCameraParams2 with a guess: CvLevMarq(6 parameters, <= 20 iterations,
                                     FLT_EPSILON on the relative parameter step, diagonal scaled by 1 + lambda)

One quirk of the library is kept on purpose: the callback's rvec / tvec share their buffers with the caller's guess, every
hypothesis is written into them, so the final ITERATIVE solve does not start from the caller's guess but from the LAST
hypothesis the loop evaluated (not the best one either).

What differs from a real build at the level of the last bits: numpy's LAPACK SVD / eigen solvers instead of OpenCV's Jacobi
SVD (null-space vectors may come with the other sign: EPnP's result does not depend on it), libm differences.  A point whose
squared error lies within rounding of 25 may flip; the tests keep their scenes away from that."""
import numpy as np

CV_RNG_COEFF = 4164903690
M64 = (1 << 64) - 1


class CvRNG:
    """cv::RNG (multiply-with-carry)"""

    def __init__(self, state=M64):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * CV_RNG_COEFF + (self.state >> 32)) & M64
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def get_subset(rng, count, model_points):
    """RANSACPointSetRegistrator::getSubset: draws until the index is new (checkSubset is the default: always true)"""
    idx = []
    for _ in range(model_points):
        k = rng.uniform(0, count)
        while k in idx:
            k = rng.uniform(0, count)
        idx.append(k)
    return idx


def update_num_iters(p, ep, model_points, max_iters):
    """RANSACUpdateNumIters"""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    tiny = np.finfo(np.float64).tiny
    num = max(1.0 - p, tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))  # cvRound


def rodrigues(r):
    r = np.asarray(r, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def rodrigues_inv(R):
    """cv::Rodrigues, matrix -> vector"""
    R = np.asarray(R, np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(np.sum(r * r) * 0.25)
    c = min(max((np.trace(R) - 1) * 0.5, -1.0), 1.0)
    th = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        x = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        y = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        z = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(x) < abs(y) and abs(x) < abs(z) and (R[1, 2] > 0) != (y * z > 0):
            z = -z
        v = np.array([x, y, z])
        return v * (th / np.linalg.norm(v))
    return r * (0.5 / s) * th


def _drodrigues(r):
    """dR/dr_k, k = 0..2 (each 3 x 3), of R = exp([r]x)"""
    r = np.asarray(r, np.float64)
    th2 = float(r @ r)
    R = rodrigues(r)
    out = []
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1.0
        if th2 < 1e-24:
            v = e
            out.append(np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]]))
            continue
        w = np.cross(r, (np.eye(3) - R) @ e)
        A = r[k] * np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]]) + \
            np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        out.append(A @ R / th2)
    return out


def project_points(P, rvec, tvec, cam, jac=False):
    """cv::projectPoints without distortion, in double: (u, v) and optionally d(u, v)/d(rvec, tvec) as [2n, 6] with rows
    (u0, v0, u1, v1, ...)"""
    P = np.asarray(P, np.float64)
    R = rodrigues(rvec)
    X = P @ R.T + np.asarray(tvec, np.float64).reshape(3)
    z = np.where(X[:, 2] != 0, 1.0 / np.where(X[:, 2] != 0, X[:, 2], 1.0), 1.0)
    x, y = X[:, 0] * z, X[:, 1] * z
    uv = np.stack([x * cam[0] + cam[2], y * cam[1] + cam[3]], 1)
    if not jac:
        return uv
    n = len(P)
    J = np.zeros((2 * n, 6))
    dR = _drodrigues(rvec)
    for k in range(3):
        dX = P @ dR[k].T
        J[0::2, k] = cam[0] * (dX[:, 0] * z - X[:, 0] * z * z * dX[:, 2])
        J[1::2, k] = cam[1] * (dX[:, 1] * z - X[:, 1] * z * z * dX[:, 2])
    J[0::2, 3] = cam[0] * z
    J[0::2, 5] = -cam[0] * X[:, 0] * z * z
    J[1::2, 4] = cam[1] * z
    J[1::2, 5] = -cam[1] * X[:, 1] * z * z
    return uv, J


# ---- epnp.cpp ---------------------------------------------------------------------------------------------------------
_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


def _null_vectors(M):
    """rows of Ut (cvSVD of M^T M, CV_SVD_U_T): eigenvectors by descending eigenvalue; v[k] = row 11 - k"""
    w, V = np.linalg.eigh(M.T @ M)          # ascending
    return [V[:, k] for k in range(4)]      # v[0] = smallest eigenvalue = Ut row 11


def _L_6x10(v):
    L = np.zeros((6, 10))
    dv = np.zeros((4, 6, 3))
    for i in range(4):
        for j, (a, b) in enumerate(_PAIRS):
            dv[i, j] = v[i][3 * a:3 * a + 3] - v[i][3 * b:3 * b + 3]
    for i in range(6):
        d = dv[:, i]
        L[i] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2],
                2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3], d[3] @ d[3]]
    return L


def _svd_solve(A, b):
    return np.linalg.lstsq(A, b, rcond=None)[0]


def _betas_approx_1(L, rho):
    b4 = _svd_solve(L[:, [0, 1, 3, 6]], rho)
    if b4[0] < 0:
        b0 = np.sqrt(-b4[0])
        return np.array([b0, -b4[1] / b0, -b4[2] / b0, -b4[3] / b0])
    b0 = np.sqrt(b4[0])
    return np.array([b0, b4[1] / b0, b4[2] / b0, b4[3] / b0])


def _betas_approx_2(L, rho):
    b3 = _svd_solve(L[:, [0, 1, 2]], rho)
    if b3[0] < 0:
        b0 = np.sqrt(-b3[0])
        b1 = np.sqrt(-b3[2]) if b3[2] < 0 else 0.0
    else:
        b0 = np.sqrt(b3[0])
        b1 = np.sqrt(b3[2]) if b3[2] > 0 else 0.0
    if b3[1] < 0:
        b0 = -b0
    return np.array([b0, b1, 0.0, 0.0])


def _betas_approx_3(L, rho):
    b5 = _svd_solve(L[:, [0, 1, 2, 3, 4]], rho)
    if b5[0] < 0:
        b0 = np.sqrt(-b5[0])
        b1 = np.sqrt(-b5[2]) if b5[2] < 0 else 0.0
    else:
        b0 = np.sqrt(b5[0])
        b1 = np.sqrt(b5[2]) if b5[2] > 0 else 0.0
    if b5[1] < 0:
        b0 = -b0
    return np.array([b0, b1, b5[3] / b0, 0.0])


def _gauss_newton(L, rho, b):
    b = b.copy()
    for _ in range(5):
        A = np.zeros((6, 4))
        r = np.zeros(6)
        for i in range(6):
            l = L[i]
            A[i] = [2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3],
                    l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3],
                    l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3],
                    l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3]]
            r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] +
                             l[4] * b[1] * b[2] + l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] +
                             l[8] * b[2] * b[3] + l[9] * b[3] * b[3])
        b = b + _svd_solve(A, r)          # epnp::qr_solve: least squares by Householder QR
    return b


def epnp(P, us, cam):
    """epnp::compute_pose for n >= 4 points: P [n, 3] world, us [n, 2] pixels, cam = (fu, fv, uc, vc) -> (R, t)"""
    P = np.asarray(P, np.float64)
    us = np.asarray(us, np.float64)
    n = len(P)
    fu, fv, uc, vc = cam
    # choose_control_points
    cws = np.zeros((4, 3))
    cws[0] = P.sum(0) / n
    PW0 = P - cws[0]
    w, V = np.linalg.eigh(PW0.T @ PW0)
    order = np.argsort(-w)
    for i in range(1, 4):
        k = np.sqrt(max(w[order[i - 1]], 0.0) / n)
        cws[i] = cws[0] + k * V[:, order[i - 1]]
    # compute_barycentric_coordinates
    CC = (cws[1:] - cws[0]).T
    CCi = np.linalg.pinv(CC)
    al = np.zeros((n, 4))
    al[:, 1:] = (P - cws[0]) @ CCi.T
    al[:, 0] = 1.0 - al[:, 1] - al[:, 2] - al[:, 3]
    # fill_M
    M = np.zeros((2 * n, 12))
    for i in range(4):
        M[0::2, 3 * i] = al[:, i] * fu
        M[0::2, 3 * i + 2] = al[:, i] * (uc - us[:, 0])
        M[1::2, 3 * i + 1] = al[:, i] * fv
        M[1::2, 3 * i + 2] = al[:, i] * (vc - us[:, 1])
    v = _null_vectors(M)
    L = _L_6x10(v)
    rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for a, b in _PAIRS])

    def R_and_t(betas):
        ccs = sum(betas[k] * v[k] for k in range(4)).reshape(4, 3)
        pcs = al @ ccs
        if pcs[0, 2] < 0.0:               # solve_for_sign
            ccs, pcs = -ccs, -pcs
        pc0, pw0 = pcs.mean(0), P.mean(0)
        ABt = (pcs - pc0).T @ (P - pw0)
        U, _, Vt = np.linalg.svd(ABt)
        R = U @ Vt
        if np.linalg.det(R) < 0:
            R[2] = -R[2]
        t = pc0 - R @ pw0
        X = P @ R.T + t
        ue = uc + fu * X[:, 0] / X[:, 2]
        ve = vc + fv * X[:, 1] / X[:, 2]
        err = np.sum(np.sqrt((us[:, 0] - ue) ** 2 + (us[:, 1] - ve) ** 2)) / n
        return err, R, t

    sols = [None]
    for f in (_betas_approx_1, _betas_approx_2, _betas_approx_3):
        with np.errstate(all="ignore"):
            sols.append(R_and_t(_gauss_newton(L, rho, f(L, rho))))
    N = 1
    if sols[2][0] < sols[1][0]:
        N = 2
    if sols[3][0] < sols[N][0]:
        N = 3
    return sols[N][1], sols[N][2]


# ---- CvLevMarq as cvFindExtrinsicCameraParams2 drives it ---------------------------------------------------------------
def refine_iterative(P, uv, cam, rvec, tvec, max_iter=20, eps=float(np.finfo(np.float32).eps)):
    """the LM of solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess = true): all in double"""
    P, uv = np.asarray(P, np.float64), np.asarray(uv, np.float64)
    param = np.concatenate([np.asarray(rvec, np.float64).reshape(3), np.asarray(tvec, np.float64).reshape(3)])
    lam_lg10, iters = -3, 0

    def residual(p, jac):
        if jac:
            pr, J = project_points(P, p[:3], p[3:], cam, jac=True)
            return (pr - uv).reshape(-1), J
        return (project_points(P, p[:3], p[3:], cam) - uv).reshape(-1)

    def step(prev, JtJ, JtErr):
        A = JtJ.copy()
        A[np.diag_indices(6)] *= 1.0 + np.exp(lam_lg10 * np.log(10.0))
        return prev - np.linalg.lstsq(A, JtErr, rcond=None)[0]          # solve(..., DECOMP_SVD)

    err, J = residual(param, True)
    while True:
        JtJ, JtErr = J.T @ J, J.T @ err
        prev = param.copy()
        param = step(prev, JtJ, JtErr)
        if iters == 0:
            prev_norm = np.linalg.norm(err)
        while True:                                  # CHECK_ERR
            err_norm = np.linalg.norm(residual(param, False))
            if err_norm > prev_norm:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    param = step(prev, JtJ, JtErr)
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        iters += 1
        dn = np.linalg.norm(param - prev) / max(np.linalg.norm(prev), np.finfo(np.float64).tiny)   # CV_RELATIVE_L2
        if iters >= max_iter or dn < eps:
            break
        prev_norm = err_norm
        err, J = residual(param, True)
    return param[:3], param[3:]


# ---- solvePnPRansac ------------------------------------------------------------------------------------------------------
def solve_pnp_ransac(obj, img, cam, rvec0=None, tvec0=None, iterations=100, thr=5.0, confidence=0.99):
    """obj [n, 3] and img [n, 2] as the reference fills them (CV_32F, cv_ransac_pnp.cpp:20-41); cam = (fx, fy, cx, cy) as the
    CV_32F camera matrix holds them (:52-53).  -> dict(ok, rvec, tvec, mask, inliers, hypotheses, looked_at, best) —
    ok False = solvePnPRansac returned false."""
    obj = np.asarray(obj, np.float32)
    img = np.asarray(img, np.float32)
    cam = tuple(float(np.float32(c)) for c in cam)
    n = len(obj)
    model_points = 5 if n > 4 else 4
    assert n > 5, "the reference's call sites hand over many more points; n <= 5 takes other branches of the library"
    rng = CvRNG(M64)
    rvec = np.zeros(3) if rvec0 is None else np.array(rvec0, np.float64)
    tvec = np.zeros(3) if tvec0 is None else np.array(tvec0, np.float64)
    niters, max_good = max(iterations, 1), 0
    best_mask, best_model, hyps = None, None, []
    t2 = float(thr) * float(thr)
    it = 0
    while it < niters:
        idx = get_subset(rng, n, model_points)
        # solvePnPGeneric, EPnP branch: undistortPoints (no distortion: (u - cx) / fx in double) into CV_32FC2, back to
        # pixels in double inside epnp::init_points
        u = img[idx].astype(np.float64)
        norm = np.stack([(u[:, 0] - cam[2]) * (1.0 / cam[0]), (u[:, 1] - cam[3]) * (1.0 / cam[1])], 1).astype(np.float32)
        us = np.stack([norm[:, 0].astype(np.float64) * cam[0] + cam[2], norm[:, 1].astype(np.float64) * cam[1] + cam[3]], 1)
        try:
            R, t = epnp(obj[idx], us, cam)
            ok = bool(np.isfinite(R).all() and np.isfinite(t).all())
        except np.linalg.LinAlgError:
            ok = False
        it += 1
        if not ok:
            hyps.append(None)
            continue
        rvec, tvec = rodrigues_inv(R), t.copy()       # written into the buffers the final solve starts from
        # computeError: projectPoints into CV_32F, difference and squared norm in float
        pr = project_points(obj, rvec, tvec, cam).astype(np.float32)
        d = img - pr
        err = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
        mask = err.astype(np.float64) <= t2
        good = int(mask.sum())
        hyps.append((rvec.copy(), tvec.copy(), good, idx))
        if good > max(max_good, model_points - 1):
            best_mask, best_model, max_good = mask, (rvec.copy(), tvec.copy()), good
            niters = update_num_iters(confidence, (n - good) / n, model_points, niters)
    if max_good <= 0:
        return dict(ok=False, rvec=rvec, tvec=tvec, mask=np.zeros(n, bool), hypotheses=hyps, looked_at=it, best=None)
    sel = np.nonzero(best_mask)[0]
    r, t = refine_iterative(obj[sel].astype(np.float64), img[sel].astype(np.float64), cam, rvec, tvec)
    return dict(ok=True, rvec=r, tvec=t, mask=best_mask, inliers=sel, hypotheses=hyps, looked_at=it, best=best_model)
