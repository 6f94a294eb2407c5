"""ctypes binding of the CPU oracle (oracle/libmslam_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (modular-slam_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmslam_oracle.so")


class Params(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("scale_factor", C.c_float), ("ini_fast_thr", C.c_int),
                ("min_fast_thr", C.c_int), ("min_size", C.c_uint)]


class Cand(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("response", C.c_float)]


CAND_DT = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4")])


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.mso_fast_atan2.restype = C.c_float
        _lib.mso_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.mso_util_cos.restype = C.c_float
        _lib.mso_util_cos.argtypes = [C.c_float]
        _lib.mso_util_sin.restype = C.c_float
        _lib.mso_util_sin.argtypes = [C.c_float]
        _lib.mso_ic_angle.restype = C.c_float
        _lib.mso_bow_score_l1.restype = C.c_double
        _lib.mso_voc_load.restype = C.c_void_p
        _lib.mso_voc_load.argtypes = [C.c_void_p, C.c_size_t]
        _lib.mso_voc_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def params(n_levels=8, scale_factor=1.2, ini_fast_thr=20, min_fast_thr=7, min_size=1000):
    return Params(n_levels, scale_factor, ini_fast_thr, min_fast_thr, min_size)


def gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().mso_gray(_p(bgr), C.c_size_t(h * w), _p(out))
    return out


def level_geometry(W, H, p):
    w = (C.c_int * 16)()
    h = (C.c_int * 16)()
    s = (C.c_float * 16)()
    lib().mso_level_geometry(W, H, C.byref(p), w, h, s)
    n = p.n_levels
    return list(w[:n]), list(h[:n]), np.array(s[:n], np.float32)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().mso_resize_linear(_p(src), sw, sh, _p(dst), dw, dh)
    return dst


def resize_tables(ssize, dsize):
    ofs = np.empty(dsize, np.int32)
    coef = np.empty(2 * dsize, np.int16)
    lib().mso_resize_tables(ssize, dsize, _p(ofs), _p(coef))
    return ofs, coef


def pyramid(gray0, p):
    H, W = gray0.shape
    w, h, _ = level_geometry(W, H, p)
    pyr = [np.ascontiguousarray(gray0)]
    for l in range(1, p.n_levels):
        pyr.append(resize_linear(pyr[-1], w[l], h[l]))
    return pyr


def fast(img, threshold, cap=70 * 70):
    """cv::FAST on a (possibly strided) sub-image view."""
    assert img.dtype == np.uint8 and img.strides[1] == 1
    out = np.zeros(cap, CAND_DT)
    n = lib().mso_fast(_p(img), img.strides[0], img.shape[1], img.shape[0], threshold, _p(out), cap)
    return out[:n].copy()


def fast_level(img, p, cap=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = cap or (w * h // 4 + 16)
    out = np.zeros(cap, CAND_DT)
    n = lib().mso_fast_level(_p(img), w, h, C.byref(p), _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def quadtree(cand, w, h, scale_factor, min_size):
    cand = np.ascontiguousarray(cand, CAND_DT)
    out = np.zeros(max(len(cand), 1), CAND_DT)
    n = lib().mso_quadtree(_p(cand), len(cand), 19, w - 19, 19, h - 19, C.c_float(scale_factor),
                           C.c_uint(min_size), _p(out), len(out))
    return out[:n].copy()


def fast_atan2(y, x):
    return lib().mso_fast_atan2(C.c_float(y), C.c_float(x))


def util_cos(v):
    return lib().mso_util_cos(C.c_float(v))


def util_sin(v):
    return lib().mso_util_sin(C.c_float(v))


def umax():
    u = (C.c_int * 16)()
    lib().mso_umax(u)
    return list(u)


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().mso_ic_angle(_p(img), img.shape[1], int(x), int(y))


def gaussian_taps():
    t = (C.c_int * 7)()
    lib().mso_gaussian_kernel_fixed(t)
    return list(t)


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty_like(img)
    lib().mso_gaussian_blur7(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def orb_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.empty(32, np.uint8)
    lib().mso_orb_descriptor(_p(blurred), blurred.shape[1], int(x), int(y), C.c_float(angle_deg), _p(d))
    return d


def detect(bgr, p, max_out=100000):
    """Returns dict(xy[n,2] f32, desc[n,32] u8, octave[n] i32, angle[n] f32, response[n] f32)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W = bgr.shape[:2]
    xy = np.empty((max_out, 2), np.float32)
    desc = np.empty((max_out, 32), np.uint8)
    octave = np.empty(max_out, np.int32)
    angle = np.empty(max_out, np.float32)
    resp = np.empty(max_out, np.float32)
    n = C.c_int(0)
    rc = lib().mso_detect(_p(bgr), W, H, C.byref(p), max_out, _p(xy), _p(desc), _p(octave), _p(angle), _p(resp),
                          C.byref(n))
    if rc != 0:
        raise RuntimeError("oracle detect: capacity %d exceeded (%d keypoints)" % (max_out, n.value))
    k = n.value
    return dict(xy=xy[:k].copy(), desc=desc[:k].copy(), octave=octave[:k].copy(), angle=angle[:k].copy(),
                response=resp[:k].copy())


def match_knn2_raw(from_desc, to_desc):
    f = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
    n = len(t)
    i0 = np.empty(n, np.int32)
    i1 = np.empty(n, np.int32)
    d0 = np.empty(n, np.int32)
    d1 = np.empty(n, np.int32)
    lib().mso_match_knn2_raw(_p(f), len(f), _p(t), n, _p(i0), _p(i1), _p(d0), _p(d1))
    return i0, i1, d0, d1


def match(from_desc, to_desc, ratio=0.7):
    f = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
    fi = np.empty(max(len(t), 1), np.int32)
    ti = np.empty(max(len(t), 1), np.int32)
    n = lib().mso_match(_p(f), len(f), _p(t), len(t), C.c_double(ratio), _p(fi), _p(ti))
    return fi[:n].copy(), ti[:n].copy()


def backproject(depth, xy, factor=1.0 / 5000.0, focal=(525.0, 525.0), principal=(319.5, 239.5)):
    depth = np.ascontiguousarray(depth, np.uint16)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    h, w = depth.shape
    n = len(xy)
    xyz = np.zeros((max(n, 1), 3), np.float64)
    valid = np.zeros(max(n, 1), np.uint8)
    lib().mso_backproject(_p(depth), w, h, C.c_float(factor), C.c_double(focal[0]), C.c_double(focal[1]),
                          C.c_double(principal[0]), C.c_double(principal[1]), _p(xy), n, _p(xyz), _p(valid))
    return xyz[:n].copy(), valid[:n].astype(bool)


# ---- cv::ORB detector mode (OrbOpenCvDetector, orb_feature.cpp:25,33-65) -------------------------------------
class CvOrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int), ("scale_factor", C.c_float), ("n_levels", C.c_int),
                ("edge_threshold", C.c_int), ("fast_threshold", C.c_int), ("order", C.c_int)]


ORDER_LIBSTDCXX, ORDER_RASTER = 0, 1


def cvorb_params(n_features=1000, scale_factor=1.2, n_levels=8, edge_threshold=31, fast_threshold=20, order=ORDER_LIBSTDCXX):
    return CvOrbParams(n_features, scale_factor, n_levels, edge_threshold, fast_threshold, order)


def std_retain_best_order(response, n_points):
    """KeyPointsFilter::retainBest's std::nth_element + std::partition as libstdc++ runs them: indices of the survivors in
    their final places"""
    r = np.ascontiguousarray(response, np.float32)
    order = np.empty(max(len(r), 1), np.int32)
    m = lib().mso_std_retain_best_order(_p(r), len(r), int(n_points), _p(order))
    return order[:m].copy()


def cvorb_geometry(W, H, p):
    w = (C.c_int * 16)()
    h = (C.c_int * 16)()
    s = (C.c_float * 16)()
    q = (C.c_int * 16)()
    lib().mso_cvorb_geometry(W, H, C.byref(p), w, h, s, q)
    n = p.n_levels
    return list(w[:n]), list(h[:n]), np.array(s[:n], np.float32), list(q[:n])


def resize_linear_exact(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().mso_resize_linear_exact(_p(src), sw, sh, _p(dst), dw, dh)
    return dst


def cvorb_pyramid(gray0, p):
    H, W = gray0.shape
    w, h, _, _ = cvorb_geometry(W, H, p)
    pyr = [np.ascontiguousarray(gray0)]
    for l in range(1, p.n_levels):
        pyr.append(resize_linear_exact(pyr[-1], w[l], h[l]))
    return pyr


def harris_response(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    lib().mso_harris_response.restype = C.c_float
    return lib().mso_harris_response(_p(img), img.shape[1], int(x), int(y))


def cvorb_level_keypoints(img, p, quota, stage):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    out = np.zeros(cap, CAND_DT)
    n = lib().mso_cvorb_level_keypoints(_p(img), w, h, C.byref(p), int(quota), int(stage), _p(out), cap)
    return out[:n].copy()


def cvorb_detect(bgr, p, max_out=100000):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W = bgr.shape[:2]
    xy = np.empty((max_out, 2), np.float32)
    desc = np.empty((max_out, 32), np.uint8)
    octave = np.empty(max_out, np.int32)
    angle = np.empty(max_out, np.float32)
    resp = np.empty(max_out, np.float32)
    n = C.c_int(0)
    rc = lib().mso_cvorb_detect(_p(bgr), W, H, C.byref(p), max_out, _p(xy), _p(desc), _p(octave), _p(angle), _p(resp),
                                C.byref(n))
    if rc != 0:
        raise RuntimeError("oracle cvorb_detect: capacity %d exceeded (%d keypoints)" % (max_out, n.value))
    k = n.value
    return dict(xy=xy[:k].copy(), desc=desc[:k].copy(), octave=octave[:k].copy(), angle=angle[:k].copy(),
                response=resp[:k].copy())


def cvorb_descriptor(blurred, x, y, angle_deg):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    d = np.empty(32, np.uint8)
    lib().mso_cvorb_descriptor(_p(blurred), blurred.shape[1], int(x), int(y), C.c_float(angle_deg), _p(d))
    return d


def sincos_f32(x):
    s, c = C.c_float(), C.c_float()
    lib().mso_sincos_f32(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def libm_sincosf(x):
    s, c = C.c_float(), C.c_float()
    lib().mso_libm_sincosf(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


class Vocabulary:
    def __init__(self, blob):
        self._blob = np.frombuffer(bytes(blob), np.uint8) if not isinstance(blob, np.ndarray) else blob
        self._h = lib().mso_voc_load(_p(self._blob), C.c_size_t(self._blob.size))
        if not self._h:
            raise ValueError("oracle: not an uncompressed DBoW3 vocabulary stream")
        v = [C.c_int() for _ in range(6)]
        lib().mso_voc_info(C.c_void_p(self._h), *[C.byref(x) for x in v])
        self.k, self.L, self.n_nodes, self.n_words, self.scoring, self.weighting = [x.value for x in v]

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mso_voc_free(C.c_void_p(self._h))
            self._h = None

    def words(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.empty(len(d), np.uint32)
        wt = np.empty(len(d), np.float64)
        lib().mso_bow_words(C.c_void_p(self._h), _p(d), len(d), _p(w), _p(wt))
        return w, wt

    def words_flat(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.empty(len(d), np.uint32)
        wt = np.empty(len(d), np.float64)
        lib().mso_bow_words_flat(C.c_void_p(self._h), _p(d), len(d), _p(w), _p(wt))
        return w, wt

    def bow_vector(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.empty(max(len(d), 1), np.uint32)
        v = np.empty(max(len(d), 1), np.float64)
        m = lib().mso_bow_vector(C.c_void_p(self._h), _p(d), len(d), _p(w), _p(v))
        return w[:m].copy(), v[:m].copy()


def bow_score_l1(w1, v1, w2, v2):
    w1 = np.ascontiguousarray(w1, np.uint32)
    w2 = np.ascontiguousarray(w2, np.uint32)
    v1 = np.ascontiguousarray(v1, np.float64)
    v2 = np.ascontiguousarray(v2, np.float64)
    return lib().mso_bow_score_l1(_p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2))


def bench_stream(frames, p, n_threads, frames_per_thread, cv_params=None, max_kp=32768):
    """The timed CPU leg (mslam_cpu_bench.c): `n_threads` pthreads, each a detect + match-vs-previous-frame loop over
    its own block of `frames_per_thread` frames of `frames` (n, H, W, 3) taken cyclically.  Returns a dict with the
    keypoints and matches produced, the wall time and the shortest / longest thread loop."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape[:3]
    out = (C.c_double * 5)()
    rc = lib().mso_bench_stream(_p(frames), n, W, H, C.byref(p) if cv_params is None else None,
                                C.byref(cv_params) if cv_params is not None else None, int(n_threads),
                                int(frames_per_thread), int(max_kp), out)
    if rc == -2:
        raise RuntimeError("oracle bench_stream: could not start %d threads" % n_threads)
    if rc != 0:
        raise RuntimeError("oracle bench_stream: a frame exceeded %d keypoints" % max_kp)
    return {"keypoints": out[0], "matches": out[1], "seconds": out[2], "thread_seconds_min": out[3],
            "thread_seconds_max": out[4], "threads": int(n_threads), "frames": int(n_threads) * int(frames_per_thread)}
