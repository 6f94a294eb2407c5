"""modular-slam_amd — MI355X-native ORB / Hamming-match / BoW front end for modular-slam.

This package is the thin Python harness over the C ABI (include/mslam_hip.h) implemented by
libmslam_hip.so (HIP kernels for gfx950, modular-slam_amd/csrc/).  It mirrors the reference's plugin
interfaces for this path (feature_interface.hpp:50-70, relocalizer.hpp:11-20,
loop_detection.hpp:10-15) with the same names and argument meaning:

    HipOrbDetector.detect(rgb_frame)          ~ IFeatureDetector<RgbFrame,uint8_t,32>::detect
    HipOrbMatcher.match(first, second)        ~ IFeatureMatcher<uint8_t,32>::match
    HipOrbRelocalizer.addKeyframe/relocalize  ~ IRelocalizer<...>
    HipLoopDetector.detectLoop()              ~ ILoopDetector<...>

There is NO CPU fallback: if the shared library is missing, or no HIP device is present, calls
raise.  (The directory name has a hyphen, so import it through `__graft_entry__.load_package()`,
which registers it as module `modular_slam_amd`.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmslam_hip.so")

OK, E_INVALID, E_RUNTIME, E_CAPACITY, E_NO_VOCABULARY, E_FORMAT, E_NO_MODEL = range(7)
DBG_PYRAMID, DBG_BLURRED, DBG_CANDIDATES, DBG_SELECTED = range(4)
MATCHER_AUTO, MATCHER_POPCOUNT = 0, 1
DETECTOR_DISTRIBUTED, DETECTOR_CV_ORB = 0, 1
BOW_ASSIGN_TREE, BOW_ASSIGN_FLAT = 0, 1
CV_ORDER_LIBSTDCXX, CV_ORDER_RASTER = 0, 1

# every symbol include/mslam_hip.h declares (tests/test_cabi.py checks the .so exports them all)
ABI_SYMBOLS = [
    "mslam_hip_default_params", "mslam_hip_abi_version", "mslam_hip_create", "mslam_hip_destroy",
    "mslam_hip_last_error", "mslam_hip_sync", "mslam_hip_detect", "mslam_hip_detect_batch_dev",
    "mslam_hip_get_batch_view", "mslam_hip_match", "mslam_hip_match_knn2", "mslam_hip_match_batch_dev",
    "mslam_hip_bow_load", "mslam_hip_bow_info", "mslam_hip_bow_words", "mslam_hip_bow_transform",
    "mslam_hip_bow_score", "mslam_hip_bow_db_add", "mslam_hip_bow_db_query", "mslam_hip_bow_db_clear",
    "mslam_hip_bow_batch_dev", "mslam_hip_get_bow_view", "mslam_hip_bow_cross_score_dev", "mslam_hip_level_geometry", "mslam_hip_debug_read",
    "mslam_hip_set_profiling", "mslam_hip_get_stage_times", "mslam_hip_copy_to_host", "mslam_hip_backproject", "mslam_hip_backproject_batch_dev",
    "mslam_hip_get_points_view", "mslam_hip_set_matcher", "mslam_hip_get_matcher",
    "mslam_hip_bow_pack_dev", "mslam_hip_bow_cross_score_packed_dev", "mslam_hip_debug_counts",
    "mslam_hip_last_match_kernel",
    "mslam_hip_join_matcher", "mslam_hip_bow_db_remove", "mslam_hip_bow_set_assignment",
    "mslam_hip_bow_db_reserve", "mslam_hip_bow_db_size", "mslam_hip_qlz_decompress",
    "mslam_hip_pnp_ransac", "mslam_hip_pnp_batch_dev", "mslam_hip_get_pnp_view", "mslam_hip_pnp_set_confidence", "mslam_hip_pack_batch_dev", "mslam_hip_packed_capacity",
    "mslam_hip_set_cv_keypoint_order",
]


class MslamHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mslam_hip error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32), ("n_levels", C.c_int32),
                ("scale_factor", C.c_float), ("ini_fast_thr", C.c_int32), ("min_fast_thr", C.c_int32),
                ("min_node_area", C.c_uint32), ("max_keypoints", C.c_int32), ("max_candidates", C.c_int32),
                ("device", C.c_int32), ("stream", C.c_void_p), ("detector", C.c_int32), ("n_features", C.c_int32),
                ("edge_threshold", C.c_int32)]


class BatchView(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("capacity", C.c_int32), ("xy", C.c_void_p), ("desc", C.c_void_p),
                ("octave", C.c_void_p), ("angle", C.c_void_p), ("response", C.c_void_p), ("count", C.c_void_p),
                ("match_from", C.c_void_p), ("match_to", C.c_void_p), ("match_count", C.c_void_p)]


class PointsView(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("xyz", C.c_void_p), ("valid", C.c_void_p)]


class PackedHeader(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("total_keypoints", C.c_int32), ("total_matches", C.c_int32), ("with_points", C.c_int32),
                ("off_kp_offset", C.c_uint64), ("off_match_offset", C.c_uint64), ("off_xy", C.c_uint64), ("off_desc", C.c_uint64),
                ("off_octave", C.c_uint64), ("off_angle", C.c_uint64), ("off_response", C.c_uint64), ("off_xyz", C.c_uint64),
                ("off_valid", C.c_uint64), ("off_match_from", C.c_uint64), ("off_match_to", C.c_uint64), ("bytes", C.c_uint64),
                ("fits", C.c_int32), ("pad", C.c_int32)]


def unpack_batch(buf):
    """numpy views of a buffer written by mslam_hip_pack_batch_dev (buf: 1-D uint8 array holding at least header.bytes)"""
    h = PackedHeader.from_buffer_copy(bytes(buf[:C.sizeof(PackedHeader)]))
    if not h.fits:
        raise MslamHipError(E_CAPACITY, "packed results need %d bytes" % h.bytes)
    nk, nm, nf = h.total_keypoints, h.total_matches, h.n_frames

    def arr(off, dt, n, *shape):
        return np.frombuffer(buf, dt, n, int(off)).reshape((-1,) + shape) if shape else np.frombuffer(buf, dt, n, int(off))
    out = {"n_frames": nf, "bytes": int(h.bytes),
           "kp_offset": arr(h.off_kp_offset, np.int32, nf + 1), "match_offset": arr(h.off_match_offset, np.int32, nf + 1),
           "xy": arr(h.off_xy, np.float32, 2 * nk, 2), "desc": arr(h.off_desc, np.uint8, 32 * nk, 32),
           "octave": arr(h.off_octave, np.int32, nk), "angle": arr(h.off_angle, np.float32, nk),
           "response": arr(h.off_response, np.float32, nk),
           "match_from": arr(h.off_match_from, np.int32, nm), "match_to": arr(h.off_match_to, np.int32, nm)}
    if h.with_points:
        out["xyz"] = arr(h.off_xyz, np.float64, 3 * nk, 3)
        out["valid"] = arr(h.off_valid, np.uint8, nk)
    return out


class PnpView(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("pose", C.c_void_p), ("n_points", C.c_void_p), ("object_points", C.c_void_p),
                ("image_points", C.c_void_p), ("inliers", C.c_void_p)]


class BowView(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("words", C.c_void_p), ("values", C.c_void_p), ("n_words", C.c_void_p),
                ("best_entry", C.c_void_p), ("best_score", C.c_void_p)]


def build(verbose=False):
    """Compile every HIP source for gfx950 into libmslam_hip.so (in-tree)."""
    subprocess.check_call(["make", "-j8", "-C", _HERE] + ([] if verbose else ["-s"]))


_lib = None


def lib():
    """Load the HIP library.  Fails loudly when it has not been built — there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MslamHipError(E_RUNTIME, "libmslam_hip.so is not built (run __graft_entry__.build()); "
                                           "the product path has no CPU fallback")
        # PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7).  Two HIP runtimes in one
        # process cannot both open the GPU, so when torch is available load it FIRST: our NEEDED
        # libamdhip64.so.7 then binds to the runtime torch already mapped.  Without torch (pure C/C++
        # hosts) the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mslam_hip_last_error.restype = C.c_char_p
        L.mslam_hip_last_error.argtypes = [C.c_void_p]
        L.mslam_hip_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.mslam_hip_destroy.argtypes = [C.c_void_p]
        L.mslam_hip_destroy.restype = None
        L.mslam_hip_packed_capacity.restype = C.c_size_t
        L.mslam_hip_packed_capacity.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mslam_hip_pack_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = Params()
    lib().mslam_hip_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


class Context:
    """Owner of one mslam_hip_ctx (one per GPU / caller thread)."""

    def __init__(self, **kw):
        self.params = default_params(**kw)
        h = C.c_void_p()
        rc = lib().mslam_hip_create(C.byref(self.params), C.byref(h))
        if rc != OK:
            raise MslamHipError(rc, (lib().mslam_hip_last_error(None) or b"").decode())
        self._h = h
        self.L = lib()

    def close(self):
        if getattr(self, "_h", None):
            self.L.mslam_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise MslamHipError(rc, (self.L.mslam_hip_last_error(self._h) or b"").decode())

    # ---- detector ------------------------------------------------------------------------------
    def detect(self, bgr, max_out=None):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        H, W = bgr.shape[:2]
        max_out = max_out or self.params.max_keypoints
        xy = np.empty((max_out, 2), np.float32)
        desc = np.empty((max_out, 32), np.uint8)
        octave = np.empty(max_out, np.int32)
        angle = np.empty(max_out, np.float32)
        resp = np.empty(max_out, np.float32)
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_detect(self._h, _p(bgr), W, H, max_out, _p(xy), _p(desc), _p(octave), _p(angle),
                                          _p(resp), C.byref(n)))
        k = n.value
        return dict(xy=xy[:k].copy(), desc=desc[:k].copy(), octave=octave[:k].copy(), angle=angle[:k].copy(),
                    response=resp[:k].copy())

    def detect_batch_dev(self, d_bgr_ptr, n_frames):
        self._chk(self.L.mslam_hip_detect_batch_dev(self._h, C.c_void_p(d_bgr_ptr), int(n_frames)))

    def match_batch_dev(self, ratio=0.7, chain_previous=True):
        self._chk(self.L.mslam_hip_match_batch_dev(self._h, C.c_double(ratio), int(bool(chain_previous))))

    def sync(self):
        self._chk(self.L.mslam_hip_sync(self._h))

    def batch_view(self):
        v = BatchView()
        self._chk(self.L.mslam_hip_get_batch_view(self._h, C.byref(v)))
        return v

    # ---- matcher -------------------------------------------------------------------------------
    def match(self, from_desc, to_desc, ratio=0.7):
        f = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
        fi = np.empty(max(len(t), 1), np.int32)
        ti = np.empty(max(len(t), 1), np.int32)
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_match(self._h, _p(f), len(f), _p(t), len(t), C.c_double(ratio), _p(fi), _p(ti),
                                         C.byref(n)))
        return fi[:n.value].copy(), ti[:n.value].copy()

    def join_matcher(self):
        self._chk(self.L.mslam_hip_join_matcher(self._h))

    def set_matcher(self, kind):
        """MATCHER_AUTO (matrix cores up to 32736 train rows) or MATCHER_POPCOUNT (xor/popcount always)."""
        self._chk(self.L.mslam_hip_set_matcher(self._h, int(kind)))

    def get_matcher(self):
        return self.L.mslam_hip_get_matcher(self._h)

    def set_cv_keypoint_order(self, order):
        """CV_ORDER_LIBSTDCXX (default: the order of a GCC build of the reference) or CV_ORDER_RASTER (FAST's order)."""
        self._chk(self.L.mslam_hip_set_cv_keypoint_order(self._h, int(order)))

    def last_match_kernel(self):
        """'matrix' / 'popcount': the kernel the last matcher launch took (None before the first)"""
        return {1: "matrix", 2: "popcount"}.get(self.L.mslam_hip_last_match_kernel(self._h))

    def match_knn2(self, from_desc, to_desc):
        f = np.ascontiguousarray(from_desc, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(to_desc, np.uint8).reshape(-1, 32)
        n = max(len(t), 1)
        out = [np.empty(n, np.int32) for _ in range(4)]
        self._chk(self.L.mslam_hip_match_knn2(self._h, _p(f), len(f), _p(t), len(t), *[_p(o) for o in out]))
        return tuple(o[:len(t)].copy() for o in out)

    # ---- RGB-D back-projection (rgbd_feature_frontend.cpp:101-138) ------------------------------
    def backproject(self, depth, xy, factor=1.0 / 5000.0, focal=(525.0, 525.0), principal=(319.5, 239.5)):
        depth = np.ascontiguousarray(depth, np.uint16)
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        h, w = depth.shape
        n = len(xy)
        xyz = np.zeros((max(n, 1), 3), np.float64)
        valid = np.zeros(max(n, 1), np.uint8)
        self._chk(self.L.mslam_hip_backproject(self._h, _p(depth), w, h, C.c_float(factor), C.c_double(focal[0]),
                                               C.c_double(focal[1]), C.c_double(principal[0]), C.c_double(principal[1]),
                                               _p(xy), n, _p(xyz), _p(valid)))
        return xyz[:n].copy(), valid[:n].astype(bool)

    def backproject_batch_dev(self, d_depth_ptr, factor=1.0 / 5000.0, focal=(525.0, 525.0), principal=(319.5, 239.5)):
        self._chk(self.L.mslam_hip_backproject_batch_dev(self._h, C.c_void_p(d_depth_ptr), C.c_float(factor),
                                                         C.c_double(focal[0]), C.c_double(focal[1]),
                                                         C.c_double(principal[0]), C.c_double(principal[1])))

    def pack_batch_dev(self, out_ptr, capacity_bytes, with_points=True):
        """exactly count[t] keypoint / match_count[t] match records per frame of the last batch, back to back, into out_ptr
        (device memory or page-locked mapped host memory) on the context's stream; parse with unpack_batch()"""
        self._chk(self.L.mslam_hip_pack_batch_dev(self._h, C.c_void_p(out_ptr), C.c_size_t(capacity_bytes), 1 if with_points else 0))

    def packed_capacity(self, n_frames, with_points=True):
        return int(self.L.mslam_hip_packed_capacity(self._h, int(n_frames), 1 if with_points else 0))

    def points_view(self):
        v = PointsView()
        self._chk(self.L.mslam_hip_get_points_view(self._h, C.byref(v)))
        return v

    def pnp_batch_dev(self, focal=(525.0, 525.0), principal=(319.5, 239.5), iterations=100, reprojection_error=5.0, seed=0):
        """one RANSAC PnP per frame of the last batch from its matches + the previous frame's back-projected points
        (after detect_batch_dev, match_batch_dev, backproject_batch_dev); results: pnp_view()"""
        self._chk(self.L.mslam_hip_pnp_batch_dev(self._h, C.c_double(focal[0]), C.c_double(focal[1]), C.c_double(principal[0]),
                                                 C.c_double(principal[1]), int(iterations), C.c_double(reprojection_error),
                                                 C.c_uint64(seed)))

    def pnp_set_confidence(self, confidence):
        """RANSAC confidence of pnp_ransac / pnp_batch_dev (default 0.99, cv_ransac_pnp.cpp:57); outside (0, 1): no early exit"""
        self._chk(self.L.mslam_hip_pnp_set_confidence(self._h, C.c_double(confidence)))

    def pnp_view(self):
        v = PnpView()
        self._chk(self.L.mslam_hip_get_pnp_view(self._h, C.byref(v)))
        return v

    # ---- PnP RANSAC (cv_ransac_pnp.cpp:14-85) ---------------------------------------------------
    def pnp_ransac(self, object_points, image_points, focal=(525.0, 525.0), principal=(319.5, 239.5), rvec=None,
                   tvec=None, iterations=100, reprojection_error=5.0, seed=0):
        """-> (rvec[3], tvec[3], inlier mask[n]) of the world -> camera transform, or None when no model was found
        (cv::solvePnPRansac returning false).  rvec/tvec given = useExtrinsicGuess."""
        obj = np.ascontiguousarray(object_points, np.float32).reshape(-1, 3)
        img = np.ascontiguousarray(image_points, np.float32).reshape(-1, 2)
        guess = rvec is not None and tvec is not None
        r = np.array(rvec if guess else (0, 0, 0), np.float64)
        t = np.array(tvec if guess else (0, 0, 0), np.float64)
        mask = np.zeros(len(obj), np.uint8)
        n_in = C.c_int(0)
        rc = self.L.mslam_hip_pnp_ransac(self._h, _p(obj), _p(img), len(obj), C.c_double(focal[0]), C.c_double(focal[1]),
                                         C.c_double(principal[0]), C.c_double(principal[1]), int(guess), int(iterations),
                                         C.c_double(reprojection_error), C.c_uint64(seed), _p(r), _p(t), _p(mask),
                                         C.byref(n_in))
        if rc == E_NO_MODEL:
            return None
        self._chk(rc)
        return r, t, mask.astype(bool)

    # ---- bag of words --------------------------------------------------------------------------
    def bow_load(self, blob):
        b = np.frombuffer(bytes(blob), np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob)
        self._chk(self.L.mslam_hip_bow_load(self._h, _p(b), C.c_size_t(b.size)))

    def bow_info(self):
        v = [C.c_int() for _ in range(6)]
        self._chk(self.L.mslam_hip_bow_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("k", "L", "n_nodes", "n_words", "scoring", "weighting"), [x.value for x in v]))

    def bow_set_assignment(self, mode):
        """BOW_ASSIGN_TREE (DBoW3's descent) or BOW_ASSIGN_FLAT (exhaustive search over all words)"""
        self._chk(self.L.mslam_hip_bow_set_assignment(self._h, int(mode)))

    def bow_words(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.empty(max(len(d), 1), np.uint32)
        wt = np.empty(max(len(d), 1), np.float64)
        self._chk(self.L.mslam_hip_bow_words(self._h, _p(d), len(d), _p(w), _p(wt)))
        return w[:len(d)].copy(), wt[:len(d)].copy()

    def bow_transform(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.empty(max(len(d), 1), np.uint32)
        v = np.empty(max(len(d), 1), np.float64)
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_bow_transform(self._h, _p(d), len(d), _p(w), _p(v), C.byref(n)))
        return w[:n.value].copy(), v[:n.value].copy()

    def bow_score(self, w1, v1, w2, v2):
        w1 = np.ascontiguousarray(w1, np.uint32)
        w2 = np.ascontiguousarray(w2, np.uint32)
        v1 = np.ascontiguousarray(v1, np.float64)
        v2 = np.ascontiguousarray(v2, np.float64)
        s = C.c_double(0)
        self._chk(self.L.mslam_hip_bow_score(self._h, _p(w1), _p(v1), len(w1), _p(w2), _p(v2), len(w2), C.byref(s)))
        return s.value

    def bow_db_add(self, desc):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        e = C.c_int(-1)
        self._chk(self.L.mslam_hip_bow_db_add(self._h, _p(d), len(d), C.byref(e)))
        return e.value

    def bow_db_query(self, desc, max_results=4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ids = np.empty(max(max_results, 1), np.int32)
        sc = np.empty(max(max_results, 1), np.float64)
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_bow_db_query(self._h, _p(d), len(d), max_results, _p(ids), _p(sc), C.byref(n)))
        return ids[:n.value].copy(), sc[:n.value].copy()

    def bow_db_remove(self, entry_id):
        self._chk(self.L.mslam_hip_bow_db_remove(self._h, int(entry_id)))

    def bow_db_reserve(self, max_entries):
        self._chk(self.L.mslam_hip_bow_db_reserve(self._h, int(max_entries)))

    def bow_db_size(self):
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_bow_db_size(self._h, C.byref(n)))
        return n.value

    def bow_db_clear(self):
        self._chk(self.L.mslam_hip_bow_db_clear(self._h))

    def bow_batch_dev(self, add_to_db=True):
        self._chk(self.L.mslam_hip_bow_batch_dev(self._h, int(bool(add_to_db))))

    def bow_cross_score_dev(self, d_words, d_values, d_n, n_sets, capacity, d_scores):
        self._chk(self.L.mslam_hip_bow_cross_score_dev(self._h, C.c_void_p(d_words), C.c_void_p(d_values),
                                                       C.c_void_p(d_n), int(n_sets), int(capacity),
                                                       C.c_void_p(d_scores)))

    def bow_pack_dev(self, k_max, d_out):
        self._chk(self.L.mslam_hip_bow_pack_dev(self._h, int(k_max), C.c_void_p(d_out)))

    def bow_cross_score_packed_dev(self, d_sets, n_sets, self_set, n_frames, k_max, d_scores, stream=None):
        self._chk(self.L.mslam_hip_bow_cross_score_packed_dev(self._h, C.c_void_p(d_sets), int(n_sets), int(self_set),
                                                              int(n_frames), int(k_max), C.c_void_p(d_scores),
                                                              C.c_void_p(stream)))

    def bow_view(self):
        v = BowView()
        self._chk(self.L.mslam_hip_get_bow_view(self._h, C.byref(v)))
        return v

    # ---- debug ---------------------------------------------------------------------------------
    def level_geometry(self):
        n = self.params.n_levels
        w = (C.c_int * 16)()
        h = (C.c_int * 16)()
        s = (C.c_float * 16)()
        self._chk(self.L.mslam_hip_level_geometry(self._h, w, h, s))
        return list(w[:n]), list(h[:n]), np.array(s[:n], np.float32)

    def debug_image(self, what, frame, level):
        w, h, _ = self.level_geometry()
        out = np.empty((h[level], w[level]), np.uint8)
        n = C.c_size_t(0)
        self._chk(self.L.mslam_hip_debug_read(self._h, what, frame, level, _p(out), C.c_size_t(out.size), C.byref(n)))
        return out

    def debug_keypoints(self, what, frame, level):
        out = np.empty((self.params.max_candidates, 3), np.float32)
        n = C.c_size_t(0)
        self._chk(self.L.mslam_hip_debug_read(self._h, what, frame, level, _p(out), C.c_size_t(out.nbytes),
                                              C.byref(n)))
        return out[:n.value].copy()

    def debug_counts(self, what, n_frames):
        """[n_frames, n_levels] FAST candidates (DBG_CANDIDATES) or selected keypoints (DBG_SELECTED) of the last batch"""
        out = np.zeros((max(n_frames, 1), self.params.n_levels), np.int32)
        self._chk(self.L.mslam_hip_debug_counts(self._h, int(what), _p(out), int(out.shape[0])))
        return out[:n_frames]

    def set_profiling(self, enable):
        """0 = off, 1/True = every stage, serialised on one stream, 2 = every stage launch, in place."""
        self._chk(self.L.mslam_hip_set_profiling(self._h, int(enable)))

    def stage_times(self, cap=256):
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = C.c_int(0)
        self._chk(self.L.mslam_hip_get_stage_times(self._h, names, ms, cap, C.byref(n)))
        return [(names[i].decode(), ms[i]) for i in range(n.value)]


def qlz_decompress(data, n_packets, capacity=None):
    """host-only: decode consecutive QuickLZ packets (what follows nChunks in a compressed DBoW3 vocabulary)"""
    src = np.frombuffer(bytes(data), np.uint8)
    cap = capacity if capacity is not None else 10000 * n_packets + 16
    dst = np.empty(max(cap, 1), np.uint8)
    n = C.c_size_t(0)
    L = lib()
    L.mslam_hip_qlz_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    rc = L.mslam_hip_qlz_decompress(_p(src), src.size, int(n_packets), _p(dst), dst.size, C.byref(n))
    if rc != OK:
        raise MslamHipError(rc, "qlz_decompress failed")
    return dst[:n.value].tobytes()


# ---- mirrors of the reference plugin interfaces ---------------------------------------------------
class RgbFrame:
    """types/rgb_frame.hpp:12-16 — interleaved 3-channel bytes (B,G,R as the providers deliver them)."""

    def __init__(self, data, width, height):
        self.data = np.ascontiguousarray(data, np.uint8).reshape(height, width, 3)
        self.size = (width, height)


class OrbKeypoint:
    """KeypointDescriptor<uint8_t,32> (feature_interface.hpp:18-30): keypoint {id, coordinates} + descriptor."""
    __slots__ = ("id", "coordinates", "descriptor")

    def __init__(self, id, coordinates, descriptor):
        self.id, self.coordinates, self.descriptor = id, coordinates, descriptor


class HipOrbDetector:
    """IFeatureDetector<RgbFrame, uint8_t, 32> backed by the HIP extractor
    (drop-in for DistributedOrbOpenCvDetector, distributed_cv_feature.cpp:1181-1222)."""

    def __init__(self, width=640, height=480, **kw):
        self.ctx = Context(width=width, height=height, **kw)

    def detect(self, sensorData):
        r = self.ctx.detect(sensorData.data)
        # id = running index, coordinates widened to double (distributed_cv_feature.cpp:1203-1208)
        return [OrbKeypoint(i, (float(x), float(y)), d) for i, ((x, y), d) in enumerate(zip(r["xy"].astype(np.float64),
                                                                                          r["desc"]))]


class HipOrbMatcher:
    """IFeatureMatcher<uint8_t, 32> (drop-in for OrbOpenCvMatcher, orb_feature.cpp:84-130)."""

    def __init__(self, ctx=None, ratio=0.7):
        self.ctx = ctx or Context()
        self.ratio = ratio

    def match(self, firstDescriptors, secondDescriptors):
        f = np.array([k.descriptor for k in firstDescriptors], np.uint8).reshape(-1, 32)
        t = np.array([k.descriptor for k in secondDescriptors], np.uint8).reshape(-1, 32)
        fi, ti = self.ctx.match(f, t, self.ratio)
        return [(int(a), int(b)) for a, b in zip(fi, ti)]  # DescriptorMatch{fromIndex, toIndex}


class HipOrbRelocalizer:
    """IRelocalizer<SensorState, uint8_t, 32> over the DBoW3 database kernels
    (what OrbRelocalizer is wired for, orb_relocalizer.cpp:26-50)."""

    def __init__(self, vocabulary_blob, ctx=None, max_results=4):
        self.ctx = ctx or Context()
        self.ctx.bow_load(vocabulary_blob)
        self.max_results = max_results
        self._entry_to_keyframe = {}

    def addKeyframe(self, keyframe, keypoints):
        assert len(keypoints) > 0  # orb_relocalizer.cpp:42
        d = np.array([k.descriptor for k in keypoints], np.uint8).reshape(-1, 32)
        self._entry_to_keyframe[self.ctx.bow_db_add(d)] = keyframe

    def removeKeyframe(self, keyframe):
        for e, k in list(self._entry_to_keyframe.items()):
            if k is keyframe:
                del self._entry_to_keyframe[e]
                self.ctx.bow_db_remove(e)

    def relocalize(self, keypoints):
        d = np.array([k.descriptor for k in keypoints], np.uint8).reshape(-1, 32)
        ids, _ = self.ctx.bow_db_query(d, self.max_results + len(self._entry_to_keyframe))
        out = [self._entry_to_keyframe[i] for i in ids if i in self._entry_to_keyframe]
        return out[:self.max_results]


class HipLoopDetector:
    """ILoopDetector<State>: detectLoop() takes no arguments (loop_detection.hpp:10-15), so it is fed
    through the relocalizer's addKeyframe; it reports the best-scoring earlier keyframe of the most
    recently added one, or None."""

    def __init__(self, relocalizer, min_score=0.05, exclude_recent=1):
        self.reloc = relocalizer
        self.min_score = min_score
        self.exclude_recent = exclude_recent
        self._last = None

    def feed(self, keyframe, keypoints):
        d = np.array([k.descriptor for k in keypoints], np.uint8).reshape(-1, 32)
        n_db = len(self.reloc._entry_to_keyframe)
        ids, sc = self.reloc.ctx.bow_db_query(d, n_db) if n_db else (np.empty(0, np.int32), np.empty(0))
        self._last = None
        for i, s in zip(ids, sc):
            if s >= self.min_score and i < n_db - self.exclude_recent and i in self.reloc._entry_to_keyframe:
                self._last = self.reloc._entry_to_keyframe[i]
                break
        self.reloc.addKeyframe(keyframe, keypoints)

    def detectLoop(self):
        return self._last


# ---- harness helper (bench / tests): copy a context-owned device array to the host ------------------
def read_device(ctx, ptr, shape, dtype):
    """Copy a device array (raw pointer from a *_view struct) into a new numpy array."""
    out = np.empty(shape, dtype)
    if out.nbytes:
        ctx._chk(ctx.L.mslam_hip_copy_to_host(ctx._h, _p(out), C.c_void_p(ptr), C.c_size_t(out.nbytes)))
    return out
