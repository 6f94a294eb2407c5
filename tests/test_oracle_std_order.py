"""The cv::ORB mode's keypoint ORDER: KeyPointsFilter::retainBest leaves its survivors where std::nth_element +
std::partition put them.  oracle/mslam_oracle.c restates libstdc++'s introselect / partition (the reference is a GCC
build); here that restatement is compared with the REAL <algorithm> of this image (g++, oracle/std_order_check/std_order.cpp)
on random, tie-heavy, sorted and depth-limit-provoking inputs — the library itself pins this part of the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mslam_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def real():
    src = os.path.join(ROOT, "oracle", "std_order_check", "std_order.cpp")
    out = os.path.join(ROOT, "oracle", "std_order_check", "libstd_order.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    lib = C.CDLL(out)

    def run(resp, n_points):
        r = np.ascontiguousarray(resp, np.float32)
        order = np.empty(max(len(r), 1), np.int32)
        m = lib.real_std_retain_best_order(r.ctypes.data_as(C.c_void_p), len(r), int(n_points), order.ctypes.data_as(C.c_void_p))
        return order[:m].copy()
    return run


def _median_of_3_killer(n):
    """the classic input that drives median-of-3 quickselect into its depth limit (-> the heap_select branch)"""
    a = np.zeros(n, np.float32)
    k = n // 2
    for i in range(1, k + 1):
        if i % 2 == 1:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_restatement_equals_the_real_library(real):
    rng = np.random.default_rng(11)
    cases = 0
    for n in list(range(0, 12)) + [17, 64, 65, 100, 257, 1000, 2049, 4096]:
        for kind in range(6):
            if kind == 0:
                r = rng.random(n).astype(np.float32)                          # all distinct (Harris responses)
            elif kind == 1:
                r = rng.integers(20, 60, n).astype(np.float32)                # heavy ties (FAST scores)
            elif kind == 2:
                r = np.sort(rng.integers(0, 255, n)).astype(np.float32)       # ascending
            elif kind == 3:
                r = np.sort(rng.integers(0, 255, n))[::-1].astype(np.float32)  # descending
            elif kind == 4:
                r = np.full(n, 7, np.float32)                                 # one value
            else:
                r = _median_of_3_killer(n)[:n] if n >= 4 else rng.random(n).astype(np.float32)
            for n_points in sorted({0, 1, 2, n // 3, n // 2, n - 1, n, n + 5}):
                if n_points < 0:
                    continue
                a, b = orc.std_retain_best_order(r, n_points), real(r, n_points)
                assert np.array_equal(a, b), (n, kind, n_points)
                if 0 < n_points < n:      # the SET is the standard's: every response >= the n-th largest
                    thr = np.sort(r)[::-1][n_points - 1]
                    assert sorted(a.tolist()) == sorted(np.nonzero(r >= thr)[0].tolist())
                cases += 1
    assert cases > 500


def test_depth_limit_branch_is_exercised(real):
    """the heap_select fallback of introselect runs on the adversarial input (and agrees with the library there, too)"""
    before = orc.lib().mso_std_heap_select_calls()
    for n in (64, 200, 1000, 4000):
        r = _median_of_3_killer(n)
        for n_points in (n // 2, n // 2 + 1, 3 * n // 4):
            assert np.array_equal(orc.std_retain_best_order(r, n_points), real(r, n_points))
    assert orc.lib().mso_std_heap_select_calls() > before


def test_cv_orb_levels_in_library_order():
    """mso_cvorb_level_keypoints in both orders: the same set, the library order is a permutation of the raster order"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth
    f = synth.make_stream(1, 640, 480, seed=1234)[0]
    g = orc.gray(f)
    for stage in (0, 1):
        a = orc.cvorb_level_keypoints(g, orc.cvorb_params(order=orc.ORDER_LIBSTDCXX), 217, stage)
        b = orc.cvorb_level_keypoints(g, orc.cvorb_params(order=orc.ORDER_RASTER), 217, stage)
        assert len(a) == len(b) and len(a) >= 217
        key = lambda k: (k["y"], k["x"])  # noqa: E731
        assert sorted(map(tuple, a.tolist()), key=lambda t: (t[1], t[0])) == sorted(map(tuple, b.tolist()), key=lambda t: (t[1], t[0]))
        assert not np.array_equal(a, b)   # ... and it IS a different order
