"""CPU checks of the oracle's cv::ORB restatement (OrbOpenCvDetector, orb_feature.cpp:25,33-65 -> OpenCV 4.8.1
orb.cpp).  Nothing in the reference pins these outputs (parity unpinned); these are independent second
restatements (numpy / pure Python) of the published algorithms plus structural properties."""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def frame():
    import synth
    return synth.make_stream(1, 640, 480, seed=1234)[0]


def test_geometry_and_quota(orc):
    p = orc.cvorb_params()
    w, h, s, q = orc.cvorb_geometry(640, 480, p)
    # layerScale = (float)pow((double)1.2f, level); sizes = cvRound(dim * (1.0f / scale))
    sf = float(np.float32(1.2))
    for l in range(8):
        sc = np.float32(sf ** l)
        assert s[l] == sc
        inv = np.float32(1.0) / sc
        assert w[l] == int(np.rint(np.float32(640) * inv)) and h[l] == int(np.rint(np.float32(480) * inv))
    assert sum(q) == 1000 and q == sorted(q, reverse=True)
    assert q == [217, 181, 151, 126, 105, 87, 73, 60]          # the well-known cv::ORB(1000) split


def _exact_axis(ssize, dsize):
    scale = 1.0 / (dsize / ssize)
    out, mn, mx = [], 0, dsize
    for d in range(dsize):
        f = scale * (d + 0.5) - 0.5
        i = math.floor(f)
        o, c1 = 0, 0
        if i >= 0 and ssize > 1:
            if i < ssize - 1:
                o, c1 = i, int(np.rint((f - i) * 256.0))
            else:
                o, mx = ssize - 1, min(mx, d)
        else:
            mn = max(mn, d + 1)
        out.append((o, 256 - c1, c1))
    return out, mn, mx


@pytest.mark.parametrize("sw,sh,dw,dh", [(64, 48, 53, 40), (37, 29, 31, 24), (50, 40, 25, 20), (20, 16, 33, 27), (9, 7, 9, 7)])
def test_resize_linear_exact(orc, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 1000 + dw)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    got = orc.resize_linear_exact(src, dw, dh)
    xs, xmin, xmax = _exact_axis(sw, dw)
    ys, ymin, ymax = _exact_axis(sh, dh)

    def hline(row):
        return [int(row[0]) << 8 if x < xmin else int(row[-1]) << 8 if x >= xmax else
                xs[x][1] * int(row[xs[x][0]]) + xs[x][2] * int(row[xs[x][0] + 1]) for x in range(dw)]
    ref = np.empty((dh, dw), np.uint8)
    for y in range(dh):
        if y < ymin:
            ref[y] = [(v + 128) >> 8 for v in hline(src[0])]
        elif y >= ymax:
            ref[y] = [(v + 128) >> 8 for v in hline(src[-1])]
        else:
            o, b0, b1 = ys[y]
            h0, h1 = hline(src[o]), hline(src[o + 1])
            ref[y] = [(a * b0 + b * b1 + 32768) >> 16 for a, b in zip(h0, h1)]
    assert np.array_equal(got, ref)
    # and it is a bilinear resize: within 1 grey level of the real-valued interpolation at pixel centres
    fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
    fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, sh - 1), np.minimum(x0 + 1, sw - 1)
    wy, wx = (fy - y0)[:, None], (fx - x0)[None, :]
    s = src.astype(np.float64)
    real = (s[y0][:, x0] * (1 - wx) + s[y0][:, x1] * wx) * (1 - wy) + (s[y1][:, x0] * (1 - wx) + s[y1][:, x1] * wx) * wy
    assert np.abs(got.astype(np.float64) - real).max() <= 1.01
    if (sw, sh) == (dw, dh):
        assert np.array_equal(got, src)                      # identity size: coefficients (256, 0)


def test_harris_response(orc, frame):
    g = orc.gray(frame).astype(np.int64)
    rng = np.random.default_rng(3)
    f32 = np.float32
    for _ in range(40):
        x, y = int(rng.integers(8, 632)), int(rng.integers(8, 472))
        a = b = c = 0
        for i in range(-3, 4):
            for j in range(-3, 4):
                yy, xx = y + i, x + j
                ix = (g[yy, xx + 1] - g[yy, xx - 1]) * 2 + (g[yy - 1, xx + 1] - g[yy - 1, xx - 1]) + (g[yy + 1, xx + 1] - g[yy + 1, xx - 1])
                iy = (g[yy + 1, xx] - g[yy - 1, xx]) * 2 + (g[yy + 1, xx - 1] - g[yy - 1, xx - 1]) + (g[yy + 1, xx + 1] - g[yy - 1, xx + 1])
                a, b, c = a + ix * ix, b + iy * iy, c + ix * iy
        scale = f32(1.0) / f32(28 * f32(255.0))
        s4 = scale * scale * scale * scale
        fa, fb, fc = f32(a), f32(b), f32(c)
        ref = ((fa * fb - fc * fc) - (f32(0.04) * (fa + fb)) * (fa + fb)) * s4
        assert orc.harris_response(g.astype(np.uint8), x, y) == ref


def test_sincos_is_correctly_rounded_and_libm_is_close(orc):
    xs = (np.linspace(0, 360, 50001, dtype=np.float32)[:-1] * np.float32(math.pi / np.float32(180.0))).astype(np.float32)
    mine = np.array([orc.sincos_f32(float(x)) for x in xs], np.float32)
    assert np.array_equal(mine[:, 0], np.sin(xs.astype(np.float64)).astype(np.float32))
    assert np.array_equal(mine[:, 1], np.cos(xs.astype(np.float64)).astype(np.float32))
    libm = np.array([orc.libm_sincosf(float(x)) for x in xs], np.float32)
    # what the reference calls (host libm cosf/sinf) is within 1 ulp and differs in a few per cent of the arguments
    diff = (libm != mine).any(axis=1).mean()
    assert diff < 0.10
    assert np.abs(libm.view(np.int32).astype(np.int64) - mine.view(np.int32).astype(np.int64)).max() <= 1


def test_level_selection_is_retain_best(orc, frame):
    """the kept SET of both retainBest calls (raster order makes it comparable with a plain threshold); the library order
    (the default, tests/test_oracle_std_order.py) is a permutation of it"""
    p = orc.cvorb_params(order=orc.ORDER_RASTER)
    pyr = orc.cvorb_pyramid(orc.gray(frame), p)
    w, h, s, q = orc.cvorb_geometry(640, 480, p)
    for l in (0, 3, 7):
        img = pyr[l]
        kp = orc.fast(img, 20, cap=img.size // 4)
        e = 31
        kp = kp[(kp["x"] >= e) & (kp["x"] < w[l] - e) & (kp["y"] >= e) & (kp["y"] < h[l] - e)]
        st0 = orc.cvorb_level_keypoints(img, p, q[l], 0)
        if len(kp) > 2 * q[l]:
            thr = np.sort(kp["response"])[::-1][2 * q[l] - 1]
            kp = kp[kp["response"] >= thr]
        assert np.array_equal(st0, kp)                                     # raster order, ties kept
        st1 = orc.cvorb_level_keypoints(img, p, q[l], 1)
        hr = np.array([orc.harris_response(img, int(x), int(y)) for x, y in zip(kp["x"], kp["y"])], np.float32)
        if len(kp) > q[l]:
            keep = hr >= np.sort(hr)[::-1][q[l] - 1]
            kp, hr = kp[keep], hr[keep]
        assert np.array_equal(st1["x"], kp["x"]) and np.array_equal(st1["y"], kp["y"]) and np.array_equal(st1["response"], hr)
        assert len(st1) >= min(q[l], len(st0))
        lib1 = orc.cvorb_level_keypoints(img, orc.cvorb_params(), q[l], 1)   # default order: libstdc++'s
        assert sorted(map(tuple, lib1.tolist())) == sorted(map(tuple, st1.tolist()))


def test_detect_structure(orc, frame):
    p = orc.cvorb_params()
    d = orc.cvorb_detect(frame, p)
    w, h, s, q = orc.cvorb_geometry(640, 480, p)
    assert 900 <= len(d["xy"]) <= 1100
    for l in range(8):
        m = d["octave"] == l
        assert q[l] <= m.sum() <= q[l] + 20                                # quota + ties at the cut
        lx, ly = d["xy"][m, 0] / s[l], d["xy"][m, 1] / s[l]
        assert lx.min() >= 30.99 and lx.max() < w[l] - 30.99 and ly.min() >= 30.99 and ly.max() < h[l] - 30.99
    assert (d["angle"] >= 0).all() and (d["angle"] < 360).all()
    # a descriptor is the steered BRIEF of its blurred level
    pyr = orc.cvorb_pyramid(orc.gray(frame), p)
    k = int(np.nonzero(d["octave"] == 2)[0][5])
    bl = orc.gaussian_blur7(pyr[2])
    x, y = int(np.rint(d["xy"][k, 0] / s[2])), int(np.rint(d["xy"][k, 1] / s[2]))
    assert np.array_equal(orc.cvorb_descriptor(bl, x, y, float(d["angle"][k])), d["desc"][k])
    assert d["angle"][k] == orc.ic_angle(pyr[2], x, y)


def test_product_sincos_equals_the_c_library_over_the_whole_domain(tmp_path):
    """include/mslam_sincos.h against the real C library's (float)cos((double)x) / (float)sin((double)x) — the expression
    OpenCV's computeOrbDescriptors evaluates in a GCC build — for every 97th float in [0, 6.5] (11 million arguments;
    MSLAM_EXHAUSTIVE=1: all 1 087 373 313 of them, 25 s: 0 mismatches against glibc 2.35, profiles/r05_g_sincos_exhaustive.txt)"""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "oracle", "sincos_check", "sincos_exhaustive.c")
    exe = str(tmp_path / "sincos_exhaustive")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"])
    stride = "1" if os.environ.get("MSLAM_EXHAUSTIVE") == "1" else "97"
    r = subprocess.run([exe, stride], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    f = dict(zip(r.stdout.split()[0::2], r.stdout.split()[1::2]))
    assert int(f["mismatches"]) == 0 and int(f["floats"]) > 10_000_000
    # the float routines of the same library (what a build resolving to cosf / sinf would call) differ in the last bit for
    # about one argument in a thousand
    assert 0 < int(f["libm_float_routines_differ"]) < int(f["floats"]) // 200
