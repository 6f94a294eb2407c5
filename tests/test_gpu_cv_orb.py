"""GPU parity of the cv::ORB detector mode (MSLAM_HIP_DETECTOR_CV_ORB: drop-in for OrbOpenCvDetector,
orb_feature.cpp:25,33-65) against the oracle's restatement of OpenCV's ORB: stage by stage and end to end, bit-exact,
in the canonical (raster) order both sides use for the implementation-defined part of the reference's order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("xy", "desc", "octave", "angle", "response")


@pytest.fixture(scope="module")
def cctx(pkg):
    c = pkg.Context(width=640, height=480, max_batch=4, detector=pkg.DETECTOR_CV_ORB)
    yield c
    c.close()


def test_cv_orb_stages(pkg, orc, cctx, bundled_frames):
    f = bundled_frames[0]
    cctx.detect(f)
    p = orc.cvorb_params()
    w, h, s, q = orc.cvorb_geometry(640, 480, p)
    gw, gh, gs = cctx.level_geometry()
    assert (w, h) == (gw, gh) and np.array_equal(s, gs)
    pyr = orc.cvorb_pyramid(orc.gray(f), p)
    for l in range(8):
        assert np.array_equal(cctx.debug_image(pkg.DBG_PYRAMID, 0, l), pyr[l]), "INTER_LINEAR_EXACT level %d" % l
        assert np.array_equal(cctx.debug_image(pkg.DBG_BLURRED, 0, l), orc.gaussian_blur7(pyr[l])), "blur %d" % l
        kp = orc.fast(pyr[l], 20, cap=pyr[l].size // 4)                 # whole-level FAST + NMS, raster order
        kp = kp[(kp["x"] >= 31) & (kp["x"] < w[l] - 31) & (kp["y"] >= 31) & (kp["y"] < h[l] - 31)]
        got = cctx.debug_keypoints(pkg.DBG_CANDIDATES, 0, l)
        assert np.array_equal(got, np.stack([kp["x"], kp["y"], kp["response"]], 1)), "FAST level %d" % l
        sel = orc.cvorb_level_keypoints(pyr[l], p, q[l], 1)              # retainBest(2n) -> Harris -> retainBest(n)
        got = cctx.debug_keypoints(pkg.DBG_SELECTED, 0, l)
        assert np.array_equal(got, np.stack([sel["x"], sel["y"], sel["response"]], 1)), "selection level %d" % l


def test_cv_orb_keypoint_order_forms(pkg, orc, bundled_frames, synth_frames):
    """the two keypoint orders of the mode: libstdc++'s (default: what a GCC build of the reference returns, so its keypoint ids
    and match indices) and FAST's raster order — each against the oracle in the same order, the same set either way, and the
    switch takes effect on a context that has already captured its single-frame graph"""
    c = pkg.Context(width=640, height=480, max_batch=2, detector=pkg.DETECTOR_CV_ORB)
    for f in (bundled_frames[0], synth_frames[1]):
        lib = c.detect(f)
        ref = orc.cvorb_detect(f, orc.cvorb_params(order=orc.ORDER_LIBSTDCXX))
        for k in ("xy", "desc", "octave", "angle", "response"):
            assert np.array_equal(lib[k], ref[k]), ("library order", k)
        c.set_cv_keypoint_order(pkg.CV_ORDER_RASTER)
        ras = c.detect(f)
        ref = orc.cvorb_detect(f, orc.cvorb_params(order=orc.ORDER_RASTER))
        for k in ("xy", "desc", "octave", "angle", "response"):
            assert np.array_equal(ras[k], ref[k]), ("raster order", k)
        c.set_cv_keypoint_order(pkg.CV_ORDER_LIBSTDCXX)
        assert len(lib["xy"]) == len(ras["xy"]) and not np.array_equal(lib["xy"], ras["xy"])
        key = lambda d: sorted(zip(d["octave"].tolist(), map(tuple, d["xy"].tolist()), map(bytes, d["desc"])))  # noqa: E731
        assert key(lib) == key(ras)
    with pytest.raises(pkg.MslamHipError):
        c.set_cv_keypoint_order(7)
    c.close()


def test_cv_orb_detect_parity(orc, cctx, bundled_frames, synth_frames):
    p = orc.cvorb_params()
    for f in list(bundled_frames) + list(synth_frames[:3]):
        got, ref = cctx.detect(f), orc.cvorb_detect(f, p)
        assert len(ref["xy"]) >= 900 and len(got["xy"]) == len(ref["xy"])
        for k in KEYS:
            assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("W,H,n,levels,scale,thr,edge", [
    (320, 240, 500, 4, 1.2, 20, 31), (800, 600, 3000, 8, 1.2, 10, 31), (641, 479, 300, 3, 1.5, 25, 25),
    (1280, 720, 2000, 8, 1.2, 20, 31), (400, 300, 0, 2, 1.2, 20, 31),
    # scale factors whose quads outgrow the 12-byte source window of k_resize_col<true>: k_resize_exact, both forms
    (512, 384, 400, 3, 2.0, 20, 31), (600, 450, 300, 3, 2.7, 20, 31)])
def test_cv_orb_parameters(pkg, orc, W, H, n, levels, scale, thr, edge):
    import synth
    f = synth.make_stream(1, W, H, seed=W + n)[0]
    c = pkg.Context(width=W, height=H, detector=pkg.DETECTOR_CV_ORB, n_features=n, n_levels=levels, scale_factor=scale,
                    ini_fast_thr=thr, edge_threshold=edge, max_keypoints=16384, max_candidates=65536)
    got = c.detect(f, max_out=16384)
    ref = orc.cvorb_detect(f, orc.cvorb_params(n_features=n, n_levels=levels, scale_factor=scale, fast_threshold=thr,
                                               edge_threshold=edge))
    assert len(got["xy"]) == len(ref["xy"]) and (n == 0 or len(ref["xy"]) > 0.5 * n)
    for k in KEYS:
        assert np.array_equal(got[k], ref[k]), k
    c.close()


def test_cv_orb_flat_and_noise(pkg, orc, cctx):
    flat = np.full((480, 640, 3), 77, np.uint8)
    assert len(cctx.detect(flat)["xy"]) == 0
    noise = np.random.default_rng(8).integers(0, 256, (480, 640, 3), dtype=np.uint8)   # many equal FAST scores: ties
    c = pkg.Context(width=640, height=480, detector=pkg.DETECTOR_CV_ORB, max_candidates=65536)
    got, ref = c.detect(noise), orc.cvorb_detect(noise, orc.cvorb_params())
    assert len(ref["xy"]) >= 500
    for k in KEYS:
        assert np.array_equal(got[k], ref[k]), k
    c.close()


def test_cv_orb_batch_and_match(pkg, orc, synth_frames):
    """the batched device path in cv::ORB mode, matched frame to frame (both matcher kernels)"""
    import torch
    frames = synth_frames[:4]
    K = 2048
    c = pkg.Context(width=640, height=480, max_batch=4, max_keypoints=K, detector=pkg.DETECTOR_CV_ORB)
    refs = [orc.cvorb_detect(f, orc.cvorb_params()) for f in frames]
    for matcher in (pkg.MATCHER_AUTO, pkg.MATCHER_POPCOUNT):
        c.set_matcher(matcher)
        c.detect_batch_dev(torch.from_numpy(frames).cuda().data_ptr(), 4)
        c.match_batch_dev(0.7, False)
        c.sync()
        v = c.batch_view()
        cnt = pkg.read_device(c, v.count, (4,), np.int32)
        desc = pkg.read_device(c, v.desc, (4, K, 32), np.uint8)
        resp = pkg.read_device(c, v.response, (4, K), np.float32)
        mc = pkg.read_device(c, v.match_count, (4,), np.int32)
        mf = pkg.read_device(c, v.match_from, (4, K), np.int32)
        mt = pkg.read_device(c, v.match_to, (4, K), np.int32)
        for t in range(4):
            assert cnt[t] == len(refs[t]["xy"]) and np.array_equal(desc[t, :cnt[t]], refs[t]["desc"])
            assert np.array_equal(resp[t, :cnt[t]], refs[t]["response"])
            if t:
                rf, rt = orc.match(refs[t]["desc"], refs[t - 1]["desc"])
                assert len(rf) > 100 and mc[t] == len(rf)
                assert np.array_equal(mf[t, :mc[t]], rf) and np.array_equal(mt[t, :mc[t]], rt)
    c.close()


def test_cv_orb_capacity_is_loud(pkg, synth_frames):
    c = pkg.Context(width=640, height=480, detector=pkg.DETECTOR_CV_ORB, max_keypoints=300)
    with pytest.raises(pkg.MslamHipError) as e:
        c.detect(synth_frames[0])
    assert e.value.code == pkg.E_CAPACITY
    c.close()
    c = pkg.Context(width=640, height=480, detector=pkg.DETECTOR_CV_ORB, max_candidates=64)   # FAST list too small
    with pytest.raises(pkg.MslamHipError) as e:
        c.detect(synth_frames[0])
    assert e.value.code == pkg.E_CAPACITY and "candidates" in str(e.value)
    c.close()
