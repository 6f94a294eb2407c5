"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("xy", "desc", "octave", "angle", "response")


def assert_same_detection(got, ref):
    assert len(got["xy"]) == len(ref["xy"])
    for k in KEYS:
        assert np.array_equal(got[k], ref[k]), k


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(width=640, height=480, max_batch=4)
    yield c
    c.close()


def test_stages_bundled_frame(pkg, orc, ctx, bundled_frames):
    """pyramid, blur, FAST candidates and quadtree selection, level by level."""
    f = bundled_frames[0]
    ctx.detect(f)
    p = orc.params()
    pyr = orc.pyramid(orc.gray(f), p)
    w, h, s = orc.level_geometry(640, 480, p)
    gw, gh, gs = ctx.level_geometry()
    assert (w, h) == (gw, gh) and np.array_equal(s, gs)
    for l in range(8):
        assert np.array_equal(ctx.debug_image(pkg.DBG_PYRAMID, 0, l), pyr[l]), "pyramid level %d" % l
        assert np.array_equal(ctx.debug_image(pkg.DBG_BLURRED, 0, l), orc.gaussian_blur7(pyr[l])), "blur %d" % l
        cand = orc.fast_level(pyr[l], p)
        got = ctx.debug_keypoints(pkg.DBG_CANDIDATES, 0, l)
        ref = np.stack([cand["x"], cand["y"], cand["response"]], 1)
        assert np.array_equal(got, ref), "FAST candidates level %d" % l
        sel = orc.quadtree(cand, w[l], h[l], s[l], 1000)
        got = ctx.debug_keypoints(pkg.DBG_SELECTED, 0, l)
        ref = np.stack([sel["x"], sel["y"], sel["response"]], 1)
        assert np.array_equal(got, ref), "quadtree level %d" % l


def test_detect_bundled_frames(orc, ctx, bundled_frames):
    for f in bundled_frames:
        assert_same_detection(ctx.detect(f), orc.detect(f, orc.params()))


def test_detect_synthetic(orc, ctx, synth_frames):
    for f in synth_frames[:3]:
        assert_same_detection(ctx.detect(f), orc.detect(f, orc.params()))


@pytest.mark.parametrize("min_area,levels", [(150, 8), (300, 3), (4000, 5)])
def test_detect_parameters(pkg, orc, synth_frames, min_area, levels):
    c = pkg.Context(width=640, height=480, n_levels=levels, min_node_area=min_area, max_keypoints=16384)
    got = c.detect(synth_frames[0], max_out=16384)
    ref = orc.detect(synth_frames[0], orc.params(n_levels=levels, min_size=min_area))
    assert_same_detection(got, ref)
    c.close()


@pytest.mark.parametrize("size", [(100, 80), (333, 207), (1280, 720)])
def test_detect_other_sizes(pkg, orc, size):
    import synth
    W, H = size
    levels = 3 if W < 200 else 8
    f = synth.make_stream(1, W, H, seed=99)[0]
    c = pkg.Context(width=W, height=H, n_levels=levels, max_keypoints=20000, max_candidates=65536)
    got = c.detect(f, max_out=20000)
    ref = orc.detect(f, orc.params(n_levels=levels))
    assert_same_detection(got, ref)
    c.close()


def test_detect_flat_and_noise_frames(pkg, orc, ctx):
    flat = np.full((480, 640, 3), 128, np.uint8)
    got = ctx.detect(flat)
    assert len(got["xy"]) == 0 and len(orc.detect(flat, orc.params())["xy"]) == 0
    noise = np.random.default_rng(5).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    c = pkg.Context(width=640, height=480, max_keypoints=32768, max_candidates=65536)
    assert_same_detection(c.detect(noise, max_out=32768), orc.detect(noise, orc.params()))
    c.close()


def test_fast_dense_cells(pkg, orc):
    """k_fast_cells with nearly every pixel a candidate (noise at thresholds 3 / 1): the record segments of phase A fill up
    (64 groups per step), the expansion runs several rounds, and the short list of passing pixels overflows its 512 entries —
    the NMS phase then walks the full candidate list; stage by stage and end to end against the oracle.  Second frame:
    saturated black / white blocks (c + t and c - t leave the byte range: the biased-difference test has no saturation case)."""
    rng = np.random.default_rng(17)
    noise = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    blocks = np.repeat(np.repeat(rng.integers(0, 2, (30, 40), dtype=np.uint8) * 255, 8, 0), 8, 1)
    blocks = np.stack([blocks] * 3, -1)
    blocks[::7, ::5] ^= 255
    for frame, ini, mn in ((noise, 3, 1), (noise, 40, 2), (blocks, 20, 7), (blocks, 250, 200)):
        c = pkg.Context(width=320, height=240, n_levels=4, ini_fast_thr=ini, min_fast_thr=mn, min_node_area=50, max_keypoints=32768,
                        max_candidates=65536)
        p = orc.params(n_levels=4, ini_fast_thr=ini, min_fast_thr=mn, min_size=50)
        got = c.detect(frame, max_out=32768)
        ref = orc.detect(frame, p)
        assert_same_detection(got, ref)
        c.close()
    assert len(ref["xy"]) >= 0


def test_quadtree_storage_forms(pkg, orc):
    """k_quadtree / k_quadtree_big pick where the working arrays of a (level, frame) live from its candidate count N and the
    length n of its node list: N <= 2048 all in LDS; levels of images above 400 k pixels with N <= 12288: node arrays in
    LDS (k_quadtree_big), falling back to global memory when the list outgrows 4096 nodes; anything larger in global
    memory.  One 1280x720 frame per regime (texture / texture with a tiny stop area / noise), compared with the oracle."""
    import synth
    tex = synth.make_stream(1, 1280, 720, seed=321)[0]
    tex_hd = synth.make_stream(1, 1920, 1080, seed=322)[0]
    noise = np.random.default_rng(6).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    seen = set()
    for frame, levels, min_area in [(tex, 4, 1000), (tex_hd, 2, 40), (noise, 2, 300)]:
        H, W = frame.shape[:2]
        c = pkg.Context(width=W, height=H, n_levels=levels, min_node_area=min_area, max_keypoints=65535,
                        max_candidates=262144)
        got = c.detect(frame, max_out=65535)
        assert_same_detection(got, orc.detect(frame, orc.params(n_levels=levels, min_size=min_area)))
        N = c.debug_counts(pkg.DBG_CANDIDATES, 1)[0]
        n = c.debug_counts(pkg.DBG_SELECTED, 1)[0]
        for l in range(levels):
            big_level = round(W / 1.2 ** l) * round(H / 1.2 ** l) > 400000
            if N[l] <= 2048:
                seen.add("lds")
            elif big_level and N[l] <= 12288:
                seen.add("big" if n[l] <= 4096 else "big-overflow")
            else:
                seen.add("global")
        c.close()
    assert seen >= {"big", "big-overflow", "global"}, seen


@pytest.mark.parametrize("mirror", ["1", "0"])
def test_capacity_is_loud(pkg, orc, synth_frames, monkeypatch, mirror):
    """an overflow must reach the caller in BOTH result paths of the synchronous call: k_describe mirroring count + flags into
    the mapped result block itself (the default — block 0 / thread 0 writes a snapshot of the flags when it starts; it is the
    last flag writer of the single-frame sequence, csrc/k_describe.hip) and the round-3 packing kernel
    (MSLAM_HIP_MIRROR_RESULTS=0, read at context creation).  Keypoint capacity, candidate capacity, and a clean call after
    each failure on the same context."""
    monkeypatch.setenv("MSLAM_HIP_MIRROR_RESULTS", mirror)
    c = pkg.Context(width=640, height=480, max_keypoints=100)
    for _ in range(2):   # the flags are cleared by the failing call: the second one fails for its own overflow
        with pytest.raises(pkg.MslamHipError) as e:
            c.detect(synth_frames[0])
        assert e.value.code == pkg.E_CAPACITY
    c.close()
    c = pkg.Context(width=640, height=480, max_candidates=256)          # FAST candidate lists overflow (an earlier kernel's flag)
    with pytest.raises(pkg.MslamHipError) as e:
        c.detect(synth_frames[0])
    assert e.value.code == pkg.E_CAPACITY
    c.close()
    c = pkg.Context(width=640, height=480)
    got, ref = c.detect(synth_frames[0]), orc.detect(synth_frames[0], orc.params())
    assert len(got["xy"]) == len(ref["xy"]) and np.array_equal(got["desc"], ref["desc"])
    c.close()


@pytest.fixture(params=["matrix", "popcount"])
def mctx(request, pkg, ctx):
    """the shared context with the matcher form selected: the matrix-core kernel (default) or the xor/popcount
    kernel north_star names (mslam_hip_set_matcher) — identical results are required of both"""
    ctx.set_matcher(pkg.MATCHER_POPCOUNT if request.param == "popcount" else pkg.MATCHER_AUTO)
    assert ctx.get_matcher() == (1 if request.param == "popcount" else 0)
    yield ctx
    ctx.set_matcher(pkg.MATCHER_AUTO)


def test_match_knn2_random(orc, mctx):
    ctx = mctx
    rng = np.random.default_rng(1)
    # (33000, 70): beyond the matrix-core kernel's train range -> the xor/popcount kernel; (17000, 130): 532 tiles
    for n_from, n_to in [(2000, 2000), (513, 257), (1, 5), (2, 3), (300, 1), (32, 64), (33, 600), (17000, 130),
                         (33000, 70)]:
        f = rng.integers(0, 256, (n_from, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (n_to, 32), dtype=np.uint8)
        got = ctx.match_knn2(f, t)
        ref = orc.match_knn2_raw(f, t)
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)


def test_match_ties_and_ratio(orc, mctx):
    """few distinct descriptors => many equal distances: the lower train index must rank first."""
    ctx = mctx
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (7, 32), dtype=np.uint8)
    f = base[rng.integers(0, 7, 700)]
    t = base[rng.integers(0, 7, 300)].copy()
    t[::3, 0] ^= 1
    for g, r in zip(ctx.match_knn2(f, t), orc.match_knn2_raw(f, t)):
        assert np.array_equal(g, r)
    for ratio in (0.7, 0.5, 1.0, 0.0):
        gf, gt = ctx.match(f, t, ratio)
        rf, rt = orc.match(f, t, ratio)
        assert np.array_equal(gf, rf) and np.array_equal(gt, rt)


def test_match_detected_frames(orc, mctx, bundled_frames):
    ctx = mctx
    a = orc.detect(bundled_frames[0], orc.params())
    b = orc.detect(bundled_frames[1], orc.params())
    gf, gt = ctx.match(b["desc"], a["desc"])
    rf, rt = orc.match(b["desc"], a["desc"])
    assert len(rf) > 100
    assert np.array_equal(gf, rf) and np.array_equal(gt, rt)


def test_match_degenerate(mctx):
    ctx = mctx
    d = np.zeros((5, 32), np.uint8)
    assert len(ctx.match(d[:1], d)[0]) == 0      # n_from < 2: reference UB, defined as no matches
    assert len(ctx.match(d, d[:0])[0]) == 0      # empty query set
    assert len(ctx.match(d[:0], d)[0]) == 0


@pytest.mark.parametrize("matcher", [0, 1])
def test_batch_device_path(pkg, orc, synth_frames, matcher):
    """detect_batch_dev + match_batch_dev on HBM-resident frames, including the chain across batches."""
    import torch
    frames = synth_frames[:6]
    dev = torch.from_numpy(frames).cuda()
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=4096)
    c.set_matcher(matcher)
    refs = [orc.detect(f, orc.params()) for f in frames]
    K = 4096
    for b in range(2):
        c.detect_batch_dev(dev[3 * b:].data_ptr(), 3)
        c.match_batch_dev(0.7, True)
        c.sync()
        v = c.batch_view()
        assert v.n_frames == 3 and v.capacity == K

        def rd(ptr, shape, dt):
            return pkg.read_device(c, ptr, shape, dt)
        cnt = rd(v.count, (3,), np.int32)
        desc = rd(v.desc, (3, K, 32), np.uint8)
        xy = rd(v.xy, (3, K, 2), np.float32)
        mc = rd(v.match_count, (3,), np.int32)
        mf = rd(v.match_from, (3, K), np.int32)
        mt = rd(v.match_to, (3, K), np.int32)
        for i in range(3):
            r = refs[3 * b + i]
            assert cnt[i] == len(r["xy"])
            assert np.array_equal(desc[i, :cnt[i]], r["desc"]) and np.array_equal(xy[i, :cnt[i]], r["xy"])
            t = 3 * b + i
            if t == 0:
                assert mc[i] == 0
            else:
                rf, rt = orc.match(refs[t]["desc"], refs[t - 1]["desc"])
                assert mc[i] == len(rf)
                assert np.array_equal(mf[i, :mc[i]], rf) and np.array_equal(mt[i, :mc[i]], rt)
    c.close()


@pytest.mark.parametrize("kcap,skip_env,pipe", [(8192, None, "0"), (4096, "0", "0"), (8192, None, "1"), (4096, None, "0")])
def test_batch_matcher_tile_skip_path(pkg, orc, synth_frames, monkeypatch, kcap, skip_env, pipe):
    """the batched matrix-core loops on long train sets.  pipe = "1" (the default): the hand-scheduled loop
    k_match_knn2_fp4<4, false, true> takes every size.  pipe = "0" (MSLAM_HIP_MATCH_PIPE=0, the round-5 loops):
    k_match_knn2_fp4<4, SKIP = true> from MSLAM_HIP_MATCH_SKIP_FROM (6000) train rows on — a (train tile, query tile) block
    that cannot change any lane's top-2 skips its update, the owed ageing of the keys is applied later — and the compiler-
    scheduled loop below.  Dense keypoint sets (min-area 60: ~4 k per frame, 125 train tiles per query) through the batched
    path, against the oracle's matcher; skip_env = "0" forces the skip path onto the K = 4096 batch shape."""
    import torch
    monkeypatch.setenv("MSLAM_HIP_MATCH_PIPE", pipe)
    if skip_env is not None:
        monkeypatch.setenv("MSLAM_HIP_MATCH_SKIP_FROM", skip_env)
    area = 60 if kcap == 8192 else 1000
    frames = synth_frames[:6]
    dev = torch.from_numpy(frames).cuda()
    c = pkg.Context(width=640, height=480, max_batch=6, max_keypoints=kcap, min_node_area=area)
    c.detect_batch_dev(dev.data_ptr(), 6)
    c.match_batch_dev(0.7, False)
    c.sync()
    assert c.last_match_kernel() == "matrix"
    v = c.batch_view()
    K = v.capacity
    p = orc.params(min_size=area)
    refs = [orc.detect(f, p) for f in frames]
    cnt = pkg.read_device(c, v.count, (6,), np.int32)
    mc = pkg.read_device(c, v.match_count, (6,), np.int32)
    mf = pkg.read_device(c, v.match_from, (6, K), np.int32)
    mt = pkg.read_device(c, v.match_to, (6, K), np.int32)
    assert [int(x) for x in cnt] == [len(r["xy"]) for r in refs]
    if kcap == 8192:
        assert cnt.min() > 3500
    assert mc[0] == 0
    for t in range(1, 6):
        rf, rt = orc.match(refs[t]["desc"], refs[t - 1]["desc"])
        assert mc[t] == len(rf) and np.array_equal(mf[t, :mc[t]], rf) and np.array_equal(mt[t, :mc[t]], rt), t
    c.close()


def test_batch_matcher_auto_switch_on_capacity(pkg, orc, synth_frames):
    """MATCHER_AUTO on the batched path: a context whose max_keypoints exceeds the matrix-core kernel's train range
    (32 736 rows: the 14-bit age field of its sort key) must take the xor/popcount kernel (k_match.hip:
    launch_match_knn2 decides on the CAPACITY, cap_from, not on the actual counts) and still match the oracle."""
    import torch
    frames = synth_frames[:3]
    dev = torch.from_numpy(frames).cuda()
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=40000)
    assert c.get_matcher() == pkg.MATCHER_AUTO
    c.set_profiling(1)
    c.detect_batch_dev(dev.data_ptr(), 3)
    c.match_batch_dev(0.7, True)
    c.sync()
    c.set_profiling(0)
    v = c.batch_view()
    K = v.capacity
    assert K == 40000
    refs = [orc.detect(f, orc.params()) for f in frames]
    mc = pkg.read_device(c, v.match_count, (3,), np.int32)
    mf = pkg.read_device(c, v.match_from, (3, K), np.int32)
    mt = pkg.read_device(c, v.match_to, (3, K), np.int32)
    for t in (1, 2):
        rf, rt = orc.match(refs[t]["desc"], refs[t - 1]["desc"])
        assert mc[t] == len(rf) and np.array_equal(mf[t, :mc[t]], rf) and np.array_equal(mt[t, :mc[t]], rt)
    assert c.last_match_kernel() == "popcount"
    c.close()
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=4096)
    c.detect_batch_dev(dev.data_ptr(), 3)
    c.match_batch_dev(0.7, True)
    c.sync()
    assert c.last_match_kernel() == "matrix"
    c.close()


def test_backproject_rgbd(pkg, orc, ctx, bundled_frames, bundled_depth):
    """row f-1: depth lookup + pin-hole back-projection of the detected keypoints (TUM intrinsics)"""
    det = orc.detect(bundled_frames[0], orc.params())
    got_xyz, got_ok = ctx.backproject(bundled_depth[0], det["xy"])
    ref_xyz, ref_ok = orc.backproject(bundled_depth[0], det["xy"])
    assert np.array_equal(got_ok, ref_ok) and 0.5 < ref_ok.mean() < 1.0   # ~5 % of the depth pixels are 0
    assert np.array_equal(got_xyz, ref_xyz)                               # f64, bit-exact
    assert (ref_xyz[ref_ok, 2] > 0).all() and ref_xyz[ref_ok, 2].max() < 6.5


def test_backproject_batch_device(pkg, orc, synth_frames):
    import torch
    import synth
    depth = synth.make_depth(3, 640, 480)
    c = pkg.Context(width=640, height=480, max_batch=3, max_keypoints=4096)
    c.detect_batch_dev(torch.from_numpy(synth_frames[:3]).cuda().data_ptr(), 3)
    d_depth = torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).cuda()
    c.backproject_batch_dev(d_depth.data_ptr())
    c.sync()
    v, pv = c.batch_view(), c.points_view()
    cnt = pkg.read_device(c, v.count, (3,), np.int32)
    xyz = pkg.read_device(c, pv.xyz, (3, 4096, 3), np.float64)
    ok = pkg.read_device(c, pv.valid, (3, 4096), np.uint8)
    for i in range(3):
        det = orc.detect(synth_frames[i], orc.params())
        rx, ro = orc.backproject(depth[i], det["xy"])
        assert np.array_equal(ok[i, :cnt[i]].astype(bool), ro) and np.array_equal(xyz[i, :cnt[i]], rx)
    c.close()


def test_packed_batch_results(pkg, synth_frames):
    """mslam_hip_pack_batch_dev: exactly count[t] keypoint records and match_count[t] match records per frame, back to back,
    equal to the capacity-strided views frame by frame; a buffer that is too small is reported, not overrun"""
    import torch
    B, K = 5, 4096
    frames = synth_frames[:B]
    dev = torch.from_numpy(frames).cuda()
    depth = torch.from_numpy(np.full((B, 480, 640), 7000, np.uint16).view(np.int16)).cuda()
    c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K)
    c.detect_batch_dev(dev.data_ptr(), B)
    c.match_batch_dev(0.7, False)
    cap = c.packed_capacity(B)
    buf = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.MslamHipError):          # 3-D points asked for, but this batch has not been back-projected
        c.pack_batch_dev(buf.data_ptr(), cap, True)
    c.backproject_batch_dev(depth.data_ptr())
    c.pack_batch_dev(buf.data_ptr(), cap, True)
    c.sync()
    p = pkg.unpack_batch(buf.cpu().numpy())
    v, pv = c.batch_view(), c.points_view()
    cnt = pkg.read_device(c, v.count, (B,), np.int32)
    mc = pkg.read_device(c, v.match_count, (B,), np.int32)
    assert p["n_frames"] == B and p["bytes"] < cap // 2
    assert np.array_equal(np.diff(p["kp_offset"]), cnt) and np.array_equal(np.diff(p["match_offset"]), mc)
    full = {"xy": pkg.read_device(c, v.xy, (B, K, 2), np.float32), "desc": pkg.read_device(c, v.desc, (B, K, 32), np.uint8),
            "octave": pkg.read_device(c, v.octave, (B, K), np.int32), "angle": pkg.read_device(c, v.angle, (B, K), np.float32),
            "response": pkg.read_device(c, v.response, (B, K), np.float32), "xyz": pkg.read_device(c, pv.xyz, (B, K, 3), np.float64),
            "valid": pkg.read_device(c, pv.valid, (B, K), np.uint8)}
    mf = pkg.read_device(c, v.match_from, (B, K), np.int32)
    mt = pkg.read_device(c, v.match_to, (B, K), np.int32)
    for t in range(B):
        a, b = p["kp_offset"][t], p["kp_offset"][t + 1]
        for k, arr in full.items():
            assert np.array_equal(p[k][a:b].view(np.uint8), arr[t, :cnt[t]].view(np.uint8)), (t, k)
        a, b = p["match_offset"][t], p["match_offset"][t + 1]
        assert np.array_equal(p["match_from"][a:b], mf[t, :mc[t]]) and np.array_equal(p["match_to"][a:b], mt[t, :mc[t]])
    assert cnt.min() > 1000 and mc[1:].min() > 100
    # too small: nothing but the header is written, header.fits == 0, and the context reports it
    small = torch.full((4096,), 0xAB, dtype=torch.uint8, device="cuda")
    c.pack_batch_dev(small.data_ptr(), 4096, True)
    with pytest.raises(pkg.MslamHipError):
        c.sync()
    h = small.cpu().numpy()
    with pytest.raises(pkg.MslamHipError):
        pkg.unpack_batch(h)
    assert (h[1024:] == 0xAB).all()
    # a second detect batch that the matcher has not run on is packed with ZERO matches, not with the first batch's pairs
    c.detect_batch_dev(dev.data_ptr(), B)
    c.pack_batch_dev(buf.data_ptr(), cap, False)
    c.sync()
    p2 = pkg.unpack_batch(buf.cpu().numpy())
    assert np.array_equal(np.diff(p2["kp_offset"]), cnt) and not np.diff(p2["match_offset"]).any() and len(p2["match_from"]) == 0
    c.match_batch_dev(0.7, True)
    c.pack_batch_dev(buf.data_ptr(), cap, False)
    c.sync()
    p3 = pkg.unpack_batch(buf.cpu().numpy())
    assert np.diff(p3["match_offset"]).min() > 100     # chained: frame 0 has the previous batch's last frame as predecessor
    c.close()


def test_full_size_batch_properties(pkg, orc, synth_frames):
    """cfg2-sized launch (250 frames in one batch): size-independent properties instead of a 250-frame oracle run —
    identical frames give identical outputs wherever they sit in the batch, a frame matched against an identical
    predecessor maps every keypoint to itself with distance 0, and spot-checked frames equal the oracle."""
    import torch
    B, K = 250, 4096
    idx = np.arange(B) % 5                       # frames 0..4 repeated; frame t+5 == frame t
    idx[101] = idx[100]                          # frame 101 == frame 100: identical consecutive frames
    frames = np.ascontiguousarray(synth_frames[idx])
    c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K)
    c.detect_batch_dev(torch.from_numpy(frames).cuda().data_ptr(), B)
    c.match_batch_dev(0.7, False)
    c.sync()
    v = c.batch_view()
    cnt = pkg.read_device(c, v.count, (B,), np.int32)
    desc = pkg.read_device(c, v.desc, (B, K, 32), np.uint8)
    xy = pkg.read_device(c, v.xy, (B, K, 2), np.float32)
    mc = pkg.read_device(c, v.match_count, (B,), np.int32)
    mf = pkg.read_device(c, v.match_from, (B, K), np.int32)
    mt = pkg.read_device(c, v.match_to, (B, K), np.int32)
    for t in range(B):
        r = idx[t]
        assert cnt[t] == cnt[r] and np.array_equal(desc[t, :cnt[t]], desc[r, :cnt[r]])
        assert np.array_equal(xy[t, :cnt[t]], xy[r, :cnt[r]])
    for r in (0, 3):
        ref = orc.detect(synth_frames[r], orc.params())
        assert cnt[r] == len(ref["xy"]) and np.array_equal(desc[r, :cnt[r]], ref["desc"])
    # identical predecessor: every query whose descriptor is unique in the frame matches itself
    n = cnt[101]
    assert mc[101] > 0.9 * n
    assert np.array_equal(mf[101, :mc[101]], mt[101, :mc[101]])
    # a regular pair equals the oracle
    rf, rt = orc.match(desc[7, :cnt[7]], desc[6, :cnt[6]])
    assert mc[7] == len(rf) and np.array_equal(mf[7, :mc[7]], rf) and np.array_equal(mt[7, :mc[7]], rt)
    c.close()


@pytest.mark.parametrize("W,H,levels,scale,ini,mn,area", [
    (240, 320, 4, 1.2, 20, 7, 1000),     # portrait: the reference's `delta_x = max_x - min_y` branch (:1045-1052)
    (1000, 200, 3, 1.2, 20, 7, 500),     # very wide: five initial quadtree nodes
    (400, 300, 5, 1.5, 20, 7, 1000),     # other pyramid scale
    (512, 384, 3, 2.0, 30, 10, 2000),    # scale 2: widest resize windows, other FAST thresholds
    (402, 301, 6, 1.3, 12, 12, 300),     # odd sizes, equal thresholds
    (640, 480, 12, 1.1, 20, 7, 1000),    # many shallow levels
    (641, 479, 1, 1.2, 20, 7, 100),      # single level, W % 4 != 0
])
def test_detect_parameter_sweep(pkg, orc, W, H, levels, scale, ini, mn, area):
    import synth
    f = synth.make_stream(1, W, H, seed=W + H)[0]
    c = pkg.Context(width=W, height=H, n_levels=levels, scale_factor=scale, ini_fast_thr=ini, min_fast_thr=mn,
                    min_node_area=area, max_keypoints=30000, max_candidates=65536)
    got = c.detect(f, max_out=30000)
    ref = orc.detect(f, orc.params(n_levels=levels, scale_factor=scale, ini_fast_thr=ini, min_fast_thr=mn, min_size=area))
    assert len(ref["xy"]) > 20
    assert_same_detection(got, ref)
    c.close()


@pytest.mark.parametrize("fused,k6", [("0", "5"), ("1", "5"), ("3", "1"), ("16", "9"), ("16", "2")])
def test_pyramid_and_blur_forms(pkg, orc, bundled_frames, synth_frames, monkeypatch, fused, k6):
    """the fused level kernels (k_level.hip: gray + blur, resize + blur) for every / some / no level, with different
    row-block heights, against the stand-alone kernels' results = the oracle: planes, blurred planes, detection, and a
    batch that spans frame boundaries inside waves"""
    import torch
    import synth
    monkeypatch.setenv("MSLAM_HIP_FUSED_LEVELS", fused)
    monkeypatch.setenv("MSLAM_HIP_LEVEL_K6", k6)
    p = orc.params()
    for W, H, frame in ((640, 480, bundled_frames[1]), (332, 208, None), (1284, 724, None)):
        if frame is None:
            frame = synth.make_stream(1, W, H, seed=W)[0]
        c = pkg.Context(width=W, height=H, max_batch=5, max_keypoints=16384, max_candidates=65536)
        got, ref = c.detect(frame), orc.detect(frame, p)
        pyr = orc.pyramid(orc.gray(frame), p)
        for l in range(8):
            assert np.array_equal(c.debug_image(pkg.DBG_PYRAMID, 0, l), pyr[l]), "pyramid level %d" % l
            assert np.array_equal(c.debug_image(pkg.DBG_BLURRED, 0, l), orc.gaussian_blur7(pyr[l])), "blur %d" % l
        assert_same_detection(got, ref)
        if W == 640:
            batch = np.stack([synth_frames[i % 6] for i in range(5)])
            c.detect_batch_dev(torch.from_numpy(batch).cuda().data_ptr(), 5)
            c.sync()
            for f in (0, 4):
                pyr = orc.pyramid(orc.gray(batch[f]), p)
                for l in (0, 1, 7):
                    assert np.array_equal(c.debug_image(pkg.DBG_PYRAMID, f, l), pyr[l]), "frame %d level %d" % (f, l)
                    assert np.array_equal(c.debug_image(pkg.DBG_BLURRED, f, l), orc.gaussian_blur7(pyr[l])), "blur %d %d" % (f, l)
        c.close()


@pytest.mark.parametrize("frames,waves", [("1", "4"), ("2", "8"), ("3", "8"), ("16", "4")])
def test_level_chain_form(pkg, orc, bundled_frames, synth_frames, monkeypatch, frames, waves):
    """the one-launch level chain (k_level_chain, MSLAM_HIP_LEVEL_CHAIN=1: a workgroup of `waves` waves takes `frames` frames
    through gray + blur and every resize + blur level, a workgroup-scope fence + barrier between levels): every plane and
    blurred plane of the first, a middle and the last frame of an 11-frame batch (the last group is short, frame boundaries
    fall inside waves) and the detections, against the oracle; sizes with one and several column waves per frame"""
    import torch
    import synth
    monkeypatch.setenv("MSLAM_HIP_LEVEL_CHAIN", "1")
    monkeypatch.setenv("MSLAM_HIP_LEVEL_CHAIN_FRAMES", frames)
    monkeypatch.setenv("MSLAM_HIP_LEVEL_CHAIN_WAVES", waves)
    p = orc.params()
    for W, H in ((640, 480), (332, 208), (1284, 724)):
        n = 11 if W < 1000 else 8
        batch = synth.make_stream(n, W, H, seed=W + 7)
        if W == 640:
            batch[3] = bundled_frames[0]
        c = pkg.Context(width=W, height=H, max_batch=n, max_keypoints=16384, max_candidates=65536)
        c.set_profiling(2)
        c.detect_batch_dev(torch.from_numpy(batch).cuda().data_ptr(), n)
        c.sync()
        assert "levels" in {nm for nm, _ in c.stage_times()}, "the chain did not take this batch"
        c.set_profiling(0)
        v, K = c.batch_view(), 16384
        cnt = pkg.read_device(c, v.count, (n,), np.int32)
        desc = pkg.read_device(c, v.desc, (n, K, 32), np.uint8)
        xy = pkg.read_device(c, v.xy, (n, K, 2), np.float32)
        for f in sorted({0, 3, n // 2, n - 1}):
            pyr = orc.pyramid(orc.gray(batch[f]), p)
            for l in range(8):
                assert np.array_equal(c.debug_image(pkg.DBG_PYRAMID, f, l), pyr[l]), "frame %d level %d" % (f, l)
                assert np.array_equal(c.debug_image(pkg.DBG_BLURRED, f, l), orc.gaussian_blur7(pyr[l])), "blur %d %d" % (f, l)
            ref = orc.detect(batch[f], p)
            assert cnt[f] == len(ref["xy"])
            assert np.array_equal(desc[f, :cnt[f]], ref["desc"]) and np.array_equal(xy[f, :cnt[f]], ref["xy"])
        c.close()


@pytest.mark.parametrize("tiled", ["1", "0"])
def test_blurred_slab_tiled_and_in_rows(pkg, orc, bundled_frames, synth_frames, monkeypatch, tiled):
    """the blurred slab in 64 x 2 pixel tiles (the default when every level is fused; k_describe<true>) and in rows
    (MSLAM_HIP_TILED_BLUR=0; k_describe<false>): blurred planes as mslam_hip_debug_read returns them (it puts tiles back into
    rows), detections of single frames and of a batch, sizes whose row pitch is and is not a multiple of 64"""
    import torch
    import synth
    monkeypatch.setenv("MSLAM_HIP_TILED_BLUR", tiled)
    p = orc.params()
    for W, H, frame in ((640, 480, bundled_frames[0]), (404, 300, None), (1000, 270, None)):
        if frame is None:
            frame = synth.make_stream(1, W, H, seed=W + 1)[0]
        c = pkg.Context(width=W, height=H, max_batch=9, max_keypoints=16384, max_candidates=65536)
        got, ref = c.detect(frame), orc.detect(frame, p)
        pyr = orc.pyramid(orc.gray(frame), p)
        for l in range(8):
            assert np.array_equal(c.debug_image(pkg.DBG_BLURRED, 0, l), orc.gaussian_blur7(pyr[l])), "blur %d" % l
        assert_same_detection(got, ref)
        if W == 640:
            batch = np.stack([synth_frames[i % 6] for i in range(9)])
            c.detect_batch_dev(torch.from_numpy(batch).cuda().data_ptr(), 9)
            c.sync()
            for f in (0, 8):
                pyr = orc.pyramid(orc.gray(batch[f]), p)
                for l in (0, 3, 7):
                    assert np.array_equal(c.debug_image(pkg.DBG_BLURRED, f, l), orc.gaussian_blur7(pyr[l])), "blur %d %d" % (f, l)
            v, K = c.batch_view(), 16384
            cnt = pkg.read_device(c, v.count, (9,), np.int32)
            desc = pkg.read_device(c, v.desc, (9, K, 32), np.uint8)
            ang = pkg.read_device(c, v.angle, (9, K), np.float32)
            for f in (0, 5, 8):
                r = orc.detect(batch[f], p)
                assert cnt[f] == len(r["xy"])
                assert np.array_equal(desc[f, :cnt[f]], r["desc"]) and np.array_equal(ang[f, :cnt[f]], r["angle"])
        c.close()


def test_sparse_corners_use_fallback_threshold(pkg, orc):
    """cells without any threshold-20 corner must fall back to threshold 7 (:922-926): a dark frame with a few
    faint squares has corners only at the low threshold."""
    img = np.full((480, 640, 3), 60, np.uint8)
    rng = np.random.default_rng(11)
    for _ in range(40):
        x, y = int(rng.integers(40, 580)), int(rng.integers(40, 420))
        img[y:y + 12, x:x + 12] = 60 + int(rng.integers(9, 18))     # contrast 9..17: below 20, above 7
    img[200:230, 300:330] = 200                                       # one strong square: its cell stays at threshold 20
    c = pkg.Context(width=640, height=480)
    got, ref = c.detect(img), orc.detect(img, orc.params())
    assert len(ref["xy"]) > 10 and ref["response"].min() < 20 <= ref["response"].max()
    assert_same_detection(got, ref)
    c.close()


def test_stage_timing_modes_do_not_change_results(pkg, orc, synth_frames):
    """mode 1 serialises every stage on one stream (and takes the plain, non-graph single-frame path), mode 2 times
    every launch in place; both must leave the results untouched and report the stages by name"""
    import torch
    ref = orc.detect(synth_frames[0], orc.params())
    c = pkg.Context(width=640, height=480, max_batch=16)
    for mode in (1, 2, 0):
        c.set_profiling(mode)
        assert_same_detection(c.detect(synth_frames[0]), ref)
    dev = torch.from_numpy(np.stack([synth_frames[i % 6] for i in range(16)])).cuda()
    # (the blur has no launch of its own when every level is produced and blurred by the fused level kernels)
    expected = {"gray", "resize", "fast", "quadtree", "describe", "match_knn2", "ratio_compact"}
    if os.environ.get("MSLAM_HIP_LEVEL_CHAIN", "0") not in ("", "0"):  # gray + resize are one launch then (k_level_chain)
        expected = (expected - {"gray", "resize"}) | {"levels"}
    for mode in (1, 2):
        c.set_profiling(mode)
        c.detect_batch_dev(dev.data_ptr(), 16)   # 16 frames: two chunks on two streams in mode 2
        c.match_batch_dev(0.7, False)
        c.sync()
        times = c.stage_times()
        assert {n for n, _ in times} - {"blur"} == expected and all(ms > 0 for _, ms in times)
        v = c.batch_view()
        cnt = pkg.read_device(c, v.count, (16,), np.int32)
        desc = pkg.read_device(c, v.desc, (16, v.capacity, 32), np.uint8)
        assert cnt[0] == len(ref["xy"]) and np.array_equal(desc[0, :cnt[0]], ref["desc"])
        assert cnt[6] == cnt[0] and np.array_equal(desc[6, :cnt[6]], ref["desc"])
    c.set_profiling(0)
    c.close()


def test_cfg5_full_hd_three_levels(pkg, orc):
    """BASELINE cfg5: 1920x1080, 3 levels, ~15 k keypoints per frame, detect + ratio-test match of two frames
    (the quadtree of every level runs in k_quadtree_big, the matcher on 15 k x 15 k pairs)"""
    import synth
    frames = synth.make_stream(2, 1920, 1080, seed=4321)
    c = pkg.Context(width=1920, height=1080, n_levels=3, min_node_area=150, max_keypoints=32768, max_candidates=131072)
    p = orc.params(n_levels=3, min_size=150)
    dets, refs = [], []
    for f in frames:
        got = c.detect(f, max_out=32768)
        ref = orc.detect(f, p)
        assert_same_detection(got, ref)
        dets.append(got)
        refs.append(ref)
    assert len(refs[0]["xy"]) > 10000
    rf, rt = orc.match(refs[1]["desc"], refs[0]["desc"])
    for matcher in (pkg.MATCHER_AUTO, pkg.MATCHER_POPCOUNT):
        c.set_matcher(matcher)
        gf, gt = c.match(dets[1]["desc"], dets[0]["desc"])
        assert len(rf) > 1000 and np.array_equal(gf, rf) and np.array_equal(gt, rt)
    c.close()


def test_bench_launch_shape_1000_frames(pkg, orc, synth_frames):
    """the launch shape bench.py times: ONE 1000-frame batch (two 500-frame chunks on two streams, ~2 GB of pyramid
    slabs, 32-bit offsets at their largest), chained to a second batch.  Size-independent properties as in the
    250-frame test, plus oracle spot checks at both chunk boundaries and the ends."""
    import torch
    B, K = 1000, 4096
    idx = np.arange(B) % 6
    idx[[499, 500, 501]] = [2, 2, 5]             # identical consecutive frames across the chunk boundary (499 | 500)
    frames = torch.from_numpy(np.ascontiguousarray(synth_frames[idx])).cuda()
    c = pkg.Context(width=640, height=480, max_batch=B, max_keypoints=K)
    refs = [orc.detect(f, orc.params()) for f in synth_frames[:6]]
    for rep in range(2):
        c.detect_batch_dev(frames.data_ptr(), B)
        c.match_batch_dev(0.7, True)
        c.sync()
        v = c.batch_view()
        cnt = pkg.read_device(c, v.count, (B,), np.int32)
        desc = pkg.read_device(c, v.desc, (B, K, 32), np.uint8)
        xy = pkg.read_device(c, v.xy, (B, K, 2), np.float32)
        ang = pkg.read_device(c, v.angle, (B, K), np.float32)
        mc = pkg.read_device(c, v.match_count, (B,), np.int32)
        mf = pkg.read_device(c, v.match_from, (B, K), np.int32)
        mt = pkg.read_device(c, v.match_to, (B, K), np.int32)
        for r in range(6):                       # every frame equals the oracle's result for its source frame
            n = len(refs[r]["xy"])
            sel = np.nonzero(idx == r)[0]
            assert (cnt[sel] == n).all()
            assert (desc[sel, :n] == refs[r]["desc"][None]).all()
            assert (xy[sel, :n] == refs[r]["xy"][None]).all()
            assert (ang[sel, :n].view(np.uint32) == refs[r]["angle"].view(np.uint32)[None]).all()
        # matches: pair (t, t-1) only depends on (idx[t], idx[t-1]); check one representative of every kind
        # against the oracle and all others against their representative
        seen = {}
        for t in range(B):
            prev = idx[t - 1] if t > 0 else (idx[B - 1] if rep == 1 else None)
            if prev is None:
                assert mc[t] == 0
                continue
            key = (idx[t], prev)
            if key not in seen:
                rf, rt = orc.match(refs[key[0]]["desc"], refs[key[1]]["desc"])
                seen[key] = (rf, rt)
            rf, rt = seen[key]
            assert mc[t] == len(rf) and np.array_equal(mf[t, :mc[t]], rf) and np.array_equal(mt[t, :mc[t]], rt), t
        assert (2, 2) in seen and np.array_equal(*seen[(2, 2)])   # identical predecessor: every match maps i -> i
    c.close()


def test_soak_random_frames_both_detectors(pkg, orc):
    """60 frames of varied content (textures, contrast, noise levels, seeds, sizes) through both detector modes: every
    output array bit-identical to the oracle (the reference's in-tree extractor and its cv::ORB detector)"""
    import synth
    rng = np.random.default_rng(2024)
    for (W, H) in ((640, 480), (424, 240), (848, 480)):
        c0 = pkg.Context(width=W, height=H, max_keypoints=20000, max_candidates=65536)
        c1 = pkg.Context(width=W, height=H, max_keypoints=20000, max_candidates=65536, detector=pkg.DETECTOR_CV_ORB)
        base = synth.make_base(W, H, seed=int(rng.integers(1 << 30)))
        for i in range(20):
            f = synth.frame_from_base(base, int(rng.integers(0, 5000)), W, H, int(rng.integers(1 << 30))).astype(np.int16)
            gain, off, noise = rng.uniform(0.3, 1.6), rng.integers(-40, 40), rng.choice([0, 0, 3, 12, 40])
            f = f * gain + off + (rng.integers(-noise, noise + 1, f.shape) if noise else 0)
            if i % 7 == 3:                           # a flat band: cells without any corner, fallback threshold, empty levels
                f[H // 3:H // 2] = 90
            f = np.clip(f, 0, 255).astype(np.uint8)
            got, ref = c0.detect(f, max_out=20000), orc.detect(f, orc.params())
            assert_same_detection(got, ref)
            got, ref = c1.detect(f, max_out=20000), orc.cvorb_detect(f, orc.cvorb_params())
            assert_same_detection(got, ref)
        c0.close()
        c1.close()
