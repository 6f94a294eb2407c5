"""CPU tests of the oracle's primitives (the checker must itself be checked): known answers, closed
forms and brute-force numpy restatements.  The reference holds no golden vector for this path
(SURVEY.md §4), so these pins are what anchors the oracle — 'parity unpinned' vs OpenCV itself."""
import math

import numpy as np
import pytest


def test_gaussian_taps_and_umax(orc):
    assert orc.gaussian_taps() == [18, 34, 48, 56, 48, 34, 18]       # 8.8 fixed point, sums to 256
    assert orc.umax() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert 31 + 2 * sum(2 * u + 1 for u in orc.umax()[1:]) == 749              # pixels in the radius-15 disc
    # disc symmetric in u/v: column-wise and row-wise extents agree
    um = orc.umax()
    for u in range(16):
        assert max(v for v in range(16) if um[v] >= u) == um[u]


def test_gray_formula(orc):
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    g = orc.gray(bgr)
    f = np.float32
    exp = (f(0.299) * bgr[..., 0].astype(f) + f(0.587) * bgr[..., 1].astype(f)) + f(0.114) * bgr[..., 2].astype(f)
    assert np.array_equal(g, np.minimum(f(255), exp).astype(np.uint8))
    white = np.full((2, 2, 3), 255, np.uint8)
    assert (orc.gray(white) == 255).all() or (orc.gray(white) == 254).all()


def test_level_geometry_reference_sizes(orc):
    w, h, s = orc.level_geometry(640, 480, orc.params())
    assert w == [640, 533, 444, 370, 309, 257, 214, 179]
    assert h == [480, 400, 333, 278, 231, 193, 161, 134]
    assert sum(a * b for a, b in zip(w, h)) == 950532
    assert s[1] == np.float32(1.2) and s[2] == np.float32(1.2) * np.float32(1.2)


def test_resize_identity_and_constant(orc):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    assert np.array_equal(orc.resize_linear(img, 50, 40), img)      # scale 1: every weight is (2048, 0)
    flat = np.full((40, 50), 77, np.uint8)
    assert (orc.resize_linear(flat, 41, 33) == 77).all()
    ofs, coef = orc.resize_tables(640, 533)
    assert (coef.reshape(-1, 2).sum(1) == 2048).all() and ofs[0] == 0 and ofs[-1] <= 638


def test_resize_matches_float_bilinear(orc):
    """the fixed-point result stays within 1 grey level of real-valued bilinear interpolation"""
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    dw, dh = 53, 40
    out = orc.resize_linear(img, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (64 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (48 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 63); x1 = np.minimum(x0 + 1, 63); fx = np.clip(sx - x0, 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 47); y1 = np.minimum(y0 + 1, 47); fy = np.clip(sy - y0, 0, 1)
    I = img.astype(np.float64)
    ref = ((I[y0][:, x0] * (1 - fx) + I[y0][:, x1] * fx) * (1 - fy)[:, None] +
           (I[y1][:, x0] * (1 - fx) + I[y1][:, x1] * fx) * fy[:, None])
    assert np.abs(out - ref).max() <= 1.0


def test_blur_against_numpy(orc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (33, 45), dtype=np.uint8)
    taps = np.array(orc.gaussian_taps(), np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")           # numpy 'reflect' == BORDER_REFLECT_101
    hor = sum(taps[k] * pad[:, k:k + 45] for k in range(7))
    ver = sum(taps[k] * hor[k:k + 33, :] for k in range(7))
    assert np.array_equal(orc.gaussian_blur7(img), ((ver + 32768) >> 16).astype(np.uint8))
    assert (orc.gaussian_blur7(np.full((20, 20), 200, np.uint8)) == 200).all()


def test_fast_atan2_accuracy_and_quadrants(orc):
    assert orc.fast_atan2(0.0, 0.0) == 0.0
    assert orc.fast_atan2(0.0, 1.0) == 0.0
    for y, x in [(1, 1), (1, -1), (-1, -1), (-1, 1), (5, 0), (-5, 0), (0, -3), (123, -4567), (-7, 1000)]:
        ref = math.degrees(math.atan2(y, x)) % 360.0
        assert abs(orc.fast_atan2(y, x) - ref) < 0.02                # OpenCV documents ~0.3 deg


def test_util_cos_sin(orc):
    for deg in np.linspace(-720, 720, 2881):
        a = np.float32(deg * math.pi / 180.0)
        assert abs(orc.util_cos(a) - math.cos(a)) < 2e-3
        assert abs(orc.util_sin(a) - math.sin(a)) < 2e-3


def _fast_bruteforce(img, thr):
    """independent restatement: 9 contiguous of 16, score = max threshold keeping the corner, 3x3 strict NMS"""
    h, w = img.shape
    circ = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    I = img.astype(int)
    score = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = [I[y, x] - I[y + dy, x + dx] for dx, dy in circ]
            best = -1
            for s in range(16):
                arc = [d[(s + j) % 16] for j in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > thr:                                           # all 9 exceed thr
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if s > 0 and s > nb.max() and score[y, x] >= thr:
                out.append((x, y, s))
    return out


def test_fast_against_bruteforce(orc):
    rng = np.random.default_rng(4)
    img = (rng.integers(0, 2, (9, 9)).repeat(6, 0).repeat(6, 1) * 120 + rng.integers(0, 30, (54, 54))).astype(np.uint8)
    for thr in (20, 7):
        got = orc.fast(img, thr)
        exp = _fast_bruteforce(img, thr)
        assert len(exp) > 3
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == exp
    sub = img[5:40, 3:50]                                            # strided view, as the cell sub-images are
    got = orc.fast(sub, 20)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == _fast_bruteforce(np.ascontiguousarray(sub), 20)


def test_fast_level_cells_cover_without_duplicates(orc, synth_frames):
    g = orc.gray(synth_frames[0])
    cand = orc.fast_level(g, orc.params())
    assert len(cand) > 500
    xy = set(zip(cand["x"].tolist(), cand["y"].tolist()))
    assert len(xy) == len(cand)                                     # cells tile without duplicates (A.1)
    assert cand["x"].min() >= 3 and cand["x"].max() < 640 - 38 - 3
    # order: cell rows, then cell columns, then row-major inside the cell
    key = [(int(y) - 3) // 64 * 1000 + (int(x) - 3) // 64 for x, y in zip(cand["x"], cand["y"])]
    assert key == sorted(key)


def _quadtree_reference(cand, w, h, sf, min_size):
    """literal python restatement with a list: children inserted at the front, pass ends at the old tail"""
    W, H = w - 38, h - 38
    ratio = W / H
    nodes = []
    nx = round(ratio) if ratio > 1 else 1
    assert ratio > 1
    dx = W / nx
    for ix in range(nx):
        nodes.append(dict(b=(int(dx * ix), 0, int(dx * (ix + 1)), int(float(H))), k=[]))
    for i, c in enumerate(cand):
        nodes[int(c["x"] / dx)]["k"].append(i)
    nodes = [n for n in nodes if n["k"]]
    while True:
        prev = len(nodes)
        new_front, keep = [], []
        for n in nodes:
            bx, by, ex, ey = n["b"]
            if len(n["k"]) == 1 or np.float32(np.float32((ex - bx) * (ey - by)) * np.float32(sf)) * np.float32(sf) <= min_size:
                keep.append(n)
                continue
            hx, hy = -(-(ex - bx) // 2), -(-(ey - by) // 2)
            ch = [dict(b=(bx, by, bx + hx, by + hy), k=[]), dict(b=(bx + hx, by, ex, by + hy), k=[]),
                  dict(b=(bx, by + hy, bx + hx, ey), k=[]), dict(b=(bx + hx, by + hy, ex, ey), k=[])]
            for i in n["k"]:
                ch[(1 if bx + hx <= cand[i]["x"] else 0) + (2 if by + hy <= cand[i]["y"] else 0)]["k"].append(i)
            for c in ch:
                if c["k"]:
                    new_front.insert(0, c)
        nodes = new_front + keep
        if len(nodes) == prev:
            break
    out = []
    for n in nodes:
        best = n["k"][0]
        for i in n["k"][1:]:
            if cand[i]["response"] > cand[best]["response"]:
                best = i
        out.append(best)
    return out


@pytest.mark.parametrize("min_size", [1000, 200, 20000])
def test_quadtree_against_list_restatement(orc, synth_frames, min_size):
    g = orc.gray(synth_frames[1])
    p = orc.params()
    cand = orc.fast_level(g, p)
    sel = orc.quadtree(cand, 640, 480, 1.0, min_size)
    exp = _quadtree_reference(cand, 640, 480, 1.0, min_size)
    assert len(sel) == len(exp) > 10
    assert np.array_equal(sel, cand[exp])
    pyr = orc.pyramid(g, p)
    w, h, s = orc.level_geometry(640, 480, p)
    c3 = orc.fast_level(pyr[3], p)
    assert np.array_equal(orc.quadtree(c3, w[3], h[3], s[3], min_size), c3[_quadtree_reference(c3, w[3], h[3], s[3], min_size)])


def test_ic_angle_and_descriptor_properties(orc):
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 3, (64, 1))       # brighter to the right: centroid at +x
    assert orc.ic_angle(ramp, 32, 32) == 0.0
    assert abs(orc.ic_angle(np.ascontiguousarray(ramp.T), 32, 32) - 90.0) < 1e-3
    assert abs(orc.ic_angle(np.ascontiguousarray(ramp[:, ::-1]), 32, 32) - 180.0) < 1e-3
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    d0 = orc.orb_descriptor(img, 32, 32, 0.0)
    assert d0.shape == (32,) and 60 < np.unpackbits(d0).sum() < 200
    # rotating the image by 90 degrees and the angle by 90 gives the same descriptor (steered BRIEF)
    rot = np.ascontiguousarray(np.rot90(img, -1))                    # (x, y) -> (63 - y, x)
    assert np.array_equal(orc.orb_descriptor(rot, 31, 32, 90.0), d0)


def test_match_against_numpy(orc):
    rng = np.random.default_rng(6)
    f = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    t[:50] = f[100:150]
    t[:50, 0] ^= 3
    D = np.unpackbits(t[:, None, :] ^ f[None, :, :], axis=2).sum(2)
    order = np.argsort(D, axis=1, kind="stable")
    i0, i1, d0, d1 = orc.match_knn2_raw(f, t)
    assert np.array_equal(i0, order[:, 0]) and np.array_equal(i1, order[:, 1])
    assert np.array_equal(d0, D[np.arange(200), order[:, 0]]) and np.array_equal(d1, D[np.arange(200), order[:, 1]])
    fi, ti = orc.match(f, t, 0.7)
    keep = d0.astype(np.float64) < 0.7 * d1.astype(np.float64)
    assert np.array_equal(ti, np.nonzero(keep)[0]) and np.array_equal(fi, i0[keep])
    assert set(range(50)) <= set(ti.tolist())
    assert len(orc.match(f[:1], t)[0]) == 0


def test_bow_against_python(orc):
    import synth
    rng = np.random.default_rng(7)
    for weighting in (0, 1, 2, 3):
        blob = synth.make_vocabulary(4, 3, seed=5, weighting=weighting)
        V = orc.Vocabulary(blob)
        assert (V.k, V.L, V.n_nodes, V.n_words) == (4, 3, 85, 64)
        rec = np.dtype([("id", "<u4"), ("pid", "<u4"), ("w", "<f8"), ("c", "<i4"), ("r", "<i4"), ("t", "<i4"),
                        ("d", "u1", (32,))])
        nodes = np.frombuffer(blob, rec, 84, 13 + 16)
        by_id = {int(n["id"]): n for n in nodes}
        children = {}
        for n in nodes:
            children.setdefault(int(n["pid"]), []).append(int(n["id"]))
        d = rng.integers(0, 256, (200, 32), dtype=np.uint8)
        d[50:100] = d[:50]                                          # repeated words
        w, wt = V.words(d)
        acc = {}
        for r in range(200):
            cur = 0
            while cur in children:
                dist = [int(np.unpackbits(d[r] ^ by_id[c]["d"]).sum()) for c in children[cur]]
                cur = children[cur][int(np.argmin(dist))]
            assert w[r] == cur - 21 and wt[r] == by_id[cur]["w"]
            if weighting in (0, 1):
                acc[w[r]] = acc[w[r]] + wt[r] if w[r] in acc else wt[r]
            else:
                acc.setdefault(w[r], wt[r])
        keys = sorted(acc)
        norm = 0.0
        for k in keys:
            norm += abs(acc[k])
        bw, bv = V.bow_vector(d)
        assert list(bw) == keys and list(bv) == [acc[k] / norm for k in keys]
    assert orc.bow_score_l1(bw, bv, bw, bv) == pytest.approx(1.0, abs=1e-12)
    assert orc.bow_score_l1(bw[:0], bv[:0], bw, bv) == 0.0


def test_backproject_against_numpy(orc, bundled_depth):
    rng = np.random.default_rng(8)
    xy = np.stack([rng.uniform(19, 620, 500), rng.uniform(19, 460, 500)], 1).astype(np.float32)
    xyz, ok = orc.backproject(bundled_depth[0], xy)
    ix, iy = xy[:, 0].astype(np.float64).astype(int), xy[:, 1].astype(np.float64).astype(int)
    d = bundled_depth[0][iy, ix].astype(np.float32) * np.float32(1.0 / 5000.0)
    assert np.array_equal(ok, d > np.finfo(np.float32).eps)
    z = d.astype(np.float64)
    ex = (xy[:, 0].astype(np.float64) - 319.5) * z * (1.0 / 525.0)
    ey = (xy[:, 1].astype(np.float64) - 239.5) * z * (1.0 / 525.0)
    assert np.array_equal(xyz[ok, 0], ex[ok]) and np.array_equal(xyz[ok, 1], ey[ok]) and np.array_equal(xyz[ok, 2], z[ok])
    assert (xyz[~ok] == 0).all() and 0.01 < (~ok).mean() < 0.10        # the bundled depth has ~5 % holes
