"""Seeded synthetic RGB-D streams: value-noise texture + filled rectangles/discs that
create corners, translated per frame, with per-frame additive noise.  numpy only; identical bytes
feed the GPU path and the CPU oracle."""
import struct

import numpy as np


def _value_noise(h, w, cell, rng):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(int)
    x0 = xs.astype(int)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_base(width, height, seed=1234, n_shapes=None, margin=(64, 48)):
    """Base BGR texture of size (height+margin_y, width+margin_x)."""
    H, W = height + margin[1], width + margin[0]
    rng = np.random.default_rng(seed)
    tex = np.zeros((H, W), np.float32)
    for cell, amp in ((64, 0.5), (16, 0.3), (4, 0.2)):
        tex += amp * _value_noise(H, W, cell, rng)
    img = np.repeat((40 + 150 * tex)[:, :, None], 3, axis=2)
    img *= np.array([0.9, 1.0, 1.1], np.float32)
    rng2 = np.random.default_rng(seed + 1)
    n_shapes = n_shapes if n_shapes is not None else max(50, int(400 * (W * H) / (704 * 528)))
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_shapes):
        cx, cy = int(rng2.integers(0, W)), int(rng2.integers(0, H))
        sz = int(rng2.integers(4, 28))
        col = rng2.integers(0, 256, 3).astype(np.float32)
        if rng2.random() < 0.6:
            x0, y0 = max(cx - sz, 0), max(cy - sz, 0)
            img[y0:cy + sz, x0:cx + sz] = col
        else:
            y0, y1, x0, x1 = max(cy - sz, 0), min(cy + sz + 1, H), max(cx - sz, 0), min(cx + sz + 1, W)
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= sz * sz
            img[y0:y1, x0:x1][m] = col
    return np.clip(img, 0, 255).astype(np.uint8)


def frame_from_base(base, t, width, height, seed):
    """Frame t: the base translated by (2t mod 64, t mod 48) px plus U{-2..2} noise (seed+766+t = 2000+t for 1234)."""
    mx, my = base.shape[1] - width, base.shape[0] - height
    ox, oy = (2 * t) % max(mx, 1), t % max(my, 1)
    crop = base[oy:oy + height, ox:ox + width].astype(np.int16)
    noise = np.random.default_rng(seed + 766 + t).integers(-2, 3, crop.shape, dtype=np.int16)
    return np.clip(crop + noise, 0, 255).astype(np.uint8)


def make_stream(n_frames, width=640, height=480, seed=1234, t0=0):
    """[n_frames, height, width, 3] uint8 BGR frames."""
    base = make_base(width, height, seed)
    out = np.empty((n_frames, height, width, 3), np.uint8)
    for i in range(n_frames):
        out[i] = frame_from_base(base, t0 + i, width, height, seed)
    return out


def make_depth(n_frames, width=640, height=480, seed=1234):
    """u16 depth = 5000*(1 + 0.5*valuenoise) with 5 % zeros (TUM scale 1/5000)."""
    rng = np.random.default_rng(seed + 7)
    d = (5000 * (1 + 0.5 * _value_noise(height, width, 32, rng))).astype(np.uint16)
    out = np.repeat(d[None], n_frames, 0)
    zero = np.random.default_rng(seed + 8).random(out.shape) < 0.05
    out[zero] = 0
    return out


def make_vocabulary(k=10, L=3, seed=77, weighting=0, scoring=0):
    """A synthetic DBoW3 binary vocabulary stream (uncompressed Vocabulary::toStream layout,
    dbow3.patch:2252-2355): complete k-ary tree of depth L, iid uniform 256-bit node descriptors,
    idf-like positive leaf weights from seeded counts.  Returns bytes."""
    rng = np.random.default_rng(seed)
    n_nodes = sum(k ** l for l in range(L + 1))
    desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    n_words = k ** L
    counts = rng.integers(1, 1000, n_words)
    idf = np.maximum(np.log(2000.0 / counts), 0.01).astype(np.float64)
    first_leaf = n_nodes - n_words
    rec = np.dtype([("id", "<u4"), ("pid", "<u4"), ("w", "<f8"), ("cols", "<i4"), ("rows", "<i4"), ("type", "<i4"),
                    ("d", "u1", (32,))])
    assert rec.itemsize == 60
    # toStream pops parents off a stack and writes each parent's children in order; node ids use the
    # heap numbering of a complete k-ary tree (children of p: p*k+1 .. p*k+k)
    order = []
    stack = [0]
    while stack:
        pid = stack.pop()
        for c in range(k):
            cid = pid * k + 1 + c
            order.append((cid, pid))
            if cid < first_leaf:
                stack.append(cid)
    order = np.array(order, np.int64)
    recs = np.zeros(n_nodes - 1, rec)
    recs["id"] = order[:, 0]
    recs["pid"] = order[:, 1]
    leaf = order[:, 0] >= first_leaf
    recs["w"][leaf] = idf[order[leaf, 0] - first_leaf]
    recs["cols"], recs["rows"], recs["type"] = 32, 1, 0
    recs["d"] = desc[order[:, 0]]
    w = np.zeros(n_words, np.dtype([("wid", "<u4"), ("nid", "<u4")]))
    w["wid"] = np.arange(n_words)
    w["nid"] = np.arange(n_words) + first_leaf
    return b"".join([struct.pack("<QBI", 88877711233, 0, n_nodes), struct.pack("<iiii", k, L, scoring, weighting),
                     recs.tobytes(), struct.pack("<I", n_words), w.tobytes()])
