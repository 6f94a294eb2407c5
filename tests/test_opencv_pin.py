"""The route from "parity unpinned" to "pinned" (SURVEY.md §8c): outputs of the REAL third-party code the reference calls.

* OpenCV: `oracle/opencv_check/opencv_dump` (C++, needs OpenCV 4.8.1, which this image does not have) dumps every OpenCV
  primitive on the path for the reference's two bundled frames; `python oracle/opencv_check/compare.py --install dump.bin
  frame0.bgr frame1.bgr 640 480` checks it against the oracle and installs it as tests/golden/opencv/.  Once such a file is
  committed this test compares the oracle with it on every run.
* DBoW3: `MSLAM_ORB_VOCABULARY=/path/to/orbvoc.dbow3` (a vocabulary written by real DBoW3, QuickLZ-compressed or not: the
  file orb_relocalizer.cpp:28 opens) makes the loader tests run against it.

Neither exists in this repository yet: the tests then SKIP with a reason that says so.  A skip here is the statement
"parity unpinned", not a pass."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "opencv")
UNPINNED = ("PARITY UNPINNED: no %s — the oracle is a restatement of third-party code that is not under /root/reference and "
            "has never been compared with its real outputs (see this file's docstring for the one command that installs them)")


def test_oracle_against_real_opencv_dump(orc):
    dumps = sorted(glob.glob(os.path.join(GOLD, "opencv_dump_*.bin")))
    if not dumps:
        pytest.skip(UNPINNED % "tests/golden/opencv/opencv_dump_<W>x<H>.bin (written by oracle/opencv_check/opencv_dump)")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "opencv_check"))
    import compare
    for path in dumps:
        wh = os.path.basename(path)[len("opencv_dump_"):-4]
        W, H = (int(v) for v in wh.split("x"))
        frames = [np.fromfile(os.path.join(GOLD, "frame%d_%s.bgr" % (f, wh)), np.uint8).reshape(H, W, 3) for f in range(2)]
        lines = []
        bad = compare.compare(compare.load(path), frames, out=lines.append)
        assert bad == 0, "\n".join(l for l in lines if l.startswith("FAIL"))


def _vocabulary_path():
    p = os.environ.get("MSLAM_ORB_VOCABULARY", "")
    return p if p and os.path.isfile(p) else None


def _oracle_vocabulary(orc, blob):
    """the oracle reads plain streams: a QuickLZ-compressed file goes through the Python decoder of tools/quicklz.py first"""
    import quicklz
    return orc.Vocabulary(quicklz.decompress_vocabulary(blob))


def test_oracle_loads_a_real_dbow3_vocabulary(orc, bundled_frames):
    """the CPU restatement of Vocabulary::fromStream / transform on a file written by real DBoW3"""
    path = _vocabulary_path()
    if not path:
        pytest.skip(UNPINNED % "real DBoW3 vocabulary (set MSLAM_ORB_VOCABULARY to an orbvoc.dbow3 file)")
    blob = open(path, "rb").read()
    voc = _oracle_vocabulary(orc, blob)
    assert voc.k >= 2 and voc.L >= 1 and voc.n_words > 0
    d = orc.detect(bundled_frames[0], orc.params())["desc"]
    words, values = voc.bow_vector(d)
    assert len(words) > 0 and np.all(np.diff(words) > 0)            # a BowVector: ascending, unique word ids
    assert abs(float(np.sum(np.abs(values))) - 1.0) < 1e-9 or voc.scoring not in (0,)  # L1-normalised for L1 scoring


@pytest.mark.gpu
def test_hip_loads_a_real_dbow3_vocabulary(pkg, orc, bundled_frames):
    """mslam_hip_bow_load (QuickLZ decoder included) on the same file: words and values equal to the oracle's"""
    path = _vocabulary_path()
    if not path:
        pytest.skip(UNPINNED % "real DBoW3 vocabulary (set MSLAM_ORB_VOCABULARY to an orbvoc.dbow3 file)")
    blob = open(path, "rb").read()
    c = pkg.Context(width=640, height=480)
    c.bow_load(blob)
    d = orc.detect(bundled_frames[0], orc.params())["desc"]
    gw, gv = c.bow_transform(d)
    rw, rv = _oracle_vocabulary(orc, blob).bow_vector(d)
    assert np.array_equal(gw, rw) and np.array_equal(gv.view(np.uint64), rv.view(np.uint64))
    c.close()


def _self_dump(orc, frames):
    """A dump with the record names / layouts opencv_dump writes, filled from the oracle itself: exercises the comparison
    tool (so that it cannot rot while no real dump is committed) — it pins nothing."""
    dump = {}
    for f in range(2):
        F = "f%d_" % f
        bgr = np.ascontiguousarray(frames[f])
        gray = orc.gray(bgr)
        dump[F + "gray"] = gray
        pyr = orc.pyramid(gray, orc.params())
        cp = orc.cvorb_params()
        epyr = orc.cvorb_pyramid(gray, cp)
        for l in range(1, 8):
            dump[F + "linear_L%d" % l] = pyr[l]
            dump[F + "exact_L%d" % l] = epyr[l]

        def rows(k):
            return np.stack([k["x"], k["y"], k["response"]], 1).astype(np.float32).reshape(-1, 3)
        for l in (0, 3, 6):
            dump[F + "blur_L%d" % l] = orc.gaussian_blur7(pyr[l])
            dump[F + "fast20_L%d" % l] = rows(orc.fast(pyr[l], 20, cap=pyr[l].size // 4))
        for i in range(2):
            for j in range(9):
                for thr in (20, 7):
                    dump[F + "cell_%d_%d_t%d" % (i, j, thr)] = rows(
                        orc.fast(gray[19 + 64 * i:19 + 64 * i + 70, 19 + 64 * j:19 + 64 * j + 70], thr))
        d = orc.cvorb_detect(bgr, cp)
        dump[F + "orb_keypoints"] = np.stack([d["xy"][:, 0], d["xy"][:, 1], d["response"], d["angle"],
                                              d["octave"].astype(np.float32)], 1).astype(np.float32)
        dump[F + "orb_descriptors"] = d["desc"].copy()
    dump["fast_atan2"] = np.array([[orc.fast_atan2(float(y * 977), float(x * 1013)) for x in range(-40, 41)]
                                   for y in range(-40, 41)], np.float32)
    i0, i1, d0, d1 = orc.match_knn2_raw(dump["f1_orb_descriptors"], dump["f0_orb_descriptors"])
    dump["knn2"] = np.stack([i0, d0, i1, d1], 1)
    return dump


def test_comparison_tool_on_a_self_dump(orc, bundled_frames):
    """compare.py's own logic: a dump equal to the oracle passes with the row order recognised; the same keypoints in
    another STL's order are an INFO line by default and a failure under --strict-order; a wrong descriptor is a failure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "opencv_check"))
    import compare
    frames = [bundled_frames[0], bundled_frames[1]]
    dump = _self_dump(orc, frames)
    lines = []
    assert compare.compare(dump, frames, out=lines.append) == 0, "\n".join(lines)
    assert sum("orb keypoint order identical row order" in l and l.startswith("PASS") for l in lines) == 2

    other = dict(dump)
    kp, de = dump["f0_orb_keypoints"].copy(), dump["f0_orb_descriptors"].copy()
    lvl0 = np.flatnonzero(kp[:, 4] == 0)
    perm = np.arange(len(kp))
    perm[lvl0[:2]] = lvl0[1::-1]                                  # swap two keypoints of level 0
    other["f0_orb_keypoints"], other["f0_orb_descriptors"] = kp[perm], de[perm]
    ref = np.stack(orc.match_knn2_raw(dump["f1_orb_descriptors"], other["f0_orb_descriptors"]), 1)
    other["knn2"] = ref[:, [0, 2, 1, 3]]
    lines = []
    assert compare.compare(other, frames, out=lines.append) == 0, "\n".join(lines)
    assert any(l.startswith("INFO") and "f0_orb keypoint order" in l for l in lines)
    lines = []
    assert compare.compare(other, frames, out=lines.append, strict_order=True) == 1
    assert any(l.startswith("FAIL") and "f0_orb keypoint order" in l for l in lines)

    broken = dict(dump)
    broken["f1_orb_descriptors"] = dump["f1_orb_descriptors"].copy()
    broken["f1_orb_descriptors"][5] ^= 0xFF
    assert compare.compare(broken, frames, out=lambda *_: None) >= 1
