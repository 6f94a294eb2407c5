"""Oracle vs the committed restatement goldens (CPU), and the HIP path vs the same goldens (GPU)."""
import os

import numpy as np
import pytest

import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_goldens.npz"))
KEYS = ("xy", "desc", "octave", "angle", "response")


def _frames(name, bundled_frames):
    return bundled_frames if name == "bundled" else list(synth.make_stream(2, 640, 480, seed=1234))


@pytest.mark.parametrize("name", ["bundled", "synth1234"])
def test_oracle_matches_goldens(orc, bundled_frames, name):
    fr = _frames(name, bundled_frames)
    dets = [orc.detect(f, orc.params()) for f in fr]
    for i, d in enumerate(dets):
        for k in KEYS:
            assert np.array_equal(d[k], G["%s_%d_%s" % (name, i, k)]), (name, i, k)
    fi, ti = orc.match(dets[1]["desc"], dets[0]["desc"])
    assert np.array_equal(fi, G[name + "_match_from"]) and np.array_equal(ti, G[name + "_match_to"])


def test_oracle_bow_goldens(orc):
    V = orc.Vocabulary(synth.make_vocabulary(10, 3))
    w, v = V.bow_vector(G["bundled_0_desc"])
    assert np.array_equal(w, G["bundled_0_bow_words"]) and np.array_equal(v, G["bundled_0_bow_values"])
    w1, v1 = V.bow_vector(G["bundled_1_desc"])
    assert orc.bow_score_l1(w, v, w1, v1) == G["bundled_01_bow_score"][0]


def test_bundled_frames_sanity():
    """the goldens describe a plausible ORB result for consecutive frames (plumbing config 1)"""
    xy0, xy1 = G["bundled_0_xy"], G["bundled_1_xy"]
    assert 1000 < len(xy0) < 2000 and xy0.min() >= 19 and xy0[:, 0].max() <= 640 - 19
    d = np.hypot(*(xy1[G["bundled_match_from"]] - xy0[G["bundled_match_to"]]).T)
    assert np.median(d) < 3.0                                         # consecutive frames: small motion


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bundled", "synth1234"])
def test_hip_matches_goldens(pkg, bundled_frames, name):
    fr = _frames(name, bundled_frames)
    c = pkg.Context(width=640, height=480)
    dets = [c.detect(f) for f in fr]
    for i, d in enumerate(dets):
        for k in KEYS:
            assert np.array_equal(d[k], G["%s_%d_%s" % (name, i, k)]), (name, i, k)
    fi, ti = c.match(dets[1]["desc"], dets[0]["desc"])
    assert np.array_equal(fi, G[name + "_match_from"]) and np.array_equal(ti, G[name + "_match_to"])
    c.bow_load(synth.make_vocabulary(10, 3))
    w, v = c.bow_transform(G["bundled_0_desc"])
    assert np.array_equal(w, G["bundled_0_bow_words"]) and np.array_equal(v, G["bundled_0_bow_values"])
    c.close()
