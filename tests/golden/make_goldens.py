#!/usr/bin/env python3
"""(lives under tests/ because it drives the oracle, which only tests may do)
Write tests/golden/*.npz: the oracle's outputs for the reference's two bundled frames and two
seeded synthetic frames ("restatement goldens, OpenCV parity unverified" — SURVEY.md §8c).  They pin
the oracle against regressions and give the GPU tests fixed expected bytes."""
import os, sys, zlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft
import synth


def main():
    orc = graft.load_oracle()
    G = os.path.join(ROOT, "tests", "golden")
    frames = [np.frombuffer(zlib.decompress(open(os.path.join(G, "frame%04d_640x480.bgr.z" % i), "rb").read()),
                            np.uint8).reshape(480, 640, 3) for i in (0, 1)]
    syn = synth.make_stream(2, 640, 480, seed=1234)
    out = {}
    for name, fr in (("bundled", frames), ("synth1234", list(syn))):
        dets = [orc.detect(f, orc.params()) for f in fr]
        for i, d in enumerate(dets):
            for k, v in d.items():
                out["%s_%d_%s" % (name, i, k)] = v
        fi, ti = orc.match(dets[1]["desc"], dets[0]["desc"])
        out["%s_match_from" % name], out["%s_match_to" % name] = fi, ti
    blob = synth.make_vocabulary(10, 3)
    V = orc.Vocabulary(blob)
    w, v = V.bow_vector(out["bundled_0_desc"])
    out["bundled_0_bow_words"], out["bundled_0_bow_values"] = w, v
    w1, v1 = V.bow_vector(out["bundled_1_desc"])
    out["bundled_01_bow_score"] = np.array([orc.bow_score_l1(w, v, w1, v1)])
    np.savez_compressed(os.path.join(G, "oracle_goldens.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.endswith("xy") or "match" in k})


if __name__ == "__main__":
    main()
