"""oracle/mslam_cpu_bench.c (the timed CPU leg of bench.py): the pthread harness must do exactly the work of the
per-frame oracle calls — same keypoint and match totals — for every sharding of the stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import mslam_oracle as orc  # noqa: E402
import synth  # noqa: E402


def _totals(frames, p, first, count, detect):
    """detect every frame of the block, match each against its predecessor (from = current, to = previous)"""
    dets = [detect(frames[(first + i) % len(frames)], p) for i in range(count)]
    kp = sum(len(d["xy"]) for d in dets)
    m = sum(len(orc.match(dets[i]["desc"], dets[i - 1]["desc"])[0]) for i in range(1, count))
    return kp, m


def test_harness_counts_equal_per_frame_calls():
    frames = synth.make_stream(5, 320, 240, seed=1234)
    p = orc.params(min_size=400)
    for threads, per_thread in ((1, 4), (3, 3), (2, 7)):
        r = orc.bench_stream(frames, p, threads, per_thread)
        kp = m = 0
        for t in range(threads):
            a, b = _totals(frames, p, t * per_thread, per_thread, orc.detect)
            kp += a
            m += b
        assert (r["keypoints"], r["matches"]) == (kp, m), (threads, per_thread)
        assert r["frames"] == threads * per_thread and r["seconds"] > 0
        assert r["thread_seconds_min"] <= r["thread_seconds_max"] <= r["seconds"] + 0.5


def test_harness_cv_orb_mode_and_overflow():
    frames = synth.make_stream(2, 320, 240, seed=1234)
    cvp = orc.cvorb_params(n_features=300)
    r = orc.bench_stream(frames, None, 2, 2, cv_params=cvp)
    kp, m = _totals(frames, cvp, 0, 2, orc.cvorb_detect)
    kp2, m2 = _totals(frames, cvp, 2, 2, orc.cvorb_detect)
    assert (r["keypoints"], r["matches"]) == (kp + kp2, m + m2)
    try:
        orc.bench_stream(frames, orc.params(min_size=400), 1, 1, max_kp=10)
    except RuntimeError as e:
        assert "exceeded" in str(e)
    else:
        raise AssertionError("capacity overflow not reported")


def test_cpu_baseline_leg_reports_cores_and_efficiency():
    import bench
    frames = synth.make_stream(4, 320, 240, seed=1234)
    r = bench.cpu_baseline(frames, dict(n_levels=8, min_size=400), 0.5)
    for k in ("value", "value_1core", "cores", "scaling_efficiency", "affinity_cpus", "frames_per_thread", "sample"):
        assert k in r, k
    assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0 and r["frames_per_thread"] >= 8
    assert abs(r["scaling_efficiency"] - r["value"] / (r["cores"] * r["value_1core"])) < 1e-9
