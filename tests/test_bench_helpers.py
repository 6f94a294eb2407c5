"""bench.py's profile plumbing and tools/summarize_counters.py on the CPU: the per-step counter summary the bench line's
`roofline.traffic` / `step_valu_issue` come from, the kernel -> stage map, and the staleness rule (a profile is only quoted for
the kernel sources it was collected on)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_kernel_names_map_to_stages():
    m = bench.stage_of_kernel
    assert m("mslam::k_gray_blur") == "gray" and m("mslam::k_gray4") == "gray"
    assert m("void mslam::k_resize_blur<false, 8>") == "resize" and m("void mslam::k_resize_col<true>") == "resize"
    assert m("mslam::k_blur2") == "blur" and m("mslam::k_fast_cells") == "fast" and m("mslam::k_zero_u32") == "fast"
    assert m("void mslam::k_match_knn2_fp4<4, false>") == "match_knn2" and m("void mslam::k_match_knn2<8, 1, 8>") == "match_knn2"
    assert m("mslam::k_describe") == "describe" and m("mslam::k_quadtree_big") == "quadtree"
    assert m("void mslam::k_describe<true>") == "describe" and m("void mslam::k_gray_blur<true>") == "gray"
    assert m("void mslam::k_resize_blur<false, 8, true>") == "resize" and m("void mslam::k_match_knn2_fp4<4, false>") == "match_knn2"
    assert m("__amd_rocclr_copyBuffer") is None
    assert m("void mslam::k_quadtree<1024, 512, 1>") == "quadtree" and m("void mslam::k_cv_select<4096, false>") == "select"


def test_every_kernel_of_the_committed_profile_has_a_stage():
    """the kernel names of the newest committed rocprofv3 summary (profiles/r*_kernel_stats.csv, the last by name) all map to a stage of the
    step — a renamed or templated kernel must not silently drop out of the per-step counter summary — and the kernel the bench
    line names for the dominant stage is one of them"""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_kernel_stats.csv")))
    assert files, "no committed kernel stats"
    names = [r["Name"].split("(")[0] for r in csv.DictReader(open(files[-1]))]
    ours = [n for n in names if "mslam::" in n]
    assert len(ours) >= 8
    for n in ours:
        assert bench.stage_of_kernel(n) is not None, n
    # the dominant-stage label of the default line: "void mslam::k_resize_blur<false, N, true> (one launch per level)"
    label = bench.STAGE_KERNEL["resize"].split(" (")[0]
    pat = re.escape(label).replace("N", r"\d+")
    assert any(re.fullmatch(pat, n) for n in ours), (label, ours)
    for st in ("gray", "fast", "describe"):
        assert bench.STAGE_KERNEL[st] in ours, (st, bench.STAGE_KERNEL[st])


def test_summarize_counters_sums_launches_per_step(tmp_path):
    """two levels of k_resize_blur and one k_gray_blur over 2 steps: per-step totals, per stage and per kernel"""
    rows = ["Kernel_Name,Counter_Name,Counter_Value"]
    for step in range(2):
        rows += ['"mslam::k_gray_blur(mslam::GrayBlurArgs)",FETCH_SIZE,100.0']
        rows += ['"void mslam::k_resize_blur<false, 8>(mslam::ResizeBlurArgs)",FETCH_SIZE,30.0']
        rows += ['"void mslam::k_resize_blur<false, 0>(mslam::ResizeBlurArgs)",FETCH_SIZE,10.0']
        rows += ['"__amd_rocclr_copyBuffer(void)",FETCH_SIZE,1.0']
    f = tmp_path / "f.csv"
    f.write_text("\n".join(rows) + "\n")
    w = tmp_path / "w.csv"
    w.write_text("Kernel_Name,Counter_Name,Counter_Value\n" + "\n".join(
        ['"mslam::k_gray_blur(mslam::GrayBlurArgs)",WRITE_SIZE,50.0'] * 2 + ['"mslam::k_gray_blur(mslam::GrayBlurArgs)",SQ_INSTS_VALU,1000000'] * 2) + "\n")
    out = tmp_path / "out.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_counters.py"), "2", str(out), str(f), str(w)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    j = json.loads(out.read_text())
    assert j["_meta"]["csrc_sha"] == bench.csrc_sha() and j["_meta"]["frames_per_step"] == 1000
    assert j["stages"]["gray"]["FETCH_SIZE"] == 100.0 and j["stages"]["gray"]["WRITE_SIZE"] == 50.0
    assert j["stages"]["resize"]["FETCH_SIZE"] == 40.0                     # both template instances, summed
    assert j["kernels"]["void mslam::k_resize_blur<false, 8>"]["launches_per_step"] == 1.0
    assert "copyBuffer" not in json.dumps(j["stages"])
    assert "memory-side MB per frame" in r.stdout and "vector-ALU issue time" in r.stdout


def test_profile_is_only_quoted_for_its_sources(tmp_path, monkeypatch):
    prof = {"_meta": {"csrc_sha": bench.csrc_sha(), "frames_per_step": 1000},
            "stages": {"gray": {"FETCH_SIZE": 500000.0, "WRITE_SIZE": 600000.0, "SQ_INSTS_VALU": 1.2e8},
                       "fast": {"SQ_INSTS_VALU": 4.8e8}}}
    p = tmp_path / "prof.json"
    p.write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "PMC_PROFILE", str(p))
    t, note = bench.pmc_traffic("gray", 1000)
    assert t == int((2 * 500000 + 600000) * 1024) and "FETCH_SIZE" in note
    assert bench.pmc_traffic("gray", 500)[0] == t // 2                      # linear in the frames per step
    assert bench.pmc_traffic("describe", 1000)[0] is None
    assert abs(bench.valu_issue_ms("fast", 1000) - 4.8e8 * 4 / (1024 * 2.4e9) * 1e3) < 1e-9
    assert abs(bench.valu_issue_ms(None, 1000) - 6.0e8 * 4 / (1024 * 2.4e9) * 1e3) < 1e-9
    prof["_meta"]["csrc_sha"] = "0" * 16                                     # collected on other sources: not quoted
    p.write_text(json.dumps(prof))
    t, note = bench.pmc_traffic("gray", 1000)
    assert t is None and note.startswith("stale")
    assert bench.valu_issue_ms(None, 1000) is None


def test_mempath_busy_fractions_and_their_gating(tmp_path, monkeypatch):
    """tools/summarize_mempath.py on two (X, GRBM_GUI_ACTIVE) passes: TA busy = TA_TA_BUSY_sum / 256 CUs and vector issue =
    SQ_INSTS_VALU x 4 / 1024 SIMDs, each over GRBM_GUI_ACTIVE / 8 XCDs of its own pass, summed over a stage's kernels and launches;
    bench.busy_fracs quotes them only for the kernel sources they were collected on; the level chain's kernel maps to its stage"""
    assert bench.stage_of_kernel("void mslam::k_level_chain<false, 8, true, 8, true>") == "levels"
    hdr = "Kernel_Name,Counter_Name,Counter_Value\n"
    ta = tmp_path / "ta.csv"
    va = tmp_path / "va.csv"
    rows_ta, rows_va = [], []
    for launch in range(3):  # describe: TA busy 256 * 600 over GRBM 8 * 1000 -> 0.6; two resize kernels: (256*100 + 256*300) / (8*400 + 8*600) -> 0.4
        rows_ta += ['"void mslam::k_describe<true>(x)",TA_TA_BUSY_sum,%d' % (256 * 600), '"void mslam::k_describe<true>(x)",GRBM_GUI_ACTIVE,%d' % (8 * 1000)]
        rows_ta += ['"void mslam::k_resize_blur<false, 8, true, 0>(x)",TA_TA_BUSY_sum,%d' % (256 * 100), '"void mslam::k_resize_blur<false, 8, true, 0>(x)",GRBM_GUI_ACTIVE,%d' % (8 * 400)]
        rows_ta += ['"void mslam::k_resize_blur<false, 0, true, 0>(x)",TA_TA_BUSY_sum,%d' % (256 * 300), '"void mslam::k_resize_blur<false, 0, true, 0>(x)",GRBM_GUI_ACTIVE,%d' % (8 * 600)]
        rows_va += ['"void mslam::k_describe<true>(x)",SQ_INSTS_VALU,%d' % (1024 * 125), '"void mslam::k_describe<true>(x)",GRBM_GUI_ACTIVE,%d' % (8 * 1000)]
    ta.write_text(hdr + "\n".join(rows_ta) + "\n")
    va.write_text(hdr + "\n".join(rows_va) + "\n")
    out = tmp_path / "mp.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_mempath.py"), str(out), str(ta), str(va)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    j = json.loads(out.read_text())
    d = j["kernels"]["mslam::k_describe<true>"]
    assert abs(d["ta_busy_frac"] - 0.6) < 1e-12 and abs(d["valu_issue_frac"] - 0.5) < 1e-12 and d["ta_busy_frac_sums"]["launches"] == 3
    monkeypatch.setattr(bench, "MEMPATH_PROFILE", str(out))
    t, v, note = bench.busy_fracs("describe")
    assert abs(t - 0.6) < 1e-12 and abs(v - 0.5) < 1e-12 and "GRBM_GUI_ACTIVE" in note
    t, v, _ = bench.busy_fracs("resize")
    assert abs(t - 0.4) < 1e-12 and v is None                            # both template instances, ratio of sums
    j["_meta"]["csrc_sha"] = "0123456789abcdef"
    out.write_text(json.dumps(j))
    t, v, note = bench.busy_fracs("describe")
    assert t is None and v is None and "stale" in note
    monkeypatch.setattr(bench, "MEMPATH_PROFILE", str(tmp_path / "missing.json"))
    assert bench.busy_fracs("describe")[0] is None
