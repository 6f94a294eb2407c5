"""The C-ABI library loads, exports every symbol include/mslam_hip.h declares, and — with no GPU —
fails loudly instead of falling back to any CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "mslam_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mslam_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_python_symbol_lists_agree(pkg):
    assert _declared() == sorted(pkg.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.mslam_hip_abi_version() == 5   # 5: mslam_hip_set_cv_keypoint_order; 2: mslam_hip_params gained detector / n_features / edge_threshold; 3: debug_counts takes its row count;
    # 4: + mslam_hip_pnp_set_confidence (the PnP entry points now end on the 0.99 confidence bound by default), mslam_hip_pack_batch_dev, mslam_hip_packed_capacity


def test_default_params_are_the_reference_operating_point(pkg):
    p = pkg.default_params()
    # distributed_cv_feature.cpp:1184-1186
    assert (p.n_levels, p.ini_fast_thr, p.min_fast_thr, p.min_node_area) == (8, 20, 7, 1000)
    assert abs(p.scale_factor - 1.2) < 1e-7 and (p.width, p.height) == (640, 480)
    # the in-tree detector by default; the cv::ORB mode's own defaults are orb_feature.cpp:25 / cv::ORB::create
    assert (p.detector, p.n_features, p.edge_threshold) == (pkg.DETECTOR_DISTRIBUTED, 1000, 31)


def test_invalid_parameters_are_rejected(pkg):
    for kw in (dict(width=0), dict(n_levels=17), dict(scale_factor=1.0), dict(min_fast_thr=30), dict(max_batch=0),
               dict(width=60, height=50), dict(width=5000, height=480), dict(detector=7),
               dict(detector=1, edge_threshold=5), dict(detector=1, width=20, height=20)):
        with pytest.raises(pkg.MslamHipError) as e:
            pkg.Context(**kw)
        assert e.value.code == pkg.E_INVALID, kw


def test_product_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.MslamHipError) as e:
        pkg.Context(width=640, height=480)
    assert e.value.code == pkg.E_RUNTIME and "no CPU fallback" in str(e.value)


def test_pattern_table_checksum():
    hdr = open(os.path.join(ROOT, "include", "mslam_orb_pattern.h")).read()
    fnv = int(re.search(r"MSLAM_ORB_PATTERN_FNV1A 0x([0-9A-F]+)u", hdr).group(1), 16)
    body = hdr[hdr.index("MSLAM_ORB_PATTERN_INIT {"):]
    vals = [int(v) for v in re.findall(r"-?\d+", body[body.index("{"):body.index("}")])]
    assert len(vals) == 1024 and all(-13 <= v <= 13 for v in vals)
    h = 0x811C9DC5
    for v in vals:
        h = ((h ^ (v & 0xFF)) * 0x01000193) & 0xFFFFFFFF
    assert h == fnv == 0x28710593
    ref = "/root/reference/src/lib/modular_slam/distributed_cv_feature.cpp"
    if os.path.exists(ref):  # build container only: the table equals the reference's numbers
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_pattern
        assert gen_pattern.parse_reference() == vals


def test_no_product_code_touches_the_oracle():
    """the shipped path must not import, link or execute anything under oracle/"""
    pkg_dir = os.path.join(ROOT, "modular-slam_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "mslam_oracle" not in txt and "oracle/" not in txt, os.path.join(dp, f)
