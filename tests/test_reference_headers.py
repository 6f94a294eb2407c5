"""The plugin adapter compiled (syntax only) against the reference's REAL interface headers — build container only.

`modular-slam_amd/host/mslam_hip_plugin.cpp` normally compiles against the repo's own mirror of the interfaces
(`host/mslam_interfaces.hpp`): Eigen / Boost / OpenCV are not in this image.  Here the same source is compiled with
`-DMSLAM_USE_REFERENCE_HEADERS -I/root/reference/src/lib/modular_slam/include`, so every `override` is checked against the
reference's own declarations (feature_interface.hpp:50-70, relocalizer.hpp:11-20, loop_detection.hpp:10-15, pnp.hpp:14-36)
and the factories against plugin_loader.hpp:13-25.  The third-party headers those files include are given as type-name
stand-ins in tests/stubs/ (no arithmetic, nothing linked or run; see tests/stubs/README.md).  Nothing of the reference is
copied into the repo; without /root/reference (the GPU box) the tests skip."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/src/lib/modular_slam/include"
STUBS = os.path.join(ROOT, "tests", "stubs")
HOST = os.path.join(ROOT, "modular-slam_amd", "host")
PLUGIN = os.path.join(HOST, "mslam_hip_plugin.cpp")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_INC) or shutil.which("g++") is None,
                                reason="needs the reference checkout under /root/reference and g++ (build container only)")

FLAGS = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror=suggest-override", "-Werror=overloaded-virtual",
         "-DMSLAM_USE_REFERENCE_HEADERS", "-I" + REF_INC, "-I" + STUBS]


def test_adapter_overrides_compile_against_the_reference_interfaces():
    r = subprocess.run(FLAGS + [PLUGIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    # and the interface declarations really came from the reference tree, the third-party names from tests/stubs
    deps = subprocess.run(["g++", "-std=c++17", "-M", "-DMSLAM_USE_REFERENCE_HEADERS", "-I" + REF_INC, "-I" + STUBS, PLUGIN],
                          capture_output=True, text=True, timeout=300).stdout
    for h in ("frontend/feature/feature_interface.hpp", "relocalizer.hpp", "loop_detection.hpp", "pnp.hpp", "orb_feature.hpp",
              "types/slam3d_types.hpp", "sensors/camera_parameters.hpp"):
        assert os.path.join(REF_INC, "modular_slam", h) in deps, h
    assert os.path.join(STUBS, "Eigen", "Dense") in deps and os.path.join(STUBS, "boost", "dynamic_bitset.hpp") in deps


def test_a_wrong_signature_is_caught():
    """the check has teeth: the same adapter with one argument type changed no longer overrides the reference's pure virtual"""
    src = open(PLUGIN).read()
    needle = "std::vector<OrbKeypoint> detect(const RgbFrame& sensorData) override"
    assert needle in src
    bad = src.replace(needle, "std::vector<OrbKeypoint> detect(const GrayScaleFrame& sensorData) override", 1)
    r = subprocess.run(FLAGS + ["-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-"], input=bad.replace(
        '"../../include/mslam_hip.h"', '"mslam_hip.h"'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "override" in r.stderr


def test_record_layout_and_loader_types_against_the_reference_headers(tmp_path):
    """sizeof(OrbKeypoint) == 64 with a 16-byte aligned Vector2 (what Eigen's fixed-size vectorizable Vector2d is), the
    descriptor at byte 32 (the matcher uses sizeof as the row stride, orb_feature.cpp:88-91), and the adapter's factories have
    the type the reference's loader imports (plugin_loader.hpp:13-25: std::unique_ptr<T>())"""
    tu = tmp_path / "layout.cpp"
    tu.write_text(textwrap.dedent("""
        #include "modular_slam/plugin_loader.hpp"
        #include "mslam_interfaces.hpp"
        #include <cstddef>
        #include <type_traits>
        namespace mslam
        {
        static_assert(alignof(Vector2) == 16 && sizeof(Vector2) == 16, "Vector2");
        static_assert(sizeof(Keypoint) == 32 && offsetof(Keypoint, coordinates) == 16, "Keypoint");
        static_assert(sizeof(OrbKeypoint) == 64 && offsetof(OrbKeypoint, descriptor) == 32, "OrbKeypoint");
        static_assert(sizeof(DescriptorMatch) == 16, "DescriptorMatch");
        std::unique_ptr<IOrbFeatureDetector> createHipOrbDetector();
        std::unique_ptr<IOrbMatcher> createHipOrbMatcher();
        std::unique_ptr<IOrbRelocalizer> createHipOrbRelocalizer();
        std::unique_ptr<IOrbLoopDetector> createHipLoopDetector();
        std::unique_ptr<ISlam3dPnp> createHipRansacPnp();
        static_assert(std::is_same<decltype(createHipOrbDetector), BlockFactoryCreator<IOrbFeatureDetector>>::value, "detector");
        static_assert(std::is_same<decltype(createHipOrbMatcher), BlockFactoryCreator<IOrbMatcher>>::value, "matcher");
        static_assert(std::is_same<decltype(createHipOrbRelocalizer), BlockFactoryCreator<IOrbRelocalizer>>::value, "reloc");
        static_assert(std::is_same<decltype(createHipLoopDetector), BlockFactoryCreator<IOrbLoopDetector>>::value, "loop");
        static_assert(std::is_same<decltype(createHipRansacPnp), BlockFactoryCreator<ISlam3dPnp>>::value, "pnp");
        // what the reference-side wiring of INTEGRATION.md instantiates
        inline void wiring()
        {
            BlockFactoryCreatorBoostFunction<IOrbFeatureDetector> d = loadFactoryMethod<IOrbFeatureDetector>("mslam_hip_plugin", "hipOrbDetectorFactory");
            BlockFactoryCreatorBoostFunction<IOrbMatcher> m = loadFactoryMethod<IOrbMatcher>("mslam_hip_plugin", "hipOrbMatcherFactory");
            BlockFactoryCreatorBoostFunction<IOrbLoopDetector> l = loadFactoryMethod<IOrbLoopDetector>("mslam_hip_plugin", "loopDetection");
            std::unique_ptr<IOrbFeatureDetector> det = d();
            std::unique_ptr<IOrbMatcher> mat = m();
            std::vector<DescriptorMatch> out = mat->match(det->detect(RgbFrame{}), det->detect(RgbFrame{}));
            (void)out;
            (void)l;
        }
        } // namespace mslam
    """))
    r = subprocess.run(FLAGS + ["-I" + HOST, str(tu)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
