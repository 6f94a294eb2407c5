import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the HIP library, the host plugin and the oracle when a fresh checkout has none of them yet
    (hipcc cross-compiles gfx950 without a GPU).  A prebuilt tree is left alone."""
    need = [os.path.join(ROOT, "modular-slam_amd", "libmslam_hip.so"),
            os.path.join(ROOT, "modular-slam_amd", "host", "libmslam_hip_plugin.so"),
            os.path.join(ROOT, "oracle", "libmslam_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        graft.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    o = graft.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


def _golden(name):
    return os.path.join(ROOT, "tests", "golden", name)


@pytest.fixture(scope="session")
def bundled_frames():
    """The reference's two bundled 640x480 frames (data/rgb/000{0,1}.png) as BGR bytes."""
    out = []
    for i in (0, 1):
        raw = zlib.decompress(open(_golden("frame%04d_640x480.bgr.z" % i), "rb").read())
        out.append(np.frombuffer(raw, np.uint8).reshape(480, 640, 3))
    return out


@pytest.fixture(scope="session")
def bundled_depth():
    out = []
    for i in (0, 1):
        raw = zlib.decompress(open(_golden("frame%04d_640x480.depth16.z" % i), "rb").read())
        out.append(np.frombuffer(raw, "<u2").reshape(480, 640))
    return out


@pytest.fixture(scope="session")
def synth_frames():
    import synth
    return synth.make_stream(6, 640, 480, seed=1234)
