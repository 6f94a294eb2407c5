#!/usr/bin/env python3
"""Convert the reference's bundled RGB-D frames (data/rgb|depth/000{0,1}.png — DATA files its tests
hold, test/CMakeLists.txt:23-24) into raw blobs under tests/golden/ so they can travel to the GPU
box (the reference itself cannot).  RGB is swapped to B,G,R byte order, which is what
cv::imread hands the reference (rgbd_file_provider.cpp:66); depth stays little-endian u16.
Run in the build container only (needs /root/reference and PIL)."""
import pathlib, zlib
import numpy as np
from PIL import Image

REF = pathlib.Path("/root/reference/data")
OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for i in (0, 1):
        rgb = np.array(Image.open(REF / "rgb" / ("%04d.png" % i)).convert("RGB"), np.uint8)
        assert rgb.shape == (480, 640, 3)
        bgr = np.ascontiguousarray(rgb[:, :, ::-1])
        (OUT / ("frame%04d_640x480.bgr.z" % i)).write_bytes(zlib.compress(bgr.tobytes(), 9))
        d = np.array(Image.open(REF / "depth" / ("%04d.png" % i))).astype("<u2")
        assert d.shape == (480, 640)
        (OUT / ("frame%04d_640x480.depth16.z" % i)).write_bytes(zlib.compress(d.tobytes(), 9))
        print(i, "bgr", bgr.mean(), "depth max", d.max(), "zeros %.3f" % (d == 0).mean())


if __name__ == "__main__":
    main()
