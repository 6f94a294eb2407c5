"""QuickLZ packets (what DBoW3 writes for a vocabulary saved with compressed = true, dbow3.patch:2325-2349):
the library's decoder (csrc/quicklz_decode.hip, host-only entry point mslam_hip_qlz_decompress) against the original
bytes and against an independent Python decoder, on packets written by tools/quicklz.py's level-1 and level-3
encoders.  QuickLZ itself does not exist in this image, so the format is a restatement (unverified against real
QuickLZ output) — what these tests pin is that the two decoders agree and that every token kind is exercised."""
import struct

import numpy as np
import pytest

import quicklz
import synth


def _cases():
    rng = np.random.default_rng(4)
    text = (b"the quick brown fox jumps over the lazy dog " * 400)[:15000]
    runs = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(200))
    mixed = bytearray(rng.integers(0, 256, 30000, dtype=np.uint8).tobytes())
    for _ in range(300):                                    # copies of earlier material at assorted distances / lengths
        a, n = int(rng.integers(0, 25000)), int(rng.integers(3, 300))
        b = int(rng.integers(0, 29000 - n))
        mixed[b:b + n] = mixed[a:a + n]
    voc = synth.make_vocabulary(10, 3, seed=3)[13:]
    return {"text": text, "runs": runs, "mixed": bytes(mixed), "random": rng.integers(0, 256, 12345, dtype=np.uint8).tobytes(),
            "vocabulary": voc, "tiny": b"abcabcabcabcabcabcabcabc", "eleven": b"0123456789A"}


@pytest.mark.parametrize("level", [1, 3])
@pytest.mark.parametrize("name", list(_cases()))
def test_round_trip(pkg, level, name):
    data = _cases()[name]
    n, packed = quicklz.compress_stream(data, level)
    assert n == -(-len(data) // 10000)
    # independent Python decoder
    out, pos = bytearray(), 0
    for _ in range(n):
        o, used = quicklz.decompress_packet(packed[pos:])
        out += o
        pos += used
    assert pos == len(packed) and bytes(out) == data
    # the library's decoder
    assert pkg.qlz_decompress(packed, n) == data
    if name in ("text", "runs", "mixed", "vocabulary"):
        assert len(packed) < 0.9 * len(data)                 # matches were really emitted
    if name == "random":
        assert packed[0] & 1 == 0                            # incompressible: stored packet


def test_short_header_and_errors(pkg):
    # a stored packet with the 3-byte header QuickLZ uses below 216 bytes: flags (bit 1 clear), csize, dsize as bytes
    data = b"short header packet"
    pkt = bytes([(1 << 6) | (1 << 2), 3 + len(data), len(data)]) + data
    assert pkg.qlz_decompress(pkt, 1) == data
    assert quicklz.decompress_packet(pkt) == (data, len(pkt))
    n, packed = quicklz.compress_stream(b"abcd" * 5000, 1)
    for bad in (packed[:40], packed[:9] + b"\xff" * 40, b"", bytes([0x47]) + struct.pack("<II", 20, 1 << 30) + b"x" * 11):
        with pytest.raises(pkg.MslamHipError) as e:
            pkg.qlz_decompress(bad, n, capacity=1 << 20)
        assert e.value.code in (pkg.E_FORMAT, pkg.E_CAPACITY)
    lvl2 = bytearray(packed)
    lvl2[0] = (lvl2[0] & ~0x0C) | (2 << 2)                  # level 2 packets are rejected, not misread
    with pytest.raises(pkg.MslamHipError):
        pkg.qlz_decompress(bytes(lvl2), n)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 3])
def test_compressed_vocabulary_loads(pkg, orc, level, synth_frames):
    """Vocabulary::fromStream with compressed = true (dbow3.patch:2594-2611): same words and vectors as the plain stream"""
    blob = synth.make_vocabulary(10, 4, seed=9)
    packed = quicklz.compress_vocabulary(blob, level)
    assert len(packed) < len(blob) and packed[8] == 1
    V = orc.Vocabulary(blob)
    c = pkg.Context(width=0, height=0, max_keypoints=4096)
    c.bow_load(packed)
    info = c.bow_info()
    assert (info["k"], info["L"], info["n_nodes"], info["n_words"]) == (V.k, V.L, V.n_nodes, V.n_words)
    d = orc.detect(synth_frames[0], orc.params())["desc"]
    gw, gwt = c.bow_words(d)
    rw, rwt = V.words(d)
    assert np.array_equal(gw, rw) and np.array_equal(gwt, rwt)
    gv, rv = c.bow_transform(d), V.bow_vector(d)
    assert np.array_equal(gv[0], rv[0]) and np.array_equal(gv[1], rv[1])
    with pytest.raises(pkg.MslamHipError) as e:
        c.bow_load(packed[:len(packed) // 2])
    assert e.value.code == pkg.E_FORMAT
    c.close()
