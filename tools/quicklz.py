"""QuickLZ 1.5.x packet format in Python: an encoder for levels 1 and 3 and an independent decoder.

Test tooling for the compressed-vocabulary path (DBoW3 Vocabulary::toStream(compressed = true),
conan_recipes/dbow3/dbow3.patch:2325-2349): `compress_vocabulary` rewrites an uncompressed DBoW3 stream in the
compressed layout — magic, compressed = 1, nnodes, nChunks, then one packet per 10 000 bytes.

QuickLZ is not available here, so these are restatements of the published format (see
modular-slam_amd/csrc/quicklz_decode.hip for the layout).  The level-1 encoder emits a match only where the DECODER's
mirrored hash table points at a position holding the same bytes, which is the property real QuickLZ streams have
by construction; it does not try to reproduce QuickLZ's own parsing choices byte for byte."""
import struct


def _hash3(v):
    v &= 0xFFFFFF
    return ((v >> 12) ^ v) & 4095


class _Tokens:
    """token writer: a 32-bit control word (sentinel in bit 31) in front of every group of up to 31 tokens"""

    def __init__(self):
        self.out = bytearray()
        self._start()

    def _start(self):
        self.cw_pos = len(self.out)
        self.out += b"\0\0\0\0"
        self.bits = 0
        self.n = 0

    def _flush(self):
        struct.pack_into("<I", self.out, self.cw_pos, self.bits | (1 << self.n if self.n < 31 else 1 << 31))

    def token(self, is_match, payload):
        if self.n == 31:
            self._flush()
            self._start()
        self.bits |= (1 if is_match else 0) << self.n
        self.n += 1
        self.out += payload

    def finish(self):
        # QuickLZ shifts the partial word down to the sentinel and sets bit 31: flags in the low bits, sentinel right above
        struct.pack_into("<I", self.out, self.cw_pos, self.bits | (1 << 31))
        return bytes(self.out)


def _header(level, csize_payload, dsize, compressed=True):
    flags = (1 if compressed else 0) | 2 | (level << 2) | (1 << 6)
    return struct.pack("<BII", flags, 9 + csize_payload, dsize)


def compress_packet(data, level=1):
    """one QuickLZ packet for `data` (level 1 or 3); falls back to a stored packet when that is smaller"""
    data = bytes(data)
    n = len(data)
    if n == 0:
        raise ValueError("empty input")
    last = n - 1
    last_matchstart = last - 6 - 4
    t = _Tokens()
    pos = 0
    if level == 1:
        table = {}            # decoder's view: hash -> most recent hashed position
        last_hashed = -1

        def update_upto(mx):
            nonlocal last_hashed
            while last_hashed < mx:
                last_hashed += 1
                p = last_hashed
                table[_hash3(data[p] | (data[p + 1] << 8) | (data[p + 2] << 16))] = p
        while pos <= last_matchstart:
            h = _hash3(data[pos] | (data[pos + 1] << 8) | (data[pos + 2] << 16))
            src = table.get(h)
            ml = 0
            if src is not None and src < pos:
                limit = min(255, last - 4 - pos + 1)
                while ml < limit and data[src + ml] == data[pos + ml]:
                    ml += 1
            if ml >= 3:
                if ml < 18:
                    t.token(True, struct.pack("<H", (h << 4) | (ml - 2)))
                else:
                    t.token(True, struct.pack("<H", h << 4) + bytes([ml]))
                pos += ml
                update_upto(pos - ml)
                last_hashed = pos - 1
            else:
                t.token(False, data[pos:pos + 1])
                pos += 1
                update_upto(pos - 3)
    elif level == 3:
        index = {}
        while pos <= last_matchstart:
            key = data[pos:pos + 3]
            src = index.get(key)
            index[key] = pos
            ml = 0
            if src is not None:
                limit = min(258, last - 4 - pos + 1)
                while ml < limit and data[src + ml] == data[pos + ml]:
                    ml += 1
            off = pos - src if src is not None else 0
            if ml >= 3 and off < (1 << 17):
                if ml == 3 and off < 64:
                    t.token(True, bytes([off << 2]))
                elif ml == 3 and off < (1 << 14):
                    t.token(True, struct.pack("<H", (off << 2) | 1))
                elif ml <= 18 and off < 1024:
                    t.token(True, struct.pack("<H", (off << 6) | ((ml - 3) << 2) | 2))
                elif ml <= 33:
                    t.token(True, struct.pack("<I", (off << 7) | ((ml - 2) << 2) | 3)[:3])
                else:
                    t.token(True, struct.pack("<I", (off << 15) | ((ml - 3) << 7) | 3))
                for p in range(pos + 1, pos + ml):
                    index[data[p:p + 3]] = p
                pos += ml
            else:
                t.token(False, data[pos:pos + 1])
                pos += 1
    else:
        raise ValueError("level 1 or 3")
    while pos <= last:
        t.token(False, data[pos:pos + 1])
        pos += 1
    payload = t.finish()
    if len(payload) >= n:
        return _header(level, n, n, compressed=False) + data
    return _header(level, len(payload), n) + payload


def decompress_packet(src):
    """independent decoder of one packet: returns (output bytes, bytes consumed)"""
    flags = src[0]
    n = 4 if flags & 2 else 1
    header = 2 * n + 1
    csize = int.from_bytes(src[1:1 + n], "little")
    dsize = int.from_bytes(src[1 + n:1 + 2 * n], "little")
    level = (flags >> 2) & 3
    if not flags & 1:
        return bytes(src[header:header + dsize]), csize
    out = bytearray()
    s = header
    cword = 1
    last = dsize - 1
    last_matchstart = last - 10
    table, last_hashed = {}, -1

    def upto(mx):
        nonlocal last_hashed
        while last_hashed < mx:
            last_hashed += 1
            p = last_hashed
            table[_hash3(out[p] | (out[p + 1] << 8) | (out[p + 2] << 16))] = p
    while True:
        if cword == 1:
            cword = int.from_bytes(src[s:s + 4], "little")
            s += 4
        fetch = int.from_bytes(src[s:s + 4].ljust(4, b"\0"), "little")
        if cword & 1:
            cword >>= 1
            if level == 1:
                frm = table[(fetch >> 4) & 0xFFF]
                if fetch & 0xF:
                    ml, s = (fetch & 0xF) + 2, s + 2
                else:
                    ml, s = (fetch >> 16) & 0xFF, s + 3
            else:
                if fetch & 3 == 0:
                    off, ml, s = (fetch & 0xFF) >> 2, 3, s + 1
                elif fetch & 2 == 0:
                    off, ml, s = (fetch & 0xFFFF) >> 2, 3, s + 2
                elif fetch & 1 == 0:
                    off, ml, s = (fetch & 0xFFFF) >> 6, ((fetch >> 2) & 15) + 3, s + 2
                elif fetch & 127 != 3:
                    off, ml, s = (fetch >> 7) & 0x1FFFF, ((fetch >> 2) & 0x1F) + 2, s + 3
                else:
                    off, ml, s = fetch >> 15, ((fetch >> 7) & 255) + 3, s + 4
                frm = len(out) - off
            for i in range(ml):
                out.append(out[frm + i])
            if level == 1:
                upto(len(out) - ml)
                last_hashed = len(out) - 1
        elif len(out) < last_matchstart:
            k = (4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0)[cword & 0xF]
            out += src[s:s + k]
            cword >>= k
            s += k
            if level == 1:
                upto(len(out) - 3)
        else:
            while len(out) <= last:
                if cword == 1:
                    s += 4
                    cword = 1 << 31
                out.append(src[s])
                s += 1
                cword >>= 1
            return bytes(out), csize


def compress_stream(data, level=1, chunk=10000):
    """DBoW3's chunking: (number of packets, concatenated packets)"""
    packets = [compress_packet(data[i:i + chunk], level) for i in range(0, len(data), chunk)]
    return len(packets), b"".join(packets)


def compress_vocabulary(blob, level=1):
    """uncompressed DBoW3 vocabulary stream -> the layout Vocabulary::toStream(compressed = true) writes"""
    sig, comp, n_nodes = struct.unpack_from("<QBI", blob, 0)
    assert sig == 88877711233 and comp == 0
    n, packed = compress_stream(blob[13:], level)
    return struct.pack("<QBII", sig, 1, n_nodes, n) + packed


def decompress_vocabulary(blob):
    """the inverse of compress_vocabulary: a DBoW3 vocabulary stream saved with compressed = true -> the plain stream
    (returned unchanged when it is not compressed)"""
    sig, comp, n_nodes = struct.unpack_from("<QBI", blob, 0)
    assert sig == 88877711233
    if comp == 0:
        return bytes(blob)
    n, = struct.unpack_from("<I", blob, 13)
    pos, out = 17, bytearray()
    for _ in range(n):
        data, used = decompress_packet(bytes(blob[pos:]))
        out += data
        pos += used
    return struct.pack("<QBI", sig, 0, n_nodes) + bytes(out)
