// quicklz_decode.hip — host-side decoder for QuickLZ 1.5.x packets (levels 1 and 3, streaming buffer 0).
//
// DBoW3 stores a vocabulary saved with compressed = true as QuickLZ packets of 10 000 input bytes each
// (Vocabulary::toStream / fromStream, conan_recipes/dbow3/dbow3.patch:2325-2349, :2594-2611: qlz_compress /
// qlz_size_compressed / qlz_decompress).  QuickLZ itself is not in the reference tree (DBoW3 bundles quicklz.c,
// version 1.5.0, compression level 1), so this is a restatement of its published packet format:
//
//   header  byte 0: bit 0 = payload is compressed, bit 1 = long header (4-byte sizes, else 1-byte), bits 2-3 = level,
//                   bits 4-5 = streaming-buffer class, bit 6 = 1;  then compressed size (whole packet, header
//                   included) and decompressed size.
//   payload a 32-bit control word precedes every group of up to 31 tokens, consumed LSB first, bit 31 is the end
//           sentinel; flag 0 = literal byte(s), flag 1 = match.
//   level 1 match: 16 bits = hash (12 bits, << 4) | (length - 2) for lengths 3..17, or 24 bits with the low nibble 0
//           and the length in the third byte.  The source of the match is NOT an offset but the most recent position
//           whose first 3 bytes hash to `hash` — the decoder mirrors the compressor's hash table
//           (hash(v) = ((v >> 12) ^ v) & 4095 over the 3-byte value), updating it for every position up to the
//           current one minus 3, and, after a match, up to the start of the match only.
//   level 3 match: five encodings with explicit offsets (1 to 4 bytes), selected by the low bits.
//   the last 10 bytes of a packet are always literals (UNCONDITIONAL_MATCHLEN 6 + UNCOMPRESSED_END 4).
// Unverified against real QuickLZ output (no QuickLZ exists in this build image): tests/test_quicklz.py checks it
// against an independent Python decoder on streams written by tools/quicklz.py's encoders.
#include "../../include/mslam_hip.h"

#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

namespace
{
inline uint32_t rd(const uint8_t* p, const uint8_t* end, int n)
{
    uint32_t v = 0;
    for(int i = 0; i < n && p + i < end; ++i)
        v |= (uint32_t)p[i] << (8 * i);
    return v;
}

inline uint32_t hash3(uint32_t v)
{
    v &= 0xFFFFFFu;
    return ((v >> 12) ^ v) & 4095u;
}

// decodes ONE packet starting at src; returns 0 or an error code; *consumed / *produced are set on success
int decode_packet(const uint8_t* src, size_t avail, uint8_t* dst, size_t dst_cap, size_t* consumed, size_t* produced)
{
    if(avail < 3)
        return MSLAM_HIP_E_FORMAT;
    const uint8_t flags = src[0];
    const int n = (flags & 2) ? 4 : 1;
    const size_t header = 2 * (size_t)n + 1;
    if(avail < header)
        return MSLAM_HIP_E_FORMAT;
    const size_t csize = rd(src + 1, src + avail, n), dsize = rd(src + 1 + n, src + avail, n);
    const int level = (flags >> 2) & 3;
    if(csize < header || csize > avail || dsize > dst_cap)
        return MSLAM_HIP_E_FORMAT;
    *consumed = csize;
    *produced = dsize;
    if(!(flags & 1))
    {
        if(csize - header < dsize)
            return MSLAM_HIP_E_FORMAT;
        std::memcpy(dst, src + header, dsize);
        return MSLAM_HIP_OK;
    }
    if(level != 1 && level != 3)
        return MSLAM_HIP_E_FORMAT; // level 2 packets are not handled
    if(dsize == 0)
        return MSLAM_HIP_OK;
    const uint8_t* s = src + header;
    const uint8_t* s_end = src + csize;
    size_t d = 0;
    const long long last = (long long)dsize - 1;
    const long long last_matchstart = last - 6 - 4;
    long long last_hashed = -1;
    std::vector<uint32_t> table; // level 1: hash -> position + 1 (0 = never set)
    if(level == 1)
        table.assign(4096, 0);
    auto update_upto = [&](long long max) { // hash every position in (last_hashed, max]
        while(last_hashed < max)
        {
            ++last_hashed;
            table[hash3(dst[last_hashed] | (dst[last_hashed + 1] << 8) | ((uint32_t)dst[last_hashed + 2] << 16))] =
                (uint32_t)last_hashed + 1;
        }
    };
    uint32_t cword = 1;
    for(;;)
    {
        if(cword == 1)
        {
            if(s + 4 > s_end)
                return MSLAM_HIP_E_FORMAT;
            cword = rd(s, s_end, 4);
            s += 4;
        }
        const uint32_t fetch = rd(s, s_end, 4);
        if(cword & 1)
        {
            cword >>= 1;
            size_t matchlen, from;
            if(level == 1)
            {
                const uint32_t h = (fetch >> 4) & 0xFFFu;
                if(table[h] == 0)
                    return MSLAM_HIP_E_FORMAT;
                from = table[h] - 1;
                if(fetch & 0xF)
                    matchlen = (fetch & 0xF) + 2, s += 2;
                else
                    matchlen = (fetch >> 16) & 0xFF, s += 3;
            }
            else
            {
                uint32_t offset;
                if((fetch & 3) == 0)
                    offset = (fetch & 0xFF) >> 2, matchlen = 3, s += 1;
                else if((fetch & 2) == 0)
                    offset = (fetch & 0xFFFF) >> 2, matchlen = 3, s += 2;
                else if((fetch & 1) == 0)
                    offset = (fetch & 0xFFFF) >> 6, matchlen = ((fetch >> 2) & 15) + 3, s += 2;
                else if((fetch & 127) != 3)
                    offset = (fetch >> 7) & 0x1FFFF, matchlen = ((fetch >> 2) & 0x1F) + 2, s += 3;
                else
                    offset = fetch >> 15, matchlen = ((fetch >> 7) & 255) + 3, s += 4;
                if(offset == 0 || offset > d)
                    return MSLAM_HIP_E_FORMAT;
                from = d - offset;
            }
            if(s > s_end || matchlen < 3 || from >= d || d + matchlen > dsize)
                return MSLAM_HIP_E_FORMAT;
            for(size_t i = 0; i < matchlen; ++i) // forward byte copy: the regions may overlap
                dst[d + i] = dst[from + i];
            d += matchlen;
            if(level == 1)
            {
                update_upto((long long)(d - matchlen));
                last_hashed = (long long)d - 1;
            }
        }
        else if((long long)d < last_matchstart)
        {
            static const int bitlut[16] = {4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
            const int k = bitlut[cword & 0xF];
            if(s + k > s_end)
                return MSLAM_HIP_E_FORMAT;
            for(int i = 0; i < k; ++i)
                dst[d + i] = s[i];
            cword >>= k;
            d += k;
            s += k;
            if(level == 1)
                update_upto((long long)d - 3);
        }
        else
        {
            while((long long)d <= last)
            {
                if(cword == 1)
                {
                    s += 4;
                    cword = 1u << 31;
                }
                if(s >= s_end)
                    return MSLAM_HIP_E_FORMAT;
                dst[d++] = *s++;
                cword >>= 1;
            }
            return MSLAM_HIP_OK;
        }
    }
}
} // namespace

namespace mslam
{
// a sequence of packets (DBoW3 writes one per 10 000 input bytes) -> the concatenated output
int qlz_decode_stream(const uint8_t* src, size_t size, uint32_t n_packets, std::vector<uint8_t>& out)
{
    size_t pos = 0;
    for(uint32_t i = 0; i < n_packets; ++i)
    {
        if(size - pos < 3)
            return MSLAM_HIP_E_FORMAT;
        const int n = (src[pos] & 2) ? 4 : 1;
        if(size - pos < (size_t)(2 * n + 1))
            return MSLAM_HIP_E_FORMAT;
        // the header is untrusted: bound both sizes before anything is allocated.  A match token is at least one input
        // byte and yields at most 258 output bytes (decode_packet), so no valid packet has dsize > 258 * csize; the whole
        // stream is capped at 4 GiB.
        const size_t csize = rd(src + pos + 1, src + size, n);
        const size_t dsize = rd(src + pos + 1 + n, src + size, n);
        if(csize < (size_t)(2 * n + 1) || csize > size - pos || dsize > csize * 258 || out.size() + dsize > (size_t)1 << 32)
            return MSLAM_HIP_E_FORMAT;
        const size_t at = out.size();
        out.resize(at + dsize + 4); // + 4: the level-1 hash update reads two bytes past a position
        size_t consumed = 0, produced = 0;
        const int rc = decode_packet(src + pos, size - pos, out.data() + at, dsize, &consumed, &produced);
        if(rc)
            return rc;
        out.resize(at + produced);
        pos += consumed;
    }
    return MSLAM_HIP_OK;
}
} // namespace mslam

extern "C" int mslam_hip_qlz_decompress(const void* src, size_t src_size, uint32_t n_packets, void* dst, size_t dst_capacity,
                                        size_t* dst_size)
{
    if(!src || !dst_size)
        return MSLAM_HIP_E_INVALID;
    std::vector<uint8_t> out;
    int rc;
    try
    {
        rc = mslam::qlz_decode_stream(static_cast<const uint8_t*>(src), src_size, n_packets, out);
    }
    catch(const std::bad_alloc&) // no C++ exception crosses the C ABI
    {
        return MSLAM_HIP_E_RUNTIME;
    }
    if(rc)
        return rc;
    *dst_size = out.size();
    if(out.size() > dst_capacity || (!dst && !out.empty()))
        return MSLAM_HIP_E_CAPACITY;
    if(!out.empty())
        std::memcpy(dst, out.data(), out.size());
    return MSLAM_HIP_OK;
}
