// arc_score.hpp — FAST-9/16 corner score of two pixels per lane (shared by k_fast.hip and k_cvorb.hip).
//
// cornerScore<16>(.., t) of OpenCV's FAST for a pixel that passes the 9-contiguous test at threshold t equals
//   S = max( max over the 16 arcs of 9 of min(d), max over the arcs of min(-d) ) - 1,   d = centre - circle pixel,
// and the pixel passes the test iff S >= t, so the test itself never has to be evaluated.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mslam
{

// Packed 3-input min / max of two 16-bit values per register.  gfx950 has no integer form, but positive floats order like
// their bit patterns, so v_pk_minimum3_f16 / v_pk_maximum3_f16 ARE the unsigned 16-bit min3 / max3 for patterns between
// 0x0400 and 0x7BFF (positive normal halves: no NaN, no denormal).  The arc score keeps its differences biased into that
// range (d + 1280 in [1025, 1535]).
__device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
constexpr int kArcBias = 1280; // 0x0500

// arc scores of TWO pixels at once: e[k] = (centre - circle pixel k) + kArcBias in each half
__device__ __forceinline__ void arc_score2(const uint32_t (&e)[16], int& sa, int& sb)
{
    uint32_t mn3[16], mx3[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn3[i] = pk_min3(e[i], e[(i + 1) & 15], e[(i + 2) & 15]);
        mx3[i] = pk_max3(e[i], e[(i + 1) & 15], e[(i + 2) & 15]);
    }
    uint32_t mn9[16], mx9[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn9[i] = pk_min3(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]);
        mx9[i] = pk_max3(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]);
    }
    uint32_t q0 = pk_max3(mn9[0], mn9[1], mn9[2]), q1 = pk_min3(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for(int i = 3; i < 15; i += 2)
    {
        q0 = pk_max3(q0, mn9[i], mn9[i + 1]);
        q1 = pk_min3(q1, mx9[i], mx9[i + 1]);
    }
    q0 = pk_max3(q0, mn9[15], mn9[15]);
    q1 = pk_min3(q1, mx9[15], mx9[15]);
    sa = max((int)(q0 & 0xFFFFu) - kArcBias, kArcBias - (int)(q1 & 0xFFFFu)) - 1;
    sb = max((int)(q0 >> 16) - kArcBias, kArcBias - (int)(q1 >> 16)) - 1;
}

// The same score from the RAW circle pixels p[k] (two candidates per register, 16-bit halves holding 0..255):
//   S = max(c - min over arcs of max(p), max over arcs of min(p) - c) - 1
// — min / max commute with the subtraction from the centre, so the sixteen differences are never formed.  The halves are
// f16 DENORMALS here (patterns 0x0000..0x00FF): they order like their integers as long as the f16 denormal mode keeps
// them (the AMDGPU default for f16 / f64; the kernels that use this are compiled without -fgpu-flush-denormals-to-zero,
// and tests/test_gpu_parity.py would see a flush at once: every score would collapse to the centre).
__device__ __forceinline__ void arc_score2_raw(const uint32_t (&p)[16], int ca, int cb, int& sa, int& sb)
{
    uint32_t mn3[16], mx3[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn3[i] = pk_min3(p[i], p[(i + 1) & 15], p[(i + 2) & 15]);
        mx3[i] = pk_max3(p[i], p[(i + 1) & 15], p[(i + 2) & 15]);
    }
    uint32_t mn9[16], mx9[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn9[i] = pk_min3(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]);
        mx9[i] = pk_max3(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]);
    }
    uint32_t q0 = pk_max3(mn9[0], mn9[1], mn9[2]), q1 = pk_min3(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for(int i = 3; i < 15; i += 2)
    {
        q0 = pk_max3(q0, mn9[i], mn9[i + 1]);
        q1 = pk_min3(q1, mx9[i], mx9[i + 1]);
    }
    q0 = pk_max3(q0, mn9[15], mn9[15]); // max over the arcs of the arc's darkest pixel
    q1 = pk_min3(q1, mx9[15], mx9[15]); // min over the arcs of the arc's brightest pixel
    sa = max(ca - (int)(q1 & 0xFFFFu), (int)(q0 & 0xFFFFu) - ca) - 1;
    sb = max(cb - (int)(q1 >> 16), (int)(q0 >> 16) - cb) - 1;
}

} // namespace mslam
