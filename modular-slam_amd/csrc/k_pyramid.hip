// k_pyramid.hip — BGR -> gray (pyramid level 0) and the chained fixed-point bilinear pyramid.
//
// gray   : reference frame.cpp:6-27 (toGrayScale); float mul/add left to right, no FMA, truncation.
// resize : cv::resize(INTER_LINEAR, CV_8UC1) as called at distributed_cv_feature.cpp:839; the
//          coefficient tables are built on the host (api.hip) exactly as cv::resize builds them,
//          the kernel does the integer interpolation (HResizeLinear / VResizeLinear<uchar,...>).
//
// Both are streaming, HBM-bound kernels: 4 pixels per lane, dword loads/stores, rows coalesced.
#include "common.hpp"

namespace mslam
{

__device__ __forceinline__ uint32_t gray_px(uint32_t c0, uint32_t c1, uint32_t c2)
{
    // (0.299f*c0 + 0.587f*c1) + 0.114f*c2 with every operation individually rounded (no contraction)
    float v = __fadd_rn(__fmul_rn(0.299f, (float)c0), __fmul_rn(0.587f, (float)c1));
    v = __fadd_rn(v, __fmul_rn(0.114f, (float)c2));
    v = fminf(255.0f, v);
    return (uint32_t)(int)v; // static_cast<uint8_t> of a non-negative float: truncation
}

// W % 4 == 0: one lane converts 4 pixels = 12 source bytes (3 aligned dwords) -> 1 dword.
__global__ __launch_bounds__(256) void k_gray4(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ pyr, int W, int H,
                                               int pitch, unsigned slab, int frame0)
{
    const int quad = blockIdx.x * 256 + threadIdx.x;
    const int n_quads = (W * H) >> 2;
    if(quad >= n_quads)
        return;
    const size_t frame = blockIdx.y + frame0;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(bgr + frame * (size_t)W * H * 3) + (size_t)quad * 3;
    const uint32_t a = src[0], b = src[1], c = src[2];
    // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3   (little endian)
    const uint32_t p0 = gray_px(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF);
    const uint32_t p1 = gray_px(a >> 24, b & 0xFF, (b >> 8) & 0xFF);
    const uint32_t p2 = gray_px((b >> 16) & 0xFF, b >> 24, c & 0xFF);
    const uint32_t p3 = gray_px((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24);
    const int px = quad << 2;
    const int y = px / W, x = px - y * W;
    *reinterpret_cast<uint32_t*>(pyr + frame * slab + (size_t)y * pitch + x) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}

// generic width: one pixel per lane
__global__ __launch_bounds__(256) void k_gray1(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ pyr, int W, int H,
                                               int pitch, unsigned slab, int frame0)
{
    const int px = blockIdx.x * 256 + threadIdx.x;
    if(px >= W * H)
        return;
    const size_t frame = blockIdx.y + frame0;
    const uint8_t* src = bgr + frame * (size_t)W * H * 3 + (size_t)px * 3;
    const int y = px / W, x = px - y * W;
    pyr[frame * slab + (size_t)y * pitch + x] = (uint8_t)gray_px(src[0], src[1], src[2]);
}

void launch_gray(const uint8_t* d_bgr, uint8_t* d_pyr, const Geometry& g, int frame0, int n_frames, hipStream_t s)
{
    const LevelGeom& l0 = g.lv[0];
    if((g.W & 3) == 0)
    {
        const int n_quads = (g.W * g.H) >> 2;
        dim3 grid((n_quads + 255) / 256, n_frames);
        hipLaunchKernelGGL(k_gray4, grid, dim3(256), 0, s, d_bgr, d_pyr + l0.offset, g.W, g.H, l0.pitch, g.slab, frame0);
    }
    else
    {
        dim3 grid((g.W * g.H + 255) / 256, n_frames);
        hipLaunchKernelGGL(k_gray1, grid, dim3(256), 0, s, d_bgr, d_pyr + l0.offset, g.W, g.H, l0.pitch, g.slab, frame0);
    }
}

// One lane produces 4 horizontally adjacent destination pixels (one dword store).
// coef words: low 16 bits = weight of S[ofs], high 16 bits = weight of S[ofs+1] (both <= 2048).
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, unsigned slab, int src_off, int sw, int sh,
                                                int spitch, int dst_off, int dw, int dh, int dpitch,
                                                const int32_t* __restrict__ xofs, const uint32_t* __restrict__ xcoef,
                                                const int32_t* __restrict__ yofs, const uint32_t* __restrict__ ycoef,
                                                int frame0)
{
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if(dy >= dh || (qx << 2) >= dw)
        return;
    const size_t frame = blockIdx.z + frame0;
    const uint8_t* src = pyr + frame * slab + src_off;
    uint8_t* dst = pyr + frame * slab + dst_off;

    int sy0 = yofs[dy], sy1 = sy0 + 1;
    sy0 = max(0, min(sy0, sh - 1)); // resizeGeneric_Invoker clips the row index, not the weight
    sy1 = max(0, min(sy1, sh - 1));
    const uint32_t yc = ycoef[dy];
    const int b0 = (int)(yc & 0xFFFF), b1 = (int)(yc >> 16);
    const uint8_t* S0 = src + (size_t)sy0 * spitch;
    const uint8_t* S1 = src + (size_t)sy1 * spitch;

    uint32_t out = 0;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        const int dx = min((qx << 2) + k, dw - 1); // pad lanes recompute the last pixel; pad bytes are never read
        const int sx = xofs[dx];
        const int sx1 = min(sx + 1, sw - 1);
        const uint32_t xc = xcoef[dx];
        const int a0 = (int)(xc & 0xFFFF), a1 = (int)(xc >> 16);
        const int r0 = S0[sx] * a0 + S0[sx1] * a1;
        const int r1 = S1[sx] * a0 + S1[sx1] * a1;
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 0xFF) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)dy * dpitch + (qx << 2)) = out;
}

// Quad form of the same interpolation: the 4 destination pixels of a lane draw on at most 12
// consecutive source bytes per row, so the lane loads three ALIGNED dwords per source row.  The host
// table holds, per destination pixel, which dword pair contains its two source bytes and at which byte
// offset, so ONE v_perm_b32 per row cuts (S[sx], S[sx+1]) out as a u16 pair and the horizontal blend
// is one v_dot2_u32_u16.  Table word: i0 (3 bits: byte index of S[sx] inside the chosen pair) |
// upper-pair flag (bit 3) | a0 << 4 | a1 << 16.
__global__ __launch_bounds__(256) void k_resize_quad(uint8_t* __restrict__ pyr, unsigned slab, int src_off, int sh,
                                                     int spitch, int dst_off, int dw, int dh, int dpitch,
                                                     const uint32_t* __restrict__ qbase, const uint4* __restrict__ qw,
                                                     const int32_t* __restrict__ yofs, const uint32_t* __restrict__ ycoef,
                                                     int frame0)
{
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if(dy >= dh || (qx << 2) >= dw)
        return;
    const size_t frame = blockIdx.z + frame0;
    const uint8_t* src = pyr + frame * slab + src_off;
    uint8_t* dst = pyr + frame * slab + dst_off;

    int sy0 = yofs[dy], sy1 = sy0 + 1;
    sy0 = max(0, min(sy0, sh - 1)); // resizeGeneric_Invoker clips the row index, not the weight
    sy1 = max(0, min(sy1, sh - 1));
    const uint32_t yc = ycoef[dy];
    const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
    const uint32_t base = qbase[qx];
    const uint4 w4 = qw[qx];
    const uint32_t* r0p = reinterpret_cast<const uint32_t*>(src + sy0 * spitch + base);
    const uint32_t* r1p = reinterpret_cast<const uint32_t*>(src + sy1 * spitch + base);
    const uint32_t a0 = r0p[0], a1 = r0p[1], a2 = r0p[2];
    const uint32_t c0 = r1p[0], c1 = r1p[1], c2 = r1p[2];
    const uint32_t ws[4] = {w4.x, w4.y, w4.z, w4.w};
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    uint32_t out = 0;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        const uint32_t w = ws[k];
        // selector bytes [i0, zero, i0+1, zero]: the pair lands as two u16 lanes
        const uint32_t sel = 0x0c010c00u + (w & 7u) * 0x00010001u;
        const bool upper = (w & 8u) != 0;
        const uint32_t p0 = __builtin_amdgcn_perm(upper ? a2 : a1, upper ? a1 : a0, sel);
        const uint32_t p1 = __builtin_amdgcn_perm(upper ? c2 : c1, upper ? c1 : c0, sel);
        const uint32_t coef = ((w >> 4) & 0xFFF) | (w & 0xFFFF0000u); // (a0, a1) as a u16 pair
        u16x2 cv, v0, v1;
        cv.x = (unsigned short)(coef & 0xFFFF);
        cv.y = (unsigned short)(coef >> 16);
        v0.x = (unsigned short)(p0 & 0xFFFF);
        v0.y = (unsigned short)(p0 >> 16);
        v1.x = (unsigned short)(p1 & 0xFFFF);
        v1.y = (unsigned short)(p1 >> 16);
        const uint32_t r0 = __builtin_amdgcn_udot2(v0, cv, 0u, false);
        const uint32_t r1 = __builtin_amdgcn_udot2(v1, cv, 0u, false);
        // both factors are below 2^24 (weights <= 2048, r >> 4 <= 32640): 24-bit multiplies are full rate
        const uint32_t v = ((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2) >> 2;
        out |= v << (8 * k); // v <= 255
    }
    *reinterpret_cast<uint32_t*>(dst + dy * dpitch + (qx << 2)) = out;
}

void launch_resize_quad(uint8_t* d_pyr, const Geometry& g, int level, const uint32_t* d_qbase, const uint4* d_qw,
                        const int32_t* d_yofs, const uint32_t* d_ycoef, int frame0, int n_frames, hipStream_t s)
{
    const LevelGeom& src = g.lv[level - 1];
    const LevelGeom& dst = g.lv[level];
    const int quads = (dst.w + 3) / 4;
    dim3 grid((quads + 63) / 64, (dst.h + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize_quad, grid, dim3(64, 4), 0, s, d_pyr, g.slab, src.offset, src.h, src.pitch, dst.offset,
                       dst.w, dst.h, dst.pitch, d_qbase, d_qw, d_yofs, d_ycoef, frame0);
}

void launch_resize(uint8_t* d_pyr, const Geometry& g, int level, const int32_t* d_xofs, const uint32_t* d_xcoef,
                   const int32_t* d_yofs, const uint32_t* d_ycoef, int frame0, int n_frames, hipStream_t s)
{
    const LevelGeom& src = g.lv[level - 1];
    const LevelGeom& dst = g.lv[level];
    const int quads = (dst.w + 3) / 4;
    dim3 grid((quads + 63) / 64, (dst.h + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize, grid, dim3(64, 4), 0, s, d_pyr, g.slab, src.offset, src.w, src.h, src.pitch, dst.offset,
                       dst.w, dst.h, dst.pitch, d_xofs, d_xcoef, d_yofs, d_ycoef, frame0);
}

} // namespace mslam
