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

// Column form of the same interpolation.  One lane owns one destination QUAD COLUMN (4 pixels wide) of one
// frame and walks down R destination rows; lanes are flattened over (frame, quad) so that narrow levels keep
// all 64 lanes busy, and every lane of a wave is on the same rows, so the row logic is scalar.
//   horizontal (HResizeLinear): the 4 pixels draw on at most 12 consecutive source bytes, loaded as three
//     ALIGNED dwords; the host table (3 x uint4 per quad: byte offset of the window, per-pixel v_perm selector
//     cutting (S[sx], S[sx+1]) out as a u16 pair, per-pixel (a0, a1) coefficient pair, flags: pair in dwords
//     (1,2) instead of (0,1)) makes it one v_perm_b32 + one v_dot2_u32_u16 per pixel.  The interpolated values
//     (>> 4) of the last two source rows stay in registers: at scale 1.2 a destination row re-uses the lower
//     source row of the previous one 4 times out of 5, so 1.2 source rows are interpolated per row, not 2;
//   vertical (VResizeLinear): two 24-bit multiplies per pixel, one dword store per quad.
// EXACT = INTER_LINEAR_EXACT (the cv::ORB detector mode's pyramid, k_cvorb.hip): the same walk with 8.8 weights,
//   h = c0 S[o] + c1 S[o+1] (u16, exact), out = (b0 h0 + b1 h1 + 2^15) >> 16; the host tables fold its edge rules in.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_resize_col(ResizeColArgs a)
{
    // the block's row table lives in lane registers (lane i: destination row R0 + i, R <= 64) and is read with
    // v_readlane: wave-uniform values without a memory access inside the loop.  Loaded by ALL lanes, before the
    // tail lanes leave: v_readlane also reads lanes that have exited since.
    const int R0 = blockIdx.y * a.R, r_end = min(R0 + a.R, a.dh);
    const int lane = threadIdx.x & 63;
    const int my_y0 = a.yofs[min(R0 + lane, a.dh - 1)];
    const uint32_t my_yc = a.ycoef[min(R0 + lane, a.dh - 1)];
    // resizeGeneric_Invoker clips the row index, not the weight
    const int s_lo = max(0, min(__builtin_amdgcn_readlane(my_y0, 0), a.sh - 1));
    const int s_hi = max(0, min(__builtin_amdgcn_readlane(my_y0, r_end - 1 - R0) + 1, a.sh - 1));
    // tail lanes stay alive on a clamped item (v_readlane must be able to read every lane's row table); only
    // their stores are masked
    const int n_items = a.n_frames * a.quads;
    const bool live = (int)(blockIdx.x * 256 + threadIdx.x) < n_items;
    const int idx = min((int)(blockIdx.x * 256 + threadIdx.x), n_items - 1);
    const int f = (int)(((float)idx + 0.5f) * a.inv_quads); // idx / quads, exact: see launch_resize_col
    const int qx = idx - f * a.quads;
    const size_t frame = (size_t)f + a.frame0;
    const uint4 t0 = a.qt[3 * qx], t1 = a.qt[3 * qx + 1], t2 = a.qt[3 * qx + 2];
    const uint8_t* src = a.pyr + frame * a.slab + a.src_off + t0.x;
    uint8_t* dst = a.pyr + frame * a.slab + a.dst_off + (qx << 2);
    const uint32_t sel[4] = {t0.z, t0.w, t1.x, t1.y}, coef[4] = {t1.z, t1.w, t2.x, t2.y};
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

    struct Raw
    {
        uint32_t d0, d1, d2;
    };
    struct Row
    {
        uint32_t h[4];
    };
    auto load = [&](int sy) {
        const uint32_t* rp = reinterpret_cast<const uint32_t*>(src + (size_t)sy * a.spitch);
        return Raw{rp[0], rp[1], rp[2]};
    };
    auto hinterp = [&](const Raw& w) {
        Row r;
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            uint32_t lo = w.d0, hi = w.d1;
            if(a.need_mask & (1 << k)) // wave-uniform: this pixel position uses the upper pair somewhere on the level
            {
                const bool up = (t0.y >> k) & 1u;
                lo = up ? w.d1 : w.d0;
                hi = up ? w.d2 : w.d1;
            }
            const uint32_t pr = __builtin_amdgcn_perm(hi, lo, sel[k]);
            u16x2 pv, cv;
            pv.x = (unsigned short)(pr & 0xFFFF);
            pv.y = (unsigned short)(pr >> 16);
            cv.x = (unsigned short)(coef[k] & 0xFFFF);
            cv.y = (unsigned short)(coef[k] >> 16);
            r.h[k] = EXACT ? __builtin_amdgcn_udot2(pv, cv, 0u, false) : __builtin_amdgcn_udot2(pv, cv, 0u, false) >> 4;
        }
        return r;
    };

    // The block's destination rows draw on the contiguous source rows s_lo .. s_hi.  They are streamed
    // through a 4-deep register ring (loads issued 4 rows ahead); after source row s is interpolated, every
    // destination row whose lower source row is s is complete and is blended and stored.
    constexpr int PF = 4;
    Raw ring[PF];
#pragma unroll
    for(int i = 0; i < PF; ++i)
        ring[i] = load(min(s_lo + i, s_hi));
    int dy = R0;
    Row h_prev{}, h_cur{};
    auto consume = [&](int s, const Raw& w) {
        h_prev = h_cur;
        h_cur = hinterp(w);
        // destination rows complete at source row s: sy1 == s (or both clipped to s)
        while(dy < r_end)
        {
            const int y0 = __builtin_amdgcn_readlane(my_y0, dy - R0);
            const int sy0 = max(0, min(y0, a.sh - 1)), sy1 = max(0, min(y0 + 1, a.sh - 1));
            if(sy1 != s)
                break;
            const uint32_t yc = (uint32_t)__builtin_amdgcn_readlane((int)my_yc, dy - R0);
            const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
            const Row& h0 = sy0 == s ? h_cur : h_prev;
            uint32_t out = 0;
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                // both factors are below 2^24 (weights <= 2048, h <= 32640): 24-bit multiplies are full rate
                // EXACT: h <= 65280, weights <= 256: the products stay below 2^24 as well
                const uint32_t v = EXACT ? (__umul24(b0, h0.h[k]) + __umul24(b1, h_cur.h[k]) + 32768u) >> 16
                                         : ((__umul24(b0, h0.h[k]) >> 16) + (__umul24(b1, h_cur.h[k]) >> 16) + 2) >> 2; // <= 255
                out |= v << (8 * k);
            }
            if(live)
                *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dpitch) = out;
            ++dy;
        }
    };
    for(int s = s_lo; s <= s_hi; s += PF)
    {
#pragma unroll
        for(int i = 0; i < PF; ++i)
        {
            if(s + i <= s_hi)
            {
                const Raw w = ring[i];
                ring[i] = load(min(s + i + PF, s_hi));
                consume(s + i, w);
            }
        }
    }
}

void launch_resize_col(const ResizeColArgs& args, hipStream_t s)
{
    // the float reciprocal reproduces idx / quads exactly: (idx + 0.5) / quads is at least 0.5 / quads away from
    // an integer, far more than the rounding error for idx < 2^22 (api.hip checks frames x quads)
    dim3 grid((args.n_frames * args.quads + 255) / 256, (args.dh + args.R - 1) / args.R);
    if(args.exact)
        hipLaunchKernelGGL(k_resize_col<true>, grid, dim3(256), 0, s, args);
    else
        hipLaunchKernelGGL(k_resize_col<false>, grid, dim3(256), 0, s, args);
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
