// k_blur.hip — 7x7 sigma=2 Gaussian blur of every pyramid level, bit-exact fixed point.
//
// Replaces the per-level clone + cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) of the reference
// (distributed_cv_feature.cpp:797-798).  OpenCV's CV_8U path is integer-only: taps in unsigned 8.8
// fixed point {18,34,48,56,48,34,18} (sum 256), horizontal pass u8*8.8 -> 8.8 (exact, <= 65280),
// vertical pass 8.8*8.8 -> 16.16 rounded by (v + 2^15) >> 16.  The taps come from the host, which
// derives them with OpenCV's error-diffusion rule (api.hip: gaussian_taps_fixed).
//
// The kernel is VALU-bound, so it is built on the packed dot-product instructions:
//   * one lane owns a strip of 4 columns x 32 rows and walks down it; per source row it loads three
//     aligned dwords (12 bytes around its 4 columns) straight from global memory (rows are read by
//     neighbouring lanes too, so L1 serves the overlap) — no LDS, no barriers;
//   * horizontal 7-tap = two v_dot4_u32_u8 per pixel on byte windows cut with v_alignbyte_b32;
//   * vertical 7-tap   = three v_dot2_u32_u16 on packed (row, row+1) pairs + one mad per pixel, with
//     the rounding constant folded into the first accumulate;
//   * REFLECT_101 at the left/right image edges is applied to the three dwords with four v_perm_b32
//     whose selectors are computed once per lane (identity for interior lanes), so there is no
//     divergent edge path; top/bottom reflection is an index computation per row.
// The row loop is fully unrolled so the sliding window lives in registers without moves.
#include "common.hpp"

namespace mslam
{

struct Taps
{
    int t[7];
};

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t dot2u(uint32_t pair, uint32_t taps, uint32_t acc)
{
    ushort2v a, b;
    a.x = (unsigned short)(pair & 0xFFFF);
    a.y = (unsigned short)(pair >> 16);
    b.x = (unsigned short)(taps & 0xFFFF);
    b.y = (unsigned short)(taps >> 16);
    return __builtin_amdgcn_udot2(a, b, acc, false);
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, Geometry g,
                                              Taps taps, int n_frames, int blocks_per_frame)
{
    // XCD-aware mapping (as k_fast_cells): the strips of one frame share rows (3-row halos) and 128-byte lines, so all
    // workgroups of a frame get ids with the same (id & 7) and meet in one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int f_local = (slot / blocks_per_frame) * 8 + xcd;
    if(f_local >= n_frames)
        return;
    const int idx = (slot % blocks_per_frame) * 256 + threadIdx.x;
    if(idx >= g.n_tiles)
        return;
    int level = 0;
#pragma unroll 1
    for(int l = 1; l < g.n_levels; ++l)
        if(idx >= g.lv[l].tile_base)
            level = l;
    const LevelGeom& lv = g.lv[level];
    const int sidx = idx - lv.tile_base;
    const int band = sidx / lv.bsx;
    const int x0 = (sidx - band * lv.bsx) * 4;
    const int y0 = band * kBlurRows;
    const int w = lv.w, h = lv.h, pitch = lv.pitch;
    const size_t frame = (size_t)f_local + g.frame0;
    // wave-uniform frame bases + 32-bit per-lane offsets (the level differs between lanes): loads and stores take the
    // SGPR-base + VGPR-offset form and no 64-bit address arithmetic is needed per row
    const uint8_t* src = pyr + frame * g.slab;
    uint8_t* dst = blur + frame * g.slab;
    const uint32_t lofs = (uint32_t)lv.offset;

    // ---- per-lane REFLECT_101 selectors for the 12-byte window [x0-4, x0+8)
    uint32_t selA = 0, selB = 0, selT = 0, selU = 0, maskT = 0;
#pragma unroll
    for(int i = 0; i < 12; ++i)
    {
        int col = x0 - 4 + i;
        if(col < 0)
            col = -col;
        else if(col >= w)
            col = 2 * (w - 1) - col;
        int s = col - (x0 - 4); // source byte index inside the unreflected window
        if(s < 0 || s > 11)
            s = i; // only feeds outputs that are discarded
        const int b = i & 3;
        if(i < 4)
            selA |= (uint32_t)(s <= 7 ? s : i) << (8 * b);
        else if(i < 8)
            selB |= (uint32_t)(s <= 7 ? s : i) << (8 * b);
        else if(s < 4)
        {
            selT |= (uint32_t)s << (8 * b);
            maskT |= 0xFFu << (8 * b);
        }
        else
            selU |= (uint32_t)(s - 4) << (8 * b);
    }
    const int offA = max(x0 - 4, 0), offC = min(x0 + 4, pitch - 4);

    const uint32_t t0123 = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 8) | ((uint32_t)taps.t[2] << 16) | ((uint32_t)taps.t[3] << 24);
    const uint32_t t456 = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 8) | ((uint32_t)taps.t[6] << 16);
    const uint32_t t01 = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 16);
    const uint32_t t23 = (uint32_t)taps.t[2] | ((uint32_t)taps.t[3] << 16);
    const uint32_t t45 = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 16);
    const uint32_t t6 = (uint32_t)taps.t[6];

    uint32_t pr[6][4]; // pr[m % 6][j] = (h[m][j], h[m+1][j]) packed, for the last six row pairs
    uint32_t hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for(int i = 0; i < kBlurRows + 6; ++i)
    {
        // REFLECT_101 of the row index without compare/select pairs: |y|, then min(y, 2 (h - 1) - y)
        const int ya = max(y0 - 3 + i, -(y0 - 3 + i));
        const int yy = min(ya, 2 * (h - 1) - ya);
        // 32-bit offsets from the (wave-uniform) level base: one 24-bit multiply per row instead of 64-bit
        // address arithmetic per load
        const uint32_t ro = lofs + __umul24((uint32_t)yy, (uint32_t)pitch);
        const uint32_t A = *reinterpret_cast<const uint32_t*>(src + (ro + (uint32_t)offA));
        const uint32_t B = *reinterpret_cast<const uint32_t*>(src + (ro + (uint32_t)x0));
        const uint32_t C = *reinterpret_cast<const uint32_t*>(src + (ro + (uint32_t)offC));
        const uint32_t A2 = __builtin_amdgcn_perm(B, A, selA);
        const uint32_t B2 = __builtin_amdgcn_perm(B, A, selB);
        const uint32_t T = __builtin_amdgcn_perm(B, A, selT);
        const uint32_t U = __builtin_amdgcn_perm(C, B, selU);
        const uint32_t C2 = (T & maskT) | (U & ~maskT);
        // horizontal pass: output column j uses window bytes j+1 .. j+7
        uint32_t hv[4];
        hv[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B2, A2, 1), t0123,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C2, B2, 1), t456, 0u, false), false);
        hv[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B2, A2, 2), t0123,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C2, B2, 2), t456, 0u, false), false);
        hv[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B2, A2, 3), t0123,
                                       __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C2, B2, 3), t456, 0u, false), false);
        hv[3] = __builtin_amdgcn_udot4(B2, t0123, __builtin_amdgcn_udot4(C2, t456, 0u, false), false);
        if(i >= 1)
        {
#pragma unroll
            for(int j = 0; j < 4; ++j)
                pr[(i - 1) % 6][j] = hprev[j] | (hv[j] << 16); // pair (row i-1, row i)
        }
        if(i >= 6)
        {
            // vertical pass for output row o = y0 + i - 6: source rows i-6 .. i
            uint32_t acc[4];
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                acc[j] = dot2u(pr[(i - 6) % 6][j], t01, 32768u);
                acc[j] = dot2u(pr[(i - 4) % 6][j], t23, acc[j]);
                acc[j] = dot2u(pr[(i - 2) % 6][j], t45, acc[j]);
                acc[j] += __umul24(hv[j], t6); // hv <= 65280, tap <= 255: the 24-bit multiply-add is full rate
            }
            // the result (acc >> 16, at most 255) is byte 2 of each accumulator: two v_perm gather the four bytes
            const uint32_t out = __builtin_amdgcn_perm(acc[1], acc[0], 0x0C0C0602u) |
                                 __builtin_amdgcn_perm(acc[3], acc[2], 0x06020C0Cu);
            const int o = y0 + i - 6;
            if(o < h)
                *reinterpret_cast<uint32_t*>(dst + (lofs + __umul24((uint32_t)o, (uint32_t)pitch) + (uint32_t)x0)) = out;
        }
#pragma unroll
        for(int j = 0; j < 4; ++j)
            hprev[j] = hv[j];
    }
}

static Taps g_taps = {{18, 34, 48, 56, 48, 34, 18}};
void set_blur_taps(const int* t)
{
    for(int i = 0; i < 7; ++i)
        g_taps.t[i] = t[i];
}

void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, int frame0, int n_frames, hipStream_t s)
{
    const int bpf = (g.n_tiles + 255) / 256;
    const unsigned grid = (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)bpf;
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_blur, dim3(grid), dim3(256), 0, s, d_pyr, d_blur, gg, g_taps, n_frames, bpf);
}

} // namespace mslam
