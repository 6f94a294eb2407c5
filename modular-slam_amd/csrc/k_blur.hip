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
#include <cstdlib>
#include <vector>

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


// ---------------------------------------------------------------------------------------------------------------------
// Round-3 form of the same filter (k_blur2): the arithmetic is unchanged, the instructions around it are gone.
//   * every WAVE is level-uniform (host table of wave descriptors: level, first strip, first band, strips per band
//     row): row pitch, level base and the row loop are scalar, so an interior source row costs no vector address
//     arithmetic at all — the three loads take the "SGPR row base + per-lane offset" form and the row base advances
//     on the scalar unit.  A wave covers 64 / 32 / 16 / 8 strips of 1 / 2 / 4 / 8 consecutive bands, whichever wastes
//     the fewest lanes on the level's width (640 px = 160 strips = 5 x 32, 444 px = 111 strips = 7 x 16, ...);
//   * bands are anchored so that the last one ENDS at the last row (it overlaps its predecessor instead of hanging
//     over the edge; the overlapping rows are written twice with identical bytes): no store is masked, and
//     REFLECT_101 in y can only happen in the first three and the last three source rows of a strip — those six
//     rows keep the per-lane index arithmetic, the other 32 do not have any;
//   * the horizontal 7-tap is two or three v_dot4_u32_u8 per pixel on the three window dwords AS LOADED, against
//     taps shifted to the pixel's position (ten scalar constants) — the six v_alignbyte_b32 per row are gone;
//   * REFLECT_101 in x: the left edge is one v_perm_b32 (identity for other lanes); the right-edge selectors run
//     only in waves that contain a lane whose window reaches beyond the last column (wave-uniform branch).
// Levels lower than 38 rows (tiny cv::ORB-mode pyramids) take the same code with every row on the per-lane path.
struct BlurK
{
    uint32_t ta[3], tb[4], tc[4]; // horizontal taps positioned for output pixel j on window dwords A / B / C (tc[0] unused)
    uint32_t t01, t23, t45, t6;   // vertical taps as u16 pairs
};

template <bool GENERIC, bool RIGHT>
__device__ __forceinline__ void blur_rows(const uint8_t* __restrict__ src_lv, uint8_t* __restrict__ dst_lv, const BlurK& k, int w, int h,
                                          uint32_t pitch, int x0, int y0)
{
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
    const uint32_t offA = (uint32_t)max(x0 - 4, 0), offB = (uint32_t)x0, offC = (uint32_t)min(x0 + 4, (int)pitch - 4);
    // per-lane part of the interior rows' addresses: the strip's first source row is y0 - 3
    const uint32_t lane_row = __umul24((uint32_t)max(y0, 0), pitch);
    const uint32_t vA = lane_row + offA, vB = lane_row + offB, vC = lane_row + offC;

    uint32_t pr[6][4]; // pr[m % 6][j] = (h[m][j], h[m+1][j]) packed, for the last six row pairs
    uint32_t hprev[4] = {0, 0, 0, 0};
#pragma unroll
    for(int i = 0; i < kBlurRows + 6; ++i)
    {
        uint32_t A, B, C;
        if(GENERIC || i < 3 || i >= kBlurRows + 3)
        {
            // REFLECT_101 of the row index without compare/select pairs: |y|, then min(y, 2 (h - 1) - y)
            const int ya = max(y0 - 3 + i, -(y0 - 3 + i));
            const int yy = GENERIC ? max(min(ya, 2 * (h - 1) - ya), 0) : min(ya, 2 * (h - 1) - ya);
            const uint32_t ro = __umul24((uint32_t)yy, pitch);
            A = *reinterpret_cast<const uint32_t*>(src_lv + (ro + offA));
            B = *reinterpret_cast<const uint32_t*>(src_lv + (ro + offB));
            C = *reinterpret_cast<const uint32_t*>(src_lv + (ro + offC));
        }
        else
        {
            const uint8_t* rowp = src_lv + (ptrdiff_t)(i - 3) * (ptrdiff_t)pitch; // wave-uniform
            A = *reinterpret_cast<const uint32_t*>(rowp + vA);
            B = *reinterpret_cast<const uint32_t*>(rowp + vB);
            C = *reinterpret_cast<const uint32_t*>(rowp + vC);
        }
        const uint32_t A2 = __builtin_amdgcn_perm(B, A, selA);
        uint32_t B2 = B, C2 = C;
        if(RIGHT)
        {
            B2 = __builtin_amdgcn_perm(B, A, selB);
            const uint32_t T = __builtin_amdgcn_perm(B, A, selT);
            const uint32_t U = __builtin_amdgcn_perm(C, B, selU);
            C2 = (T & maskT) | (U & ~maskT);
        }
        // horizontal pass: output column j uses window bytes j+1 .. j+7 = the dwords as they are against shifted taps
        uint32_t hv[4];
        hv[0] = __builtin_amdgcn_udot4(A2, k.ta[0], __builtin_amdgcn_udot4(B2, k.tb[0], 0u, false), false);
        hv[1] = __builtin_amdgcn_udot4(A2, k.ta[1], __builtin_amdgcn_udot4(B2, k.tb[1], __builtin_amdgcn_udot4(C2, k.tc[1], 0u, false), false), false);
        hv[2] = __builtin_amdgcn_udot4(A2, k.ta[2], __builtin_amdgcn_udot4(B2, k.tb[2], __builtin_amdgcn_udot4(C2, k.tc[2], 0u, false), false), false);
        hv[3] = __builtin_amdgcn_udot4(B2, k.tb[3], __builtin_amdgcn_udot4(C2, k.tc[3], 0u, false), false);
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
                acc[j] = dot2u(pr[(i - 6) % 6][j], k.t01, 32768u);
                acc[j] = dot2u(pr[(i - 4) % 6][j], k.t23, acc[j]);
                acc[j] = dot2u(pr[(i - 2) % 6][j], k.t45, acc[j]);
                acc[j] += __umul24(hv[j], k.t6); // hv <= 65280, tap <= 255: the 24-bit multiply-add is full rate
            }
            // the result (acc >> 16, at most 255) is byte 2 of each accumulator: two v_perm gather the four bytes
            const uint32_t out = __builtin_amdgcn_perm(acc[1], acc[0], 0x0C0C0602u) |
                                 __builtin_amdgcn_perm(acc[3], acc[2], 0x06020C0Cu);
            if(GENERIC)
            {
                const int o = y0 + i - 6;
                if(o < h)
                    *reinterpret_cast<uint32_t*>(dst_lv + (__umul24((uint32_t)o, pitch) + offB)) = out;
            }
            else
            {
                uint8_t* rowp = dst_lv + (ptrdiff_t)(i - 6) * (ptrdiff_t)pitch; // wave-uniform
                *reinterpret_cast<uint32_t*>(rowp + vB) = out;
            }
        }
#pragma unroll
        for(int j = 0; j < 4; ++j)
            hprev[j] = hv[j];
    }
}

template <int OCC>
__device__ __forceinline__ void blur2_body(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, const Geometry& g, const BlurK& k,
                                               const BlurWave* __restrict__ waves, int n_frames, int wpf, int bpf)
{
    // XCD-aware mapping (as k_fast_cells): the strips of one frame share rows (3-row halos) and 128-byte lines, so all
    // workgroups of a frame get ids with the same (id & 7) and meet in one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int f_local = (slot / bpf) * 8 + xcd;
    if(f_local >= n_frames)
        return;
    const int wid = __builtin_amdgcn_readfirstlane((slot % bpf) * 4 + (int)(threadIdx.x >> 6));
    if(wid >= wpf)
        return;
    const BlurWave d = waves[wid];
    const int lane = threadIdx.x & 63;
    const LevelGeom& lv = g.lv[d.level];
    const int w = lv.w, h = lv.h;
    const uint32_t pitch = (uint32_t)lv.pitch;
    int strip, band;
    const int n_bands = (h + kBlurRows - 1) / kBlurRows;
    if(d.lg_spg == 0)
    {
        // lanes = consecutive (band, strip) items of the level, row-major: item / bsx by a float reciprocal (exact for items < 2^22)
        const int item = d.strip0 + lane;
        band = (int)(((float)item + 0.5f) * d.inv_bsx);
        strip = item - band * lv.bsx;
    }
    else
    {
        strip = d.strip0 + (lane & ((1 << d.lg_spg) - 1));
        band = d.band0 + (lane >> d.lg_spg);
    }
    if(strip >= lv.bsx || band >= n_bands)
        return;
    const int x0 = strip * 4;
    const size_t frame = (size_t)f_local + g.frame0;
    const uint8_t* src_lv = pyr + frame * g.slab + (uint32_t)lv.offset;
    uint8_t* dst_lv = blur + frame * g.slab + (uint32_t)lv.offset;
    // does any lane of this wave reach beyond the last column (x0 + 7 >= w)?  wave-uniform
    const bool right_any = __ballot(x0 + 7 >= w) != 0ull;
    if(d.generic)
        blur_rows<true, true>(src_lv, dst_lv, k, w, h, pitch, x0, band * kBlurRows);
    else if(right_any)
        blur_rows<false, true>(src_lv, dst_lv, k, w, h, pitch, x0, min(band * kBlurRows, h - kBlurRows));
    else
        blur_rows<false, false>(src_lv, dst_lv, k, w, h, pitch, x0, min(band * kBlurRows, h - kBlurRows));
}
__global__ __launch_bounds__(256) void k_blur2(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, Geometry g, BlurK k,
                                               const BlurWave* __restrict__ waves, int n_frames, int wpf, int bpf)
{
    blur2_body<0>(pyr, blur, g, k, waves, n_frames, wpf, bpf);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_blur2_occ8(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, Geometry g, BlurK k,
                                               const BlurWave* __restrict__ waves, int n_frames, int wpf, int bpf)
{
    blur2_body<8>(pyr, blur, g, k, waves, n_frames, wpf, bpf);
}

static Taps g_taps = {{18, 34, 48, 56, 48, 34, 18}};
void set_blur_taps(const int* t)
{
    for(int i = 0; i < 7; ++i)
        g_taps.t[i] = t[i];
}

// the wave descriptors of one frame (k_blur2): per level the strips-per-row / bands-per-wave split that idles the fewest lanes
void build_blur_waves(const Geometry& g, std::vector<BlurWave>& out)
{
    out.clear();
    for(int l = 0; l < g.n_levels; ++l)
    {
        const LevelGeom& lv = g.lv[l];
        const int n_bands = (lv.h + kBlurRows - 1) / kBlurRows;
        int best_lg = 6;
        long best = -1;
        static const int force_lg = getenv("MSLAM_BLUR_LG") ? atoi(getenv("MSLAM_BLUR_LG")) : 0;
        if(force_lg == 0)
        {
            const int n_items = lv.bsx * n_bands;
            for(int i0 = 0; i0 < n_items; i0 += 64)
            {
                BlurWave w{};
                w.level = l;
                w.strip0 = i0; // first item
                w.band0 = 0;
                w.lg_spg = 0;
                w.generic = (int16_t)(lv.h < kBlurRows + 6 ? 1 : 0);
                w.inv_bsx = 1.0f / (float)lv.bsx;
                out.push_back(w);
            }
            continue;
        }
        for(int lg = 6; lg >= 3; --lg)
        {
            if(force_lg && lg != force_lg)
                continue;
            const int spg = 1 << lg, gb = 64 >> lg;
            const long lanes = 64L * ((lv.bsx + spg - 1) / spg) * ((n_bands + gb - 1) / gb);
            if(best < 0 || lanes < best)
                best = lanes, best_lg = lg;
        }
        const int spg = 1 << best_lg, gb = 64 >> best_lg;
        for(int b0 = 0; b0 < n_bands; b0 += gb)
            for(int s0 = 0; s0 < lv.bsx; s0 += spg)
            {
                BlurWave w{};
                w.level = l;
                w.strip0 = s0;
                w.band0 = b0;
                w.lg_spg = (int16_t)best_lg;
                w.generic = (int16_t)(lv.h < kBlurRows + 6 ? 1 : 0);
                out.push_back(w);
            }
    }
}

void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, const BlurWave* d_waves, int wpf, int frame0, int n_frames,
                 hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    static const bool old_form = getenv("MSLAM_BLUR_OLD") != nullptr; // round-2 kernel, kept for A/B timing
    if(old_form || !d_waves)
    {
        const int bpf = (g.n_tiles + 255) / 256;
        const unsigned grid = (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)bpf;
        hipLaunchKernelGGL(k_blur, dim3(grid), dim3(256), 0, s, d_pyr, d_blur, gg, g_taps, n_frames, bpf);
        return;
    }
    const int* t = g_taps.t;
    BlurK k{};
    // window byte m (0..11 over dwords A, B, C) of output pixel j carries tap m - j - 1
    for(int j = 0; j < 4; ++j)
        for(int m = 0; m < 12; ++m)
        {
            const int tap = m - j - 1;
            if(tap < 0 || tap > 6)
                continue;
            const uint32_t v = (uint32_t)t[tap] << (8 * (m & 3));
            if(m < 4)
                k.ta[j < 3 ? j : 0] |= v; // j == 3 never touches dword A
            else if(m < 8)
                k.tb[j] |= v;
            else
                k.tc[j] |= v;
        }
    k.t01 = (uint32_t)t[0] | ((uint32_t)t[1] << 16);
    k.t23 = (uint32_t)t[2] | ((uint32_t)t[3] << 16);
    k.t45 = (uint32_t)t[4] | ((uint32_t)t[5] << 16);
    k.t6 = (uint32_t)t[6];
    const int bpf = (wpf + 3) / 4;
    const unsigned grid = (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)bpf;
    static const bool occ8 = getenv("MSLAM_BLUR_OCC8") != nullptr;
    if(occ8)
        hipLaunchKernelGGL(k_blur2_occ8, dim3(grid), dim3(256), 0, s, d_pyr, d_blur, gg, k, d_waves, n_frames, wpf, bpf);
    else
        hipLaunchKernelGGL(k_blur2, dim3(grid), dim3(256), 0, s, d_pyr, d_blur, gg, k, d_waves, n_frames, wpf, bpf);
}

} // namespace mslam
