// k_blur.hip — 7x7 sigma=2 Gaussian blur of every pyramid level, bit-exact fixed point.
//
// Replaces the per-level clone + cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) of the reference
// (distributed_cv_feature.cpp:797-798).  OpenCV's CV_8U path is integer-only: taps in unsigned 8.8
// fixed point {18,34,48,56,48,34,18} (sum 256), horizontal pass u8*8.8 -> 8.8 (exact, <= 65280),
// vertical pass 8.8*8.8 -> 16.16 rounded by (v + 2^15) >> 16.  The taps come from the host, which
// derives them with OpenCV's error-diffusion rule (api.hip: gaussian_taps_fixed).
//
// Since round 3 the blur of a level normally happens inside the kernel that produces the level (k_level.hip).  This
// stand-alone kernel is the generic form: it serves the levels those kernels cannot take (level-0 widths that are not
// a multiple of 4, scale factors beyond k_resize_col's 12-byte window, batches beyond 32-bit offsets) and the
// MSLAM_HIP_FUSED_LEVELS=n test switch.
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

// k_blur2:
//   * one lane owns a strip of 4 columns x 32 rows and walks down it; per source row it loads three aligned dwords
//     (12 bytes around its 4 columns) straight from global memory — no LDS, no barriers;
//   * every WAVE is level-uniform (host table of wave descriptors; its lanes are 64 consecutive (band, strip) items of
//     the level): row pitch, level base and the row loop are scalar, so an interior source row costs no vector address
//     arithmetic — the three loads take the "SGPR row base + per-lane offset" form and the row base advances on the
//     scalar unit.  (Waves of 32 / 16 / 8 strips x 2 / 4 / 8 bands waste fewer lanes on some widths but touch more rows
//     per load: measured 0.63 vs 0.50 ms per 1000 frames, so the table builder only emits the flat form.)
//   * bands are anchored so that the last one ENDS at the last row (it overlaps its predecessor instead of hanging
//     over the edge; the overlapping rows are written twice with identical bytes): no store is masked, and
//     REFLECT_101 in y can only happen in the first three and the last three source rows of a strip — those six
//     rows keep the per-lane index arithmetic, the other 32 do not have any;
//   * horizontal 7-tap: two or three v_dot4_u32_u8 per pixel on the three window dwords AS LOADED, against taps
//     shifted to the pixel's position (ten scalar constants) — no byte alignment instructions;
//   * vertical 7-tap: three v_dot2_u32_u16 on packed (row, row+1) pairs + one mad per pixel, the rounding constant
//     folded into the first accumulate; the row loop is fully unrolled so the sliding window lives in registers;
//   * REFLECT_101 in x: the left edge is one v_perm_b32 (identity for other lanes); the right-edge selectors run
//     only in waves that contain a lane whose window reaches beyond the last column (wave-uniform branch).
// Levels lower than 38 rows (tiny cv::ORB-mode pyramids) take the same code with every row on the per-lane path.
// Measured (round 3): 40 % fewer vector instructions than the round-2 kernel at the same 0.50 ms per 1000 frames —
// the stand-alone blur is bound by its memory side (it re-reads every level plane), which is why it was fused.
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

__global__ __launch_bounds__(256) void k_blur2(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, Geometry g, BlurK k,
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
    const int n_bands = (h + kBlurRows - 1) / kBlurRows;
    // lanes = consecutive (band, strip) items of the level, row-major: item / bsx by a float reciprocal (exact for items < 2^22)
    const int item = d.item0 + lane;
    const int band = (int)(((float)item + 0.5f) * d.inv_bsx);
    const int strip = item - band * lv.bsx;
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
static Taps g_taps = {{18, 34, 48, 56, 48, 34, 18}};
void set_blur_taps(const int* t)
{
    for(int i = 0; i < 7; ++i)
        g_taps.t[i] = t[i];
}

BlurK make_blur_k()
{
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
    return k;
}

// the wave descriptors of one frame (k_blur2): per level the strips-per-row / bands-per-wave split that idles the fewest lanes
void build_blur_waves(const Geometry& g, int first_level, std::vector<BlurWave>& out)
{
    out.clear();
    for(int l = first_level; l < g.n_levels; ++l)
    {
        const LevelGeom& lv = g.lv[l];
        const int n_bands = (lv.h + kBlurRows - 1) / kBlurRows;
        const int n_items = lv.bsx * n_bands;
        for(int i0 = 0; i0 < n_items; i0 += 64)
        {
            BlurWave w{};
            w.level = l;
            w.item0 = i0;
            w.generic = (int16_t)(lv.h < kBlurRows + 6 ? 1 : 0);
            w.inv_bsx = 1.0f / (float)lv.bsx;
            out.push_back(w);
        }
    }
}

void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, const BlurWave* d_waves, int wpf, int frame0, int n_frames,
                 hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    if(!d_waves || wpf <= 0)
        return;
    const BlurK k = make_blur_k();
    const int bpf = (wpf + 3) / 4;
    const unsigned grid = (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)bpf;
    hipLaunchKernelGGL(k_blur2, dim3(grid), dim3(256), 0, s, d_pyr, d_blur, gg, k, d_waves, n_frames, wpf, bpf);
}

} // namespace mslam
