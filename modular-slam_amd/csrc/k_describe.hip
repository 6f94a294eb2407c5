// k_describe.hip — intensity-centroid orientation + rotated-BRIEF descriptor + output packing.
//
// Replaces, per keypoint: ic_angle (reference distributed_cv_feature.cpp:543-570, on the UNBLURRED
// level, :977), compute_orb_descriptor scalar branch (:572-629, on the blurred level, :797-801),
// correct_keypoint_scale (:1166-1179) and the output packing of detect() (:1200-1214).
//
// One wave per keypoint (see k_describe).  Descriptor: lane i evaluates pairs i, i+64, i+128, i+192, so
// the four __ballot masks ARE descriptor bytes 0-7, 8-15, 16-23, 24-31 (bit i of byte j = pair 8j+i,
// LSB first, :614-625).  All float arithmetic is individually rounded (no FMA) and follows
// the reference expression order; cvRound = round-half-even, cvFloor = floor.
#include "common.hpp"
#include "../../include/mslam_orb_pattern.h"

namespace mslam
{

// the sampling pattern as floats (x0, y0, x1, y1 per pair): the reference multiplies float table entries
__constant__ __attribute__((aligned(16))) float c_pattern_f[1024] = MSLAM_ORB_PATTERN_INIT;

// cv::fastAtan2 (OpenCV core, atan_f32), degrees in [0, 360)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    constexpr float scale = (float)(180 / 3.1415926535897932384626433832795);
    constexpr float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    constexpr float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float)2.2204460492503131e-16; // (float)DBL_EPSILON
    float a, c, c2;
    if(ax >= ay)
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    else
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fadd_rn(__fmul_rn(p7, c2), p5);
    a = __fadd_rn(__fmul_rn(a, c2), p3);
    a = __fadd_rn(__fmul_rn(a, c2), p1);
    a = __fmul_rn(a, c);
    if(!(ax >= ay))
        a = __fsub_rn(90.f, a);
    if(x < 0)
        a = __fsub_rn(180.f, a);
    if(y < 0)
        a = __fsub_rn(360.f, a);
    return a;
}

// util::cos / util::sin (:456-503): degree-4 polynomial with quadrant folding
__device__ __forceinline__ float poly_cos(float v)
{
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(c1, __fmul_rn(v2, __fadd_rn(c2, __fmul_rn(c3, v2))));
}
__device__ __forceinline__ float util_cos(float v)
{
    constexpr float PI = 3.14159265358979f;
    constexpr float PI_2 = PI / 2.0f, TWO_PI = 2.0f * PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    v = __fsub_rn(v, __fmul_rn((float)(int)floorf(__fmul_rn(v, INV_TWO_PI)), TWO_PI));
    v = (0.0f < v) ? v : -v;
    if(v < PI_2)
        return poly_cos(v);
    else if(v < PI)
        return -poly_cos(__fsub_rn(PI, v));
    else if(v < THREE_PI_2)
        return -poly_cos(__fsub_rn(v, PI));
    else
        return poly_cos(__fsub_rn(TWO_PI, v));
}
__device__ __forceinline__ float util_sin(float v)
{
    constexpr float PI_2 = 3.14159265358979f / 2.0f;
    return util_cos(__fsub_rn(PI_2, v));
}

constexpr int kPatchR = 19;               // sample radius of the rotated pattern (orb_patch_radius_)
constexpr int kPatchRows = 2 * kPatchR + 1; // 39
constexpr int kPatchDw = 11;              // 44 aligned bytes per staged row cover the 39 needed ones
constexpr int kGroup = 64;                // keypoints per workgroup pass: one lane each in the trig phase
constexpr int kBlocksPerFrame = 32;

// 64-lane integer sum with DPP adds (VALU only, no LDS crossbar); the total lands in lane 63
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xE, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xC, true); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true); // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true); // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// A workgroup (4 waves) takes 64 keypoints at a time through three phases:
//  A. moments, one wave per keypoint (16 each): the radius-15 disc is 31 rows x 8 dwords; lane t (+64k)
//     owns dword (row t/8, column group t%8), loads it with one unaligned dword load and folds it into
//     m10/m01 with two signed v_dot4 against per-lane weight bytes (u resp. v inside the disc, 0
//     outside; host-built table).  Pixels are biased by -128 (xor 0x80) to fit i8; the bias cancels
//     exactly because the disc is symmetric (sum of u = sum of v = 0).  Integer sums: order-free.
//  B. one LANE per keypoint: fastAtan2, the f64 degree->radian product, util::cos/sin and the scalar
//     outputs (coordinates, octave, angle, response) — the wave-uniform float work of phase C is
//     thereby done once per keypoint instead of once per lane.
//  C. descriptors, one wave per keypoint: the 39x39 blurred patch is staged in LDS with coalesced
//     aligned dword loads, the 512 rotated sample points are LDS byte gathers, and four ballots are
//     the 32 descriptor bytes.
// Everything a wave needs about "its" keypoint is wave-uniform and kept in SGPRs (readfirstlane).
__global__ __launch_bounds__(256) void k_describe(Geometry g, DescArgs a)
{
    __shared__ uint32_t patch[4][2][kPatchRows * kPatchDw];
    __shared__ int s_m10[kGroup], s_m01[kGroup];
    __shared__ uint32_t s_kp[kGroup];   // packed candidate word
    __shared__ int s_level[kGroup];
    __shared__ float s_ca[kGroup], s_sa[kGroup];

    const size_t frame = blockIdx.y + g.frame0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // levels are concatenated in order (:787-808)
    const uint32_t* sel_cnt = a.sel_cnt + frame * g.n_levels;
    int total = 0;
    for(int l = 0; l < g.n_levels; ++l)
        total += (int)sel_cnt[l];
    if(blockIdx.x == 0 && threadIdx.x == 0)
    {
        a.count[frame] = min(total, a.max_kp);
        if(total > a.max_kp)
            atomicOr(a.flags, kFlagKpOverflow);
    }
    const int n_kp = min(total, a.max_kp);

    // per-lane disc weights (constant across keypoints)
    uint32_t wu[4], wv[4];
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        wu[k] = a.orient_w[lane + 64 * k];
        wv[k] = a.orient_w[256 + lane + 64 * k];
    }
    const uint8_t* pyr = a.pyr + frame * g.slab;
    const uint8_t* blur = a.blur + frame * g.slab;

    for(int base = blockIdx.x * kGroup; base < n_kp; base += gridDim.x * kGroup)
    {
        const int n_here = min(kGroup, n_kp - base);

        // ---- 0. one lane per keypoint: which level, which candidate word
        if(threadIdx.x < n_here)
        {
            int local = base + threadIdx.x, level = 0;
            for(int l = 0; l < g.n_levels; ++l)
            {
                const int c = (int)sel_cnt[l];
                if(local < c)
                {
                    level = l;
                    break;
                }
                local -= c;
            }
            s_kp[threadIdx.x] = a.sel[(frame * g.n_levels + level) * (size_t)a.cand_cap + local];
            s_level[threadIdx.x] = level;
        }
        __syncthreads();

        // ---- A. moments; two keypoints per wave are in flight at a time
        {
            auto fetch = [&](int k, uint32_t (&dw)[4]) {
                const uint32_t p = __builtin_amdgcn_readfirstlane(s_kp[k]);
                const LevelGeom& lv = g.lv[__builtin_amdgcn_readfirstlane(s_level[k])];
                const int px = kp_x(p) + kBorder, py = kp_y(p) + kBorder; // :966-967
                const uint8_t* raw = pyr + lv.offset + (py - 15) * lv.pitch + (px - 15);
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    const int t = lane + 64 * q;
                    dw[q] = t < 31 * 8 ? load_u32_unaligned(raw + (t >> 3) * lv.pitch + 4 * (t & 7)) : 0x80808080u;
                }
            };
            auto reduce = [&](int k, const uint32_t (&dw)[4]) {
                int m10 = 0, m01 = 0;
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    const uint32_t x = dw[q] ^ 0x80808080u;
                    m10 = __builtin_amdgcn_sdot4((int)x, (int)wu[q], m10, false);
                    m01 = __builtin_amdgcn_sdot4((int)x, (int)wv[q], m01, false);
                }
                m10 = wave_sum_dpp(m10);
                m01 = wave_sum_dpp(m01);
                if(lane == 0)
                {
                    s_m10[k] = m10;
                    s_m01[k] = m01;
                }
            };
            for(int k = wave; k < n_here; k += 8)
            {
                uint32_t d0[4], d1[4];
                fetch(k, d0);
                if(k + 4 < n_here)
                    fetch(k + 4, d1);
                reduce(k, d0);
                if(k + 4 < n_here)
                    reduce(k + 4, d1);
            }
        }
        __syncthreads();

        // ---- B. one lane per keypoint: angle, cos/sin, scalar outputs
        if(threadIdx.x < n_here)
        {
            const int k = threadIdx.x;
            const float angle = fast_atan2_deg((float)s_m01[k], (float)s_m10[k]);
            const float rad = (float)((double)angle * 3.14159265358979323846 / 180.0); // :574
            s_ca[k] = util_cos(rad);
            s_sa[k] = util_sin(rad);
            const uint32_t p = s_kp[k];
            const int level = s_level[k];
            const float scale = g.lv[level].scale;
            const float fx = (float)(kp_x(p) + kBorder), fy = (float)(kp_y(p) + kBorder);
            const size_t o = frame * (size_t)a.max_kp + base + k;
            // correct_keypoint_scale (:1166-1179): float multiply, skipped for level 0
            a.xy[2 * o] = level == 0 ? fx : __fmul_rn(fx, scale);
            a.xy[2 * o + 1] = level == 0 ? fy : __fmul_rn(fy, scale);
            a.octave[o] = level;
            a.angle[o] = angle;
            a.response[o] = (float)kp_score(p);
        }
        __syncthreads();

        // ---- C. descriptors; two keypoints per wave are in flight at a time
        {
            auto stage = [&](int k, int buf, int& sh_out) {
                const uint32_t p = __builtin_amdgcn_readfirstlane(s_kp[k]);
                const LevelGeom& lv = g.lv[__builtin_amdgcn_readfirstlane(s_level[k])];
                const int px = kp_x(p) + kBorder, py = kp_y(p) + kBorder;
                const int bx0 = px - kPatchR, sh = bx0 & 3;
                sh_out = sh;
                const uint8_t* bsrc = blur + lv.offset + (py - kPatchR) * lv.pitch + (bx0 - sh);
                const uint32_t* bsrc32 = reinterpret_cast<const uint32_t*>(bsrc); // wave-uniform base, 32-bit lane offsets
                const uint32_t pitch = (uint32_t)lv.pitch;
#pragma unroll
                for(int q = 0; q < 7; ++q)
                {
                    const uint32_t t = (uint32_t)lane + 64u * q;
                    if(t < (uint32_t)(kPatchRows * kPatchDw))
                    {
                        const uint32_t r = (t * 5958u) >> 16; // t / 11 for t < 429
                        const uint32_t c = t - r * kPatchDw;
                        patch[wave][buf][t] = bsrc32[(__umul24(r, pitch) >> 2) + c];
                    }
                }
            };
            auto describe = [&](int k, int buf, int sh) {
                const float ca = s_ca[k], sa = s_sa[k];
                const uint8_t* bc =
                    reinterpret_cast<const uint8_t*>(patch[wave][buf]) + kPatchR * (kPatchDw * 4) + kPatchR + sh; // centre
                unsigned long long bits[4];
#pragma unroll
                for(int t = 0; t < 4; ++t)
                {
                    const float4 q = reinterpret_cast<const float4*>(c_pattern_f)[lane + 64 * t];
                    // GET_VALUE (:603-605): row = cvRound(x*sin + y*cos), col = cvRound(x*cos - y*sin)
                    const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(q.x, sa), __fmul_rn(q.y, ca)));
                    const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(q.x, ca), __fmul_rn(q.y, sa)));
                    const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(q.z, sa), __fmul_rn(q.w, ca)));
                    const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(q.z, ca), __fmul_rn(q.w, sa)));
                    const int v0 = bc[__mul24(r0, kPatchDw * 4) + c0];
                    const int v1 = bc[__mul24(r1, kPatchDw * 4) + c1];
                    bits[t] = __ballot(v0 < v1);
                }
                if(lane < 4)
                {
                    unsigned long long w = bits[0];
                    w = lane == 1 ? bits[1] : w;
                    w = lane == 2 ? bits[2] : w;
                    w = lane == 3 ? bits[3] : w;
                    reinterpret_cast<unsigned long long*>(a.desc + (frame * (size_t)a.max_kp + base + k) * 32)[lane] = w;
                }
            };
            for(int k = wave; k < n_here; k += 8)
            {
                int sh0 = 0, sh1 = 0;
                const bool two = k + 4 < n_here;
                __builtin_amdgcn_wave_barrier(); // the previous pair's gathers are done before the patches are overwritten
                stage(k, 0, sh0);
                if(two)
                    stage(k + 4, 1, sh1);
                __builtin_amdgcn_wave_barrier();
                describe(k, 0, sh0);
                if(two)
                    describe(k + 4, 1, sh1);
            }
        }
        __syncthreads(); // LDS keypoint slots are reused by the next group
    }
}

void launch_describe(const Geometry& g, const DescArgs& a, int frame0, int n_frames, hipStream_t s)
{
    dim3 grid(kBlocksPerFrame, n_frames);
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, s, gg, a);
}

} // namespace mslam
