// k_describe.hip — intensity-centroid orientation + rotated-BRIEF descriptor + output packing.
//
// Replaces, per keypoint: ic_angle (reference distributed_cv_feature.cpp:543-570, on the UNBLURRED
// level, :977), compute_orb_descriptor scalar branch (:572-629, on the blurred level, :797-801),
// correct_keypoint_scale (:1166-1179) and the output packing of detect() (:1200-1214).
//
// One wave per keypoint.  Orientation: lanes 0..30 each own one column u of the radius-15 disc and
// walk its rows, so every step of the walk reads 31 consecutive bytes; the integer moments are
// reduced across the wave (exact, order-free).  Descriptor: lane i evaluates pairs i, i+64, i+128,
// i+192, so the four __ballot masks ARE descriptor bytes 0-7, 8-15, 16-23, 24-31 (bit i of byte j =
// pair 8j+i, LSB first, :614-625).  All float arithmetic is individually rounded (no FMA) and follows
// the reference expression order; cvRound = round-half-even, cvFloor = floor.
#include "common.hpp"
#include "../../include/mslam_orb_pattern.h"

namespace mslam
{

__constant__ int8_t c_pattern[1024] = MSLAM_ORB_PATTERN_INIT;
// u_max_ of orb_impl's constructor (:522-541) for half patch 15; api.hip recomputes and checks it.
__constant__ int8_t c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

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

__global__ __launch_bounds__(256) void k_describe(Geometry g, DescArgs a)
{
    const size_t frame = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;

    // locate keypoint i: levels are concatenated in order (:787-808)
    const uint32_t* sel_cnt = a.sel_cnt + frame * g.n_levels;
    int level = -1, local = 0, total = 0;
    for(int l = 0; l < g.n_levels; ++l)
    {
        const int c = (int)sel_cnt[l];
        if(level < 0 && i < total + c)
        {
            level = l;
            local = i - total;
        }
        total += c;
    }
    if(i == 0 && lane == 0)
    {
        a.count[frame] = min(total, a.max_kp);
        if(total > a.max_kp)
            atomicOr(a.flags, kFlagKpOverflow);
    }
    if(level < 0 || i >= a.max_kp)
        return;

    const LevelGeom& lv = g.lv[level];
    const uint32_t p = a.sel[(frame * g.n_levels + level) * (size_t)a.cand_cap + local];
    const int px = kp_x(p) + kBorder, py = kp_y(p) + kBorder; // :966-967
    const int pitch = lv.pitch;

    // ---- orientation on the unblurred level
    const uint8_t* center = a.pyr + frame * g.slab + lv.offset + (size_t)py * pitch + px;
    int m10 = 0, m01 = 0;
    if(lane < 31)
    {
        const int u = lane - 15;
        const int dv = c_umax[u < 0 ? -u : u];
        int col = 0;
        for(int v = -dv; v <= dv; ++v)
        {
            const int I = center[v * pitch + u];
            col += I;
            m01 += v * I;
        }
        m10 = u * col;
    }
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
    {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- rotated BRIEF on the blurred level
    const float rad = (float)((double)angle * 3.14159265358979323846 / 180.0); // :574
    const float ca = util_cos(rad), sa = util_sin(rad);
    const uint8_t* bc = a.blur + frame * g.slab + lv.offset + (size_t)py * pitch + px;
    unsigned long long bits[4];
#pragma unroll
    for(int t = 0; t < 4; ++t)
    {
        const int pair = lane + 64 * t;
        const char4 q = reinterpret_cast<const char4*>(c_pattern)[pair];
        const float x0 = (float)q.x, y0 = (float)q.y, x1 = (float)q.z, y1 = (float)q.w;
        // GET_VALUE (:603-605): row = cvRound(x*sin + y*cos), col = cvRound(x*cos - y*sin)
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, sa), __fmul_rn(y0, ca)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sa)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, sa), __fmul_rn(y1, ca)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sa)));
        const int v0 = bc[r0 * pitch + c0];
        const int v1 = bc[r1 * pitch + c1];
        bits[t] = __ballot(v0 < v1);
    }

    const size_t o = frame * (size_t)a.max_kp + i;
    if(lane < 4)
    {
        unsigned long long w = bits[0];
        w = lane == 1 ? bits[1] : w;
        w = lane == 2 ? bits[2] : w;
        w = lane == 3 ? bits[3] : w;
        reinterpret_cast<unsigned long long*>(a.desc + o * 32)[lane] = w;
    }
    if(lane == 0)
    {
        // correct_keypoint_scale (:1166-1179): float multiply, skipped for level 0
        const float fx = (float)px, fy = (float)py;
        a.xy[2 * o] = level == 0 ? fx : __fmul_rn(fx, lv.scale);
        a.xy[2 * o + 1] = level == 0 ? fy : __fmul_rn(fy, lv.scale);
        a.octave[o] = level;
        a.angle[o] = angle;
        a.response[o] = (float)kp_score(p);
    }
}

void launch_describe(const Geometry& g, const DescArgs& a, int n_frames, hipStream_t s)
{
    dim3 grid((a.max_kp + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_describe, grid, dim3(256), 0, s, g, a);
}

} // namespace mslam
