// k_blur.hip — 7x7 sigma=2 Gaussian blur of every pyramid level, bit-exact fixed point.
//
// Replaces the per-level clone + cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) of the reference
// (distributed_cv_feature.cpp:797-798).  OpenCV's CV_8U path is integer-only: taps in unsigned 8.8
// fixed point {18,34,48,56,48,34,18} (sum 256), horizontal pass u8*8.8 -> 8.8, vertical pass
// 8.8*8.8 -> 16.16 rounded by (v + 2^15) >> 16.  The taps are passed in by the host, which derives
// them with OpenCV's error-diffusion rule (api.hip: gaussian_taps_fixed).
//
// One 256-thread workgroup per 64x32 output tile: the 70x38 source window (REFLECT_101 at the image
// edges) is staged in LDS, the horizontal pass writes a 16-bit LDS plane, the vertical pass streams
// out one coalesced 64-byte row segment per wave.
#include "common.hpp"

namespace mslam
{

constexpr int kBW = 64, kBH = 32;
constexpr int kRawP = 72;

__device__ __forceinline__ int reflect101(int p, int len)
{
    if(len == 1)
        return 0;
    while(p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct Taps
{
    int t[7];
};

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, Geometry g,
                                              const BlurTile* __restrict__ tiles, Taps taps)
{
    __shared__ uint8_t raw[(kBH + 6) * kRawP];
    __shared__ uint16_t hb[(kBH + 6) * kBW];

    const BlurTile t = tiles[blockIdx.x];
    const size_t frame = blockIdx.y;
    const LevelGeom& lv = g.lv[t.level];
    const uint8_t* src = pyr + frame * g.slab + lv.offset;
    uint8_t* dst = blur + frame * g.slab + lv.offset;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for(int r = wave; r < kBH + 6; r += 4)
    {
        const int sy = reflect101(t.y0 - 3 + r, lv.h);
        const uint8_t* s = src + (size_t)sy * lv.pitch;
        raw[r * kRawP + lane] = s[reflect101(t.x0 - 3 + lane, lv.w)];
        if(lane < 6)
            raw[r * kRawP + 64 + lane] = s[reflect101(t.x0 - 3 + 64 + lane, lv.w)];
    }
    __syncthreads();
    for(int idx = tid; idx < (kBH + 6) * kBW; idx += 256)
    {
        const int r = idx >> 6, c = idx & 63;
        const uint8_t* p = &raw[r * kRawP + c];
        uint32_t acc = 0;
#pragma unroll
        for(int k = 0; k < 7; ++k)
            acc += (uint32_t)taps.t[k] * p[k];
        hb[idx] = (uint16_t)acc; // <= 255*256, exact
    }
    __syncthreads();
    for(int idx = tid; idx < kBH * kBW; idx += 256)
    {
        const int r = idx >> 6, c = idx & 63;
        const int x = t.x0 + c, y = t.y0 + r;
        if(x >= lv.w || y >= lv.h)
            continue;
        uint32_t acc = 0;
#pragma unroll
        for(int k = 0; k < 7; ++k)
            acc += (uint32_t)taps.t[k] * hb[(r + k) * kBW + c];
        dst[(size_t)y * lv.pitch + x] = (uint8_t)min(255u, (acc + 32768u) >> 16);
    }
}

void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, const BlurTile* d_tiles, int n_frames,
                 hipStream_t s);

static Taps g_taps = {{18, 34, 48, 56, 48, 34, 18}};
void set_blur_taps(const int* t)
{
    for(int i = 0; i < 7; ++i)
        g_taps.t[i] = t[i];
}

void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, const BlurTile* d_tiles, int n_frames,
                 hipStream_t s)
{
    dim3 grid(g.n_tiles, n_frames);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, s, d_pyr, d_blur, g, d_tiles, g_taps);
}

} // namespace mslam
