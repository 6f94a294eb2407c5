// k_fast.hip — per-cell FAST-9/16 corner score + 3x3 non-max suppression + ordered emission.
//
// Replaces the cell loop of compute_fast_keypoints (reference distributed_cv_feature.cpp:858-952):
// for every 64-px cell (70x70 sub-image with the 6-px overlap) cv::FAST(sub, thr=ini, nms=true) and,
// only if that cell yields nothing, cv::FAST(sub, thr=min, nms=true) (:918-926).
//
// One 256-thread workgroup per (cell, frame).  The 70x70 tile is staged in LDS once; a wave owns 16
// rows and its 64 lanes own the 64 tested columns, so a row's keypoints are one __ballot and the
// reference's row-major output order falls out of popcounts (no atomics-order dependence).
//
// Score identity used: for a pixel that passes the 9-contiguous test at threshold t, OpenCV's
// cornerScore<16>(.., t) equals  S = max(max_arc min(d), max_arc min(-d)) - 1  over the 16 arcs of 9
// circle pixels (d = v - p), and the pixel passes at threshold t iff S >= t.  So one S map serves both
// thresholds: map_t = (S >= t ? S : 0).  (The parity tests check this against a literal restatement of the scalar loops.)
#include "common.hpp"

namespace mslam
{

constexpr int kTileP = 72; // tile row pitch (bytes)
constexpr int kScP = 68;   // score row pitch

__device__ __forceinline__ bool has_arc9(uint32_t m16)
{
    const uint32_t m = m16 | (m16 << 16);
    uint32_t x = m & (m >> 1);
    x &= x >> 2;
    x &= x >> 4;
    x &= m >> 8;
    return (x & 0xFFFFu) != 0;
}

__device__ __forceinline__ int arc_score(const int (&d)[16])
{
    // sliding min / max of width 9 over the circular sequence d[0..15], by doubling
    int mn2[16], mx2[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn2[i] = min(d[i], d[(i + 1) & 15]);
        mx2[i] = max(d[i], d[(i + 1) & 15]);
    }
    int mn4[16], mx4[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn4[i] = min(mn2[i], mn2[(i + 2) & 15]);
        mx4[i] = max(mx2[i], mx2[(i + 2) & 15]);
    }
    int q0 = -1000, q1 = 1000;
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        const int mn9 = min(min(mn4[i], mn4[(i + 4) & 15]), d[(i + 8) & 15]);
        const int mx9 = max(max(mx4[i], mx4[(i + 4) & 15]), d[(i + 8) & 15]);
        q0 = max(q0, mn9);
        q1 = min(q1, mx9);
    }
    return max(q0, -q1) - 1;
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, Geometry g,
                                                    const CellDesc* __restrict__ cells, uint32_t* __restrict__ cell_cnt,
                                                    uint32_t* __restrict__ cell_kp, int ini_thr, int min_thr)
{
    __shared__ uint8_t tile[70 * kTileP];
    __shared__ uint8_t sc[66 * kScP];
    __shared__ uint32_t row_cnt[64];
    __shared__ uint32_t total;

    const int cell_id = blockIdx.x;
    const size_t frame = blockIdx.y;
    const CellDesc c = cells[cell_id];
    const LevelGeom& lv = g.lv[c.level];
    const int cw = c.cw, ch = c.ch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // zero the score map (untested pixels must read 0, like FAST_t's zeroed row buffers)
    for(int i = tid; i < 66 * kScP / 4; i += 256)
        reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if(tid < 64)
        row_cnt[tid] = 0;
    if(tid == 0)
        total = 0;

    // stage the sub-image
    const uint8_t* src = pyr + frame * g.slab + lv.offset + (size_t)c.y0 * lv.pitch + c.x0;
    for(int r = wave; r < ch; r += 4)
    {
        const uint8_t* s = src + (size_t)r * lv.pitch;
        if(lane < cw)
            tile[r * kTileP + lane] = s[lane];
        if(lane + 64 < cw)
            tile[r * kTileP + lane + 64] = s[lane + 64];
    }
    __syncthreads();

    // score map: S for pixels that are corners at min_thr (superset of the corners at ini_thr)
    const int x = lane + 3;
    const bool col_ok = x < cw - 3;
#pragma unroll 1
    for(int r = 0; r < 16; ++r)
    {
        const int y = 3 + wave * 16 + r;
        if(!(col_ok && y < ch - 3))
            continue;
        const uint8_t* p = &tile[y * kTileP + x];
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * kTileP];
        d[1] = v - p[3 * kTileP + 1];
        d[2] = v - p[2 * kTileP + 2];
        d[3] = v - p[kTileP + 3];
        d[4] = v - p[3];
        d[5] = v - p[-kTileP + 3];
        d[6] = v - p[-2 * kTileP + 2];
        d[7] = v - p[-3 * kTileP + 1];
        d[8] = v - p[-3 * kTileP];
        d[9] = v - p[-3 * kTileP - 1];
        d[10] = v - p[-2 * kTileP - 2];
        d[11] = v - p[-kTileP - 3];
        d[12] = v - p[-3];
        d[13] = v - p[kTileP - 3];
        d[14] = v - p[2 * kTileP - 2];
        d[15] = v - p[3 * kTileP - 1];
        uint32_t dark = 0, bright = 0; // circle pixel darker / brighter than the centre by more than min_thr
#pragma unroll
        for(int k = 0; k < 16; ++k)
        {
            dark |= (uint32_t)(d[k] > min_thr) << k;
            bright |= (uint32_t)(d[k] < -min_thr) << k;
        }
        if(has_arc9(dark) || has_arc9(bright))
            sc[(y - 2) * kScP + (x - 2)] = (uint8_t)arc_score(d);
    }
    __syncthreads();

    // NMS at ini_thr; fall back to min_thr only when the whole cell is empty (:922-926)
    uint32_t flags = 0; // bit r: (row 16*wave+r, this column) is a keypoint
    for(int pass = 0; pass < 2; ++pass)
    {
        const int thr = pass == 0 ? ini_thr : min_thr;
        flags = 0;
#pragma unroll 1
        for(int r = 0; r < 16; ++r)
        {
            const int y = 3 + wave * 16 + r;
            bool kp = false;
            if(col_ok && y < ch - 3)
            {
                const uint8_t* q = &sc[(y - 2) * kScP + (x - 2)];
                const int s = q[0];
                if(s >= thr && s > 0)
                {
                    // neighbours below thr are non-corners at this threshold: their map value is 0
                    int m = 0;
                    int n;
                    n = q[-kScP - 1]; m = max(m, n >= thr ? n : 0);
                    n = q[-kScP];     m = max(m, n >= thr ? n : 0);
                    n = q[-kScP + 1]; m = max(m, n >= thr ? n : 0);
                    n = q[-1];        m = max(m, n >= thr ? n : 0);
                    n = q[1];         m = max(m, n >= thr ? n : 0);
                    n = q[kScP - 1];  m = max(m, n >= thr ? n : 0);
                    n = q[kScP];      m = max(m, n >= thr ? n : 0);
                    n = q[kScP + 1];  m = max(m, n >= thr ? n : 0);
                    kp = s > m;
                }
            }
            const unsigned long long b = __ballot(kp);
            if(kp)
                flags |= 1u << r;
            if(lane == 0)
                row_cnt[wave * 16 + r] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        if(tid < 64)
        {
            uint32_t v = row_cnt[tid];
            for(int o = 32; o > 0; o >>= 1)
                v += __shfl_xor(v, o);
            if(tid == 0)
                total = v;
        }
        __syncthreads();
        if(total != 0 || pass == 1)
            break;
        __syncthreads(); // everyone has read `total` before row_cnt/total are rewritten
    }

    const uint32_t n_total = total;
    uint32_t* out = cell_kp + (frame * g.n_cells + cell_id) * (size_t)kCellCap;
    if(tid == 0)
        cell_cnt[frame * g.n_cells + cell_id] = n_total;
    if(n_total == 0)
        return;
    // ordered emission: rows ascending, columns ascending inside a row
    uint32_t base = 0;
    for(int rr = 0; rr < wave * 16; ++rr)
        base += row_cnt[rr];
#pragma unroll 1
    for(int r = 0; r < 16; ++r)
    {
        const bool kp = (flags >> r) & 1u;
        const unsigned long long b = __ballot(kp);
        if(kp)
        {
            const int y = 3 + wave * 16 + r;
            const uint32_t pos = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            out[pos] = pack_kp(x + c.ox, y + c.oy, sc[(y - 2) * kScP + (x - 2)]);
        }
        base += (uint32_t)__popcll(b);
    }
}

void launch_fast(const uint8_t* d_pyr, const Geometry& g, const CellDesc* d_cells, uint32_t* d_cell_cnt,
                 uint32_t* d_cell_kp, int ini_thr, int min_thr, int n_frames, hipStream_t s)
{
    dim3 grid(g.n_cells, n_frames);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d_pyr, g, d_cells, d_cell_cnt, d_cell_kp, ini_thr, min_thr);
}

} // namespace mslam
