// k_fast.hip — per-cell FAST-9/16 corner score + 3x3 non-max suppression + ordered emission.
//
// Replaces the cell loop of compute_fast_keypoints (reference distributed_cv_feature.cpp:858-952):
// for every 64-px cell (70x70 sub-image with the 6-px overlap) cv::FAST(sub, thr=ini, nms=true) and,
// only if that cell yields nothing, cv::FAST(sub, thr=min, nms=true) (:918-926).
//
// One 256-thread workgroup per (cell, frame).  The kernel is integer-VALU bound, so it is organised
// to spend instructions only where corners can be:
//   A. every tested pixel takes the 4-point compass test (a 9-arc always contains two adjacent
//      compass points of one polarity); survivors are compacted into an LDS list with one ballot per row;
//   B. the list is processed densely (all 64 lanes busy): 16 circle reads and the arc score
//        S = max(max_arc min(d), max_arc min(-d)) - 1,   d = centre - circle pixel,
//      over the 16 arcs of 9 pixels, with 3-input min/max.  For a pixel that passes the 9-contiguous
//      test at threshold t this is OpenCV's cornerScore<16>(.., t), and it passes iff S >= t, so the
//      test itself never has to be evaluated separately;
//   C. listed pixels with S >= t are checked against their 8 neighbours in the LDS score map (strict >)
//      and set one bit in a 64x64 bitmap; row popcounts of the bitmap give the reference's row-major
//      output order with no dependence on atomic ordering.
// If the cell is empty at the first threshold the same three steps run again with the fallback one.
#include "common.hpp"

namespace mslam
{

constexpr int kTileP = 76; // tile row pitch in bytes: 19 dwords, column 0 = sub-image column -3 (dword aligned)
constexpr int kTileX = 3;  // tile column of sub-image column 0
constexpr int kScP = 68;   // score-map row pitch

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ int arc_score(const int (&d)[16])
{
    // window-9 min and max over the circular sequence: two levels of 3-input ops
    int mn3[16], mx3[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn3[i] = min3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
        mx3[i] = max3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    }
    int mn9[16], mx9[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn9[i] = min3i(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]);
        mx9[i] = max3i(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]);
    }
    int q0 = max3i(mn9[0], mn9[1], mn9[2]), q1 = min3i(mx9[0], mx9[1], mx9[2]);
#pragma unroll
    for(int i = 3; i < 15; i += 2)
    {
        q0 = max3i(q0, mn9[i], mn9[i + 1]);
        q1 = min3i(q1, mx9[i], mx9[i + 1]);
    }
    q0 = max(q0, mn9[15]);
    q1 = min(q1, mx9[15]);
    return max(q0, -q1) - 1;
}

__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, Geometry g,
                                                    const CellDesc* __restrict__ cells, uint32_t* __restrict__ cell_cnt,
                                                    uint32_t* __restrict__ cell_kp, int ini_thr, int min_thr)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[70 * kTileP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[66 * kScP];
    __shared__ uint16_t cand[64 * 64];
    __shared__ uint32_t bitmap[64 * 2];
    __shared__ uint32_t n_cand;

    const int cell_id = blockIdx.x;
    const size_t frame = blockIdx.y + g.frame0;
    const CellDesc c = cells[cell_id];
    const LevelGeom& lv = g.lv[c.level];
    const int cw = c.cw, ch = c.ch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // zero the score map (untested pixels must read 0, like FAST_t's zeroed row buffers) and the bitmap
    for(int i = tid; i < 66 * kScP / 4; i += 256)
        reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if(tid < 128)
        bitmap[tid] = 0;
    if(tid == 0)
        n_cand = 0;

    // stage the sub-image with aligned dword loads: x0 - 3 is a multiple of 4 (x0 = 19 + 64 j)
    {
        const uint8_t* src = pyr + frame * g.slab + lv.offset + (size_t)c.y0 * lv.pitch + (c.x0 - kTileX);
        const int n_dw = ch * 19;
        for(int i = tid; i < n_dw; i += 256)
        {
            const int r = i / 19, q = i - r * 19;
            reinterpret_cast<uint32_t*>(tile)[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)r * lv.pitch + 4 * q);
        }
    }
    __syncthreads();

    const int x = lane + 3; // tested columns: 3 <= x < cw - 3
    const bool col_ok = x < cw - 3;
    uint32_t total = 0;
    for(int pass = 0; pass < 2; ++pass)
    {
        const int thr = pass == 0 ? ini_thr : min_thr;

        // ---- A. compass test + compaction
#pragma unroll 1
        for(int r = 0; r < 16; ++r)
        {
            const int y = 3 + wave * 16 + r;
            bool keep = false;
            if(col_ok && y < ch - 3)
            {
                const uint8_t* p = &tile[y * kTileP + x + kTileX];
                const int v = p[0];
                const int hi = v + thr, lo = v - thr;
                const int p0 = p[3 * kTileP], p4 = p[3], p8 = p[-3 * kTileP], p12 = p[-3];
                const bool b0 = p0 > hi, b4 = p4 > hi, b8 = p8 > hi, b12 = p12 > hi;
                const bool d0 = p0 < lo, d4 = p4 < lo, d8 = p8 < lo, d12 = p12 < lo;
                keep = ((b0 | b8) & (b4 | b12)) | ((d0 | d8) & (d4 | d12));
            }
            const unsigned long long b = __ballot(keep);
            if(b != 0)
            {
                uint32_t base = 0;
                if(lane == 0)
                    base = atomicAdd(&n_cand, (uint32_t)__popcll(b));
                base = __builtin_amdgcn_readfirstlane(base);
                if(keep)
                    cand[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)((y << 8) | x);
            }
        }
        __syncthreads();
        const uint32_t n = n_cand;

        // ---- B. arc score of the survivors
        for(uint32_t i = tid; i < n; i += 256)
        {
            const uint32_t cxy = cand[i];
            const int cy = (int)(cxy >> 8), cx = (int)(cxy & 0xFF);
            const uint8_t* p = &tile[cy * kTileP + cx + kTileX];
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
            const int s = arc_score(d);
            if(s >= thr && s > 0)
                sc[(cy - 2) * kScP + (cx - 2)] = (uint8_t)s;
        }
        __syncthreads();

        // ---- C. 3x3 strict non-max suppression over the listed pixels -> bitmap
        for(uint32_t i = tid; i < n; i += 256)
        {
            const uint32_t cxy = cand[i];
            const int cy = (int)(cxy >> 8), cx = (int)(cxy & 0xFF);
            const uint8_t* q = &sc[(cy - 2) * kScP + (cx - 2)];
            const int s = q[0];
            if(s >= thr && s > 0)
            {
                const int m = max(max3i(q[-kScP - 1], q[-kScP], q[-kScP + 1]),
                                  max(max3i(q[-1], q[1], q[kScP - 1]), max(q[kScP], q[kScP + 1])));
                if(s > m)
                    atomicOr(&bitmap[(cy - 3) * 2 + ((cx - 3) >> 5)], 1u << ((cx - 3) & 31));
            }
        }
        __syncthreads();

        // ---- ordered emission: row r of the bitmap = tested row r + 3, bit = tested column - 3
        if(wave == 0)
        {
            const unsigned long long m = bitmap[lane * 2] | ((unsigned long long)bitmap[lane * 2 + 1] << 32);
            const uint32_t cnt = (uint32_t)__popcll(m);
            uint32_t inc = cnt;
#pragma unroll
            for(int o = 1; o < 64; o <<= 1)
            {
                const uint32_t t = __shfl_up(inc, o);
                if(lane >= o)
                    inc += t;
            }
            total = __shfl(inc, 63);
            if(total != 0)
            {
                uint32_t* out = cell_kp + (frame * g.n_cells + cell_id) * (size_t)kCellCap;
                uint32_t pos = inc - cnt;
                unsigned long long rest = m;
                const int y = lane + 3;
                while(rest)
                {
                    const int bx = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    out[pos++] = pack_kp(bx + 3 + c.ox, y + c.oy, sc[(y - 2) * kScP + (bx + 3 - 2)]);
                }
            }
            if(lane == 0)
                n_cand = total != 0 ? 0xFFFFFFFFu : 0u; // tells the other waves whether to run the fallback pass
        }
        __syncthreads();
        const bool done = n_cand != 0;
        __syncthreads();
        if(done || pass == 1)
            break;
        // fallback pass: the map written so far (S >= ini_thr) is a subset of the fallback map; n_cand is 0 again
    }
    if(tid == 0)
        cell_cnt[frame * g.n_cells + cell_id] = total;
}

void launch_fast(const uint8_t* d_pyr, const Geometry& g, const CellDesc* d_cells, uint32_t* d_cell_cnt,
                 uint32_t* d_cell_kp, int ini_thr, int min_thr, int frame0, int n_frames, hipStream_t s)
{
    dim3 grid(g.n_cells, n_frames);
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d_pyr, gg, d_cells, d_cell_cnt, d_cell_kp, ini_thr, min_thr);
}

} // namespace mslam
