// k_cvorb.hip — the kernels only the cv::ORB detector mode needs (MSLAM_HIP_DETECTOR_CV_ORB).
//
// Replaces OrbOpenCvDetector::Pimpl::detect (reference orb_feature.cpp:33-65): toGrayScale, then
// cv::ORB::create(1000)->detectAndCompute (orb_feature.cpp:25,40).  Everything below that call is OpenCV 4.8.1
// (features2d/src/orb.cpp; not in the reference tree), restated stage by stage:
//   pyramid      resize(prev, sz, INTER_LINEAR_EXACT), chained          -> k_resize_exact
//   FAST         FastFeatureDetector(20, nonmax = true) on the WHOLE level -> k_fast_score (score plane, 64x64
//                tiles staged in LDS) + k_fast_nms (3x3 strict NMS, runByImageBorder(31), raster-order list)
//   selection    retainBest(2 n_l) by FAST score, HarrisResponses (7x7 block, k = 0.04), retainBest(n_l) by Harris
//                                                                       -> k_cv_select (one workgroup per level)
//   orientation, blur, rBRIEF: k_blur and k_describe (shared with the in-tree detector; k_describe takes its
//                cos/sin from include/mslam_sincos.h in this mode and the Harris response as the keypoint response)
// retainBest keeps the SET {response >= n-th largest} (ties kept); its ORDER in the reference comes from
// std::nth_element / std::partition and is implementation-defined, so lists stay in FAST's raster order here.
#include "common.hpp"

namespace mslam
{

// ---- INTER_LINEAR_EXACT (imgproc resize.cpp: resize_bitExact, interpolationLinear<uchar>, ufixedpoint16) ------
// horizontal: h = c0 * S[o] + c1 * S[o+1] in 8.8 (u16, exact); left of xmin / right of xmax: edge sample << 8.
// vertical  : (h0 * b0 + h1 * b1 + 2^15) >> 16; above ymin / below ymax: (h + 128) >> 8 of the first / last row.
__global__ __launch_bounds__(256) void k_resize_exact(ExactResizeArgs a)
{
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if(dy >= a.dh || (qx << 2) >= a.dw)
        return;
    const size_t frame = blockIdx.z + a.frame0;
    const uint8_t* src = a.pyr + frame * a.slab + a.src_off;
    uint8_t* dst = a.pyr + frame * a.slab + a.dst_off;
    const bool two = dy >= a.ymin && dy < a.ymax;
    const int r0 = dy < a.ymin ? 0 : (dy >= a.ymax ? a.sh - 1 : a.yofs[dy]);
    const uint32_t yc = a.ycoef[dy];
    const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
    const uint8_t* S0 = src + (size_t)r0 * a.spitch;
    const uint8_t* S1 = S0 + (two ? a.spitch : 0);
    uint32_t out = 0;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        const int dx = min((qx << 2) + k, a.dw - 1);
        uint32_t h0, h1;
        if(dx < a.xmin)
            h0 = (uint32_t)S0[0] << 8, h1 = (uint32_t)S1[0] << 8;
        else if(dx >= a.xmax)
            h0 = (uint32_t)S0[a.sw - 1] << 8, h1 = (uint32_t)S1[a.sw - 1] << 8;
        else
        {
            const int o = a.xofs[dx];
            const uint32_t xc = a.xcoef[dx];
            const uint32_t c0 = xc & 0xFFFF, c1 = xc >> 16;
            h0 = c0 * S0[o] + c1 * S0[o + 1];
            h1 = c0 * S1[o] + c1 * S1[o + 1];
        }
        const uint32_t v = two ? (h0 * b0 + h1 * b1 + 32768u) >> 16 : (h0 + 128u) >> 8;
        out |= (v & 0xFFu) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dpitch + (qx << 2)) = out;
}

void launch_resize_exact(const ExactResizeArgs& a, int n_frames, hipStream_t s)
{
    dim3 grid(((a.dw + 3) / 4 + 63) / 64, (a.dh + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_resize_exact, grid, dim3(64, 4), 0, s, a);
}

// ---- whole-level FAST-9/16 score plane --------------------------------------------------------------------------
// score(x, y) = cornerScore<16> when the pixel passes the 9-contiguous test at `thr`, else 0, for 3 <= x < w-3,
// 3 <= y < h-3 (FAST_t's tested range); everything else 0.  One workgroup per 64x64 tile: the 70x70 pixels it
// needs are staged in LDS, the 4-point compass test rejects most pixels, the survivors are compacted into an
// LDS list and scored densely (same arc-score formulation as k_fast_cells: S = max over the 16 arcs of 9 of
// min(d) resp. min(-d), minus 1; the pixel is a corner iff S >= thr).
constexpr int kSP = 72; // LDS tile pitch

__device__ __forceinline__ int cv_min3(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int cv_max3(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ int cv_arc_score(const int (&d)[16])
{
    int mn3[16], mx3[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        mn3[i] = cv_min3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
        mx3[i] = cv_max3(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    }
    int q0 = -1000, q1 = 1000;
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        q0 = max(q0, cv_min3(mn3[i], mn3[(i + 3) & 15], mn3[(i + 6) & 15]));
        q1 = min(q1, cv_max3(mx3[i], mx3[(i + 3) & 15], mx3[(i + 6) & 15]));
    }
    return max(q0, -q1) - 1;
}

__global__ __launch_bounds__(256) void k_fast_score(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ plane,
                                                    Geometry g, int level, int tiles_x, int thr)
{
    __shared__ uint8_t tile[70 * kSP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[64 * 64];
    __shared__ uint16_t list[64 * 64];
    __shared__ uint32_t n_list;

    const LevelGeom& lv = g.lv[level];
    const int w = lv.w, h = lv.h, pitch = lv.pitch;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X0 = tx * 64, Y0 = ty * 64; // the tile's output block; staged pixels start at (X0-3, Y0-3)
    const size_t frame = blockIdx.y + g.frame0;
    const uint8_t* src = pyr + frame * g.slab + lv.offset;
    const int tid = threadIdx.x;

    for(int i = tid; i < 70 * 70; i += 256)
    {
        const int r = i / 70, c = i - r * 70;
        const int y = min(max(Y0 - 3 + r, 0), h - 1), x = min(max(X0 - 3 + c, 0), w - 1); // clamped: only read by untested pixels
        tile[r * kSP + c] = src[(size_t)y * pitch + x];
    }
    for(int i = tid; i < 64 * 64 / 4; i += 256)
        reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if(tid == 0)
        n_list = 0;
    __syncthreads();

    // compass test: a 9-arc contains two adjacent compass points of one polarity
    for(int i = tid; i < 64 * 64; i += 256)
    {
        const int ly = i >> 6, lx = i & 63;
        const int x = X0 + lx, y = Y0 + ly;
        bool keep = false;
        if(x >= 3 && x < w - 3 && y >= 3 && y < h - 3)
        {
            const uint8_t* p = &tile[(ly + 3) * kSP + lx + 3];
            const int v = p[0], hi = v + thr, lo = v - thr;
            const int p0 = p[3 * kSP], p8 = p[-3 * kSP], p4 = p[3], p12 = p[-3];
            keep = ((p0 > hi || p8 > hi) && (p4 > hi || p12 > hi)) || ((p0 < lo || p8 < lo) && (p4 < lo || p12 < lo));
        }
        const unsigned long long m = __ballot(keep);
        if(m)
        {
            uint32_t base = 0;
            if((tid & 63) == (__ffsll((long long)m) - 1))
                base = atomicAdd(&n_list, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
            if(keep)
                list[base + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const uint32_t n = n_list;
    for(uint32_t i = tid; i < n; i += 256)
    {
        const int idx = list[i];
        const int ly = idx >> 6, lx = idx & 63;
        const uint8_t* p = &tile[(ly + 3) * kSP + lx + 3];
        const int v = p[0];
        int d[16];
        d[0] = v - p[3 * kSP];
        d[1] = v - p[3 * kSP + 1];
        d[2] = v - p[2 * kSP + 2];
        d[3] = v - p[kSP + 3];
        d[4] = v - p[3];
        d[5] = v - p[-kSP + 3];
        d[6] = v - p[-2 * kSP + 2];
        d[7] = v - p[-3 * kSP + 1];
        d[8] = v - p[-3 * kSP];
        d[9] = v - p[-3 * kSP - 1];
        d[10] = v - p[-2 * kSP - 2];
        d[11] = v - p[-kSP - 3];
        d[12] = v - p[-3];
        d[13] = v - p[kSP - 3];
        d[14] = v - p[2 * kSP - 2];
        d[15] = v - p[3 * kSP - 1];
        const int s = cv_arc_score(d);
        if(s >= thr && s > 0)
            sc[idx] = (uint8_t)s;
    }
    __syncthreads();
    // store the 64x64 block (rows beyond the level, or dwords beyond its pitch, are not written)
    uint8_t* dst = plane + frame * g.slab + lv.offset;
    for(int i = tid; i < 64 * 16; i += 256)
    {
        const int ly = i >> 4, q = i & 15;
        const int y = Y0 + ly, x = X0 + 4 * q;
        if(y < h && x < pitch)
            *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x) = reinterpret_cast<const uint32_t*>(sc)[i];
    }
}

void launch_fast_score(const uint8_t* d_pyr, uint8_t* d_plane, const Geometry& g, int level, int thr, int frame0,
                       int n_frames, hipStream_t s)
{
    const LevelGeom& lv = g.lv[level];
    const int tiles_x = (lv.pitch + 63) / 64, tiles_y = (lv.h + 63) / 64;
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_score, dim3(tiles_x * tiles_y, n_frames), dim3(256), 0, s, d_pyr, d_plane, gg, level, tiles_x,
                       thr);
}

// ---- 3x3 strict NMS + runByImageBorder + raster-order candidate list: one workgroup per (level, frame) ---------
__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t* __restrict__ plane, Geometry g, CvSelectArgs a)
{
    __shared__ uint32_t wsum[4];
    const int level = blockIdx.x;
    const size_t frame = blockIdx.y + g.frame0;
    const LevelGeom& lv = g.lv[level];
    const int w = lv.w, h = lv.h, pitch = lv.pitch, qpr = pitch >> 2; // dwords per row
    const uint8_t* sc = plane + frame * g.slab + lv.offset;
    uint32_t* out = a.cand + (frame * g.n_levels + level) * (size_t)a.cand_cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = a.edge;
    // only rows / dwords that can hold a kept keypoint are scanned: e <= y < h - e, e <= x < w - e
    const int y_lo = e, y_hi = h - e, q_lo = e >> 2, q_hi = min(qpr, ((w - e + 3) >> 2));
    const int qn = max(q_hi - q_lo, 0), rows = max(y_hi - y_lo, 0);
    const int total = qn * rows;
    uint32_t running = 0;
    for(int base = 0; base < total; base += 256)
    {
        const int i = base + tid;
        uint32_t found = 0, packed[4];
        if(i < total)
        {
            const int r = i / qn, q = q_lo + (i - r * qn), y = y_lo + r;
            const uint32_t dw = *reinterpret_cast<const uint32_t*>(sc + (size_t)y * pitch + 4 * q);
            if(dw != 0)
            {
#pragma unroll
                for(int b = 0; b < 4; ++b)
                {
                    const int s = (int)((dw >> (8 * b)) & 0xFF), x = 4 * q + b;
                    if(s != 0 && x >= e && x < w - e)
                    {
                        // every neighbour is inside the level here (e >= 1); untested pixels hold 0
                        const uint8_t* p = sc + (size_t)y * pitch + x;
                        const int m = max(max(max(p[-pitch - 1], p[-pitch]), max(p[-pitch + 1], p[-1])),
                                          max(max(p[1], p[pitch - 1]), max(p[pitch], p[pitch + 1])));
                        if(s > m)
                            packed[found++] = pack_kp(x, y, s);
                    }
                }
            }
        }
        // ordered compaction: exclusive prefix of `found` over the workgroup
        uint32_t inc = found;
#pragma unroll
        for(int o = 1; o < 64; o <<= 1)
        {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            inc += lane >= o ? t : 0;
        }
        if(lane == 63)
            wsum[wave] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for(int k = 0; k < 4; ++k)
        {
            pre += k < wave ? wsum[k] : 0;
            tot += wsum[k];
        }
        uint32_t pos = running + pre + inc - found;
        for(uint32_t k = 0; k < found; ++k, ++pos)
            if(pos < (uint32_t)a.cand_cap)
                out[pos] = packed[k];
        running += tot;
        __syncthreads();
    }
    if(tid == 0)
    {
        a.cand_cnt[frame * g.n_levels + level] = min(running, (uint32_t)a.cand_cap);
        if(running > (uint32_t)a.cand_cap)
            atomicOr(a.flags, kFlagCandOverflow);
    }
}

// ---- retainBest(2n) by FAST score -> Harris -> retainBest(n) by Harris: one workgroup per (level, frame) -------
__device__ __forceinline__ float harris_at(const uint8_t* img, int pitch, int x, int y)
{
    // orb.cpp HarrisResponses: blockSize 7, Sobel-like 3x3 gradients, integer sums, float response
    int a = 0, b = 0, c = 0;
    for(int i = -3; i <= 3; ++i)
    {
        const uint8_t* r0 = img + (size_t)(y + i - 1) * pitch + x;
        const uint8_t* r1 = r0 + pitch;
        const uint8_t* r2 = r1 + pitch;
#pragma unroll
        for(int j = -3; j <= 3; ++j)
        {
            const int Ix = (r1[j + 1] - r1[j - 1]) * 2 + (r0[j + 1] - r0[j - 1]) + (r2[j + 1] - r2[j - 1]);
            const int Iy = (r2[j] - r0[j]) * 2 + (r2[j - 1] - r0[j - 1]) + (r2[j + 1] - r0[j + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    }
    const float harris_k = 0.04f;
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float s4 = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float sum = __fadd_rn(fa, fb);
    const float v = __fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(harris_k, sum), sum));
    return __fmul_rn(v, s4);
}

__global__ __launch_bounds__(256) void k_cv_select(const uint8_t* __restrict__ pyr, Geometry g, CvSelectArgs a)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ int s_thr;
    const int level = blockIdx.x;
    const size_t frame = blockIdx.y + g.frame0;
    const LevelGeom& lv = g.lv[level];
    const size_t slot = frame * g.n_levels + level;
    const uint32_t* cand = a.cand + slot * (size_t)a.cand_cap;
    uint32_t* kept = a.tmp_kp + slot * (size_t)a.cand_cap;   // after the first retainBest
    float* kresp = a.tmp_resp + slot * (size_t)a.cand_cap;   // their Harris responses
    uint32_t* sel = a.sel + slot * (size_t)a.cand_cap;
    float* sresp = a.sel_resp + slot * (size_t)a.cand_cap;
    const uint8_t* img = pyr + frame * g.slab + lv.offset;
    const int n = (int)a.cand_cnt[slot];
    const int quota = a.quota[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ordered compaction of a predicate over [0, count): returns the total; `emit(i, pos)` stores element i
    auto compact = [&](int count, auto&& pred, auto&& emit) -> int {
        uint32_t running = 0;
        for(int base = 0; base < count; base += 256)
        {
            const int i = base + tid;
            const bool ok = i < count && pred(i);
            const unsigned long long m = __ballot(ok);
            if(lane == 0)
                wsum[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for(int k = 0; k < 4; ++k)
            {
                pre += k < wave ? wsum[k] : 0;
                tot += wsum[k];
            }
            if(ok)
                emit(i, running + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
            running += tot;
            __syncthreads();
        }
        return (int)running;
    };

    // 1. retainBest(2 * quota) by FAST score (integer 1..255): threshold = the (2 quota)-th largest score
    int thr = 0;
    if(n > 2 * quota)
    {
        hist[tid] = 0;
        __syncthreads();
        for(int i = tid; i < n; i += 256)
            atomicAdd(&hist[kp_score(cand[i])], 1u);
        __syncthreads();
        if(tid == 0)
        {
            int acc = 0, t = 255;
            for(; t > 0; --t)
            {
                acc += (int)hist[t];
                if(acc >= 2 * quota)
                    break;
            }
            s_thr = 2 * quota == 0 ? 256 : t;
        }
        __syncthreads();
        thr = s_thr;
    }
    const int m1 = compact(
        n, [&](int i) { return kp_score(cand[i]) >= thr; }, [&](int i, uint32_t pos) { kept[pos] = cand[i]; });
    __syncthreads();
    // 2. Harris responses (the writes above are visible to the whole workgroup after the barrier)
    for(int i = tid; i < m1; i += 256)
    {
        const uint32_t p = kept[i];
        kresp[i] = harris_at(img, lv.pitch, kp_x(p), kp_y(p));
    }
    __syncthreads();
    // 3. retainBest(quota) by Harris: keep iff fewer than `quota` responses are strictly greater (= response >= the
    //    quota-th largest; ties kept)
    const int m2 = compact(
        m1,
        [&](int i) {
            if(m1 <= quota)
                return true;
            if(quota == 0)
                return false;
            const float r = kresp[i];
            int greater = 0;
            for(int j = 0; j < m1; ++j)
                greater += kresp[j] > r ? 1 : 0;
            return greater < quota;
        },
        [&](int i, uint32_t pos) {
            const uint32_t p = kept[i];
            // k_describe takes coordinates relative to the (19, 19) origin of the in-tree detector's lists
            sel[pos] = pack_kp(kp_x(p) - kBorder, kp_y(p) - kBorder, kp_score(p));
            sresp[pos] = kresp[i];
        });
    if(tid == 0)
        a.sel_cnt[slot] = (uint32_t)m2;
}

void launch_cv_select(const uint8_t* d_pyr, const uint8_t* d_plane, const Geometry& g, const CvSelectArgs& a, int frame0,
                      int n_frames, hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_nms, dim3(g.n_levels, n_frames), dim3(256), 0, s, d_plane, gg, a);
    hipLaunchKernelGGL(k_cv_select, dim3(g.n_levels, n_frames), dim3(256), 0, s, d_pyr, gg, a);
}

} // namespace mslam
