// k_cvorb.hip — the kernels only the cv::ORB detector mode needs (MSLAM_HIP_DETECTOR_CV_ORB).
//
// Replaces OrbOpenCvDetector::Pimpl::detect (reference orb_feature.cpp:33-65): toGrayScale, then
// cv::ORB::create(1000)->detectAndCompute (orb_feature.cpp:25,40).  Everything below that call is OpenCV 4.8.1
// (features2d/src/orb.cpp; not in the reference tree), restated stage by stage:
//   pyramid      resize(prev, sz, INTER_LINEAR_EXACT), chained          -> k_resize_exact
//   FAST         FastFeatureDetector(20, nonmax = true) on the WHOLE level -> k_fast_tiles (64x64 tiles staged in LDS:
//                scores incl. a 1-px ring, 3x3 strict NMS, runByImageBorder(31), per-level list)
//   selection    retainBest(2 n_l) by FAST score, HarrisResponses (7x7 block, k = 0.04), retainBest(n_l) by Harris
//                                                                       -> k_cv_select (one workgroup per level; first
//                puts the list into FAST's raster order)
//   orientation, blur, rBRIEF: k_blur and k_describe (shared with the in-tree detector; k_describe takes its
//                cos/sin from include/mslam_sincos.h in this mode and the Harris response as the keypoint response)
// retainBest keeps the SET {response >= n-th largest} (ties kept); its ORDER in the reference comes from
// std::nth_element / std::partition and is implementation-defined, so lists stay in FAST's raster order here.
#include "common.hpp"
#include "arc_score.hpp"

namespace mslam
{

// ---- INTER_LINEAR_EXACT (imgproc resize.cpp: resize_bitExact, interpolationLinear<uchar>, ufixedpoint16) ------
// horizontal: h = c0 * S[o] + c1 * S[o+1] in 8.8 (u16, exact); left of xmin / right of xmax: edge sample << 8.
// vertical  : (h0 * b0 + h1 * b1 + 2^15) >> 16; above ymin / below ymax: (h + 128) >> 8 of the first / last row.
// The host tables fold the edge rules in: a position left of xmin is (offset 0, coefficients 256 / 0), right of xmax
// (offset sw-2, 0 / 256) — the same 8.8 value as "edge sample << 8" — and a row above ymin / below ymax likewise, where
// (h * 256 + 2^15) >> 16 == (h + 128) >> 8.  The kernel is therefore branch-free: 4 pixels per lane, all loads first.
// The kernel is branch-free: 4 destination pixels and R = 4 destination rows per lane.  The 4 pixels draw on at most
// 12 consecutive source bytes (scale <= 2), fetched per source row as three aligned dwords; the pair (S[o], S[o+1])
// of pixel k is cut out with one v_perm_b32 whose selector depends on (o_k - window base) only, i.e. is computed once
// per lane, as are the horizontal coefficients.
template <bool WINDOW>
__global__ __launch_bounds__(256) void k_resize_exact(ExactResizeArgs a)
{
    constexpr int R = 4;
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int dy0 = (blockIdx.y * 4 + threadIdx.y) * R;
    if(dy0 >= a.dh || (qx << 2) >= a.dw)
        return;
    const size_t frame = blockIdx.z + a.frame0;
    const uint8_t* src = a.pyr + frame * a.slab + a.src_off;
    uint8_t* dst = a.pyr + frame * a.slab + a.dst_off;
    int o[4];
    uint32_t c0[4], c1[4], sel[4];
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        const int dx = min((qx << 2) + k, a.dw - 1);
        o[k] = a.xofs[dx];
        const uint32_t xc = a.xcoef[dx];
        c0[k] = xc & 0xFFFF, c1[k] = xc >> 16;
    }
    const int base = o[0] & ~3;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        // bytes (s, s+1) of the 12-byte window -> u16 lanes [S[o], 0, S[o+1], 0]: from dwords (0,1) when s <= 6, else (1,2)
        const uint32_t sft = (uint32_t)(o[k] - base);
        sel[k] = 0x0c010c00u + (sft > 6 ? sft - 4 : sft) * 0x00010001u;
    }
    for(int r = 0; r < R; ++r)
    {
        const int dy = dy0 + r;
        if(dy >= a.dh)
            break;
        const int r0 = a.yofs[dy];
        const uint32_t yc = a.ycoef[dy];
        const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
        const uint8_t* S0 = src + (size_t)r0 * a.spitch;
        const uint8_t* S1 = S0 + a.spitch;
        uint32_t h0[4], h1[4];
        if(WINDOW)
        {
            const uint32_t* W0 = reinterpret_cast<const uint32_t*>(S0 + base);
            const uint32_t* W1 = reinterpret_cast<const uint32_t*>(S1 + base);
            const uint32_t a0 = W0[0], a1 = W0[1], a2 = W0[2], e0 = W1[0], e1 = W1[1], e2 = W1[2];
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                const bool up = (uint32_t)(o[k] - base) > 6;
                const uint32_t pa = __builtin_amdgcn_perm(up ? a2 : a1, up ? a1 : a0, sel[k]); // S0[o] | S0[o+1] << 16
                const uint32_t pe = __builtin_amdgcn_perm(up ? e2 : e1, up ? e1 : e0, sel[k]);
                // every factor is below 2^16 and every product below 2^24: full-rate 24-bit multiplies
                h0[k] = __umul24(c0[k], pa & 0xFFFF) + __umul24(c1[k], pa >> 16);
                h1[k] = __umul24(c0[k], pe & 0xFFFF) + __umul24(c1[k], pe >> 16);
            }
        }
        else
        {
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                h0[k] = __umul24(c0[k], S0[o[k]]) + __umul24(c1[k], S0[o[k] + 1]);
                h1[k] = __umul24(c0[k], S1[o[k]]) + __umul24(c1[k], S1[o[k] + 1]);
            }
        }
        uint32_t out = 0;
#pragma unroll
        for(int k = 0; k < 4; ++k)
            out |= (((__umul24(h0[k], b0) + __umul24(h1[k], b1) + 32768u) >> 16) & 0xFFu) << (8 * k);
        *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dpitch + (qx << 2)) = out;
    }
}

void launch_resize_exact(const ExactResizeArgs& a, int n_frames, hipStream_t s)
{
    dim3 grid(((a.dw + 3) / 4 + 63) / 64, (a.dh + 15) / 16, n_frames);
    if(a.window12) // every quad's source pixels fit the aligned 12-byte window (checked on the host)
        hipLaunchKernelGGL(k_resize_exact<true>, grid, dim3(64, 4), 0, s, a);
    else
        hipLaunchKernelGGL(k_resize_exact<false>, grid, dim3(64, 4), 0, s, a);
}

// ---- whole-level FAST-9/16 with 3x3 non-max suppression, per 64x64 tile -------------------------------------------
// score(x, y) = cornerScore<16> when the pixel passes the 9-contiguous test at `thr`, else 0, for 3 <= x < w-3,
// 3 <= y < h-3 (FAST_t's tested range); a keypoint is a pixel whose score is strictly greater than its 8 neighbours'.
// One workgroup per 64x64 tile: the 72x72 pixels it needs are staged in LDS, scores are computed for the tile plus a
// one-pixel ring (66x66: the neighbours of the tile's own pixels), the 4-point compass test rejects most pixels, the
// survivors are compacted into an LDS list and scored densely (same arc-score formulation as k_fast_cells:
// S = max over the 16 arcs of 9 of min(d) resp. min(-d), minus 1; corner iff S >= thr).  Keypoints inside the
// runByImageBorder(edge) rectangle are appended to the level's list (one global atomic per wave); the list is put into
// FAST's raster order by k_cv_select.
constexpr int kSP = 72; // LDS tile pitch = tile width 64 + 2 * (3 + 1)
constexpr int kSc = 68; // score-map pitch (66 used)

__global__ __launch_bounds__(256) void k_fast_tiles(const uint8_t* __restrict__ pyr, Geometry g, int level, int tiles_x,
                                                    int thr, CvSelectArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[72 * kSP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[66 * kSc];
    __shared__ uint16_t list[66 * 66];
    __shared__ uint32_t n_list;

    const LevelGeom& lv = g.lv[level];
    const int w = lv.w, h = lv.h, pitch = lv.pitch;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int X0 = tx * 64, Y0 = ty * 64; // the tile's own block; scores cover (X0-1, Y0-1) + 66x66, pixels (X0-4, Y0-4) + 72x72
    const size_t frame = blockIdx.y + g.frame0;
    const uint8_t* src = pyr + frame * g.slab + lv.offset;
    const int tid = threadIdx.x;

    for(int i = tid; i < 72 * 18; i += 256) // 72 rows x 18 dwords (4 pixels each)
    {
        const int r = i / 18, q = i - r * 18;
        const int y = min(max(Y0 - 4 + r, 0), h - 1), x = X0 - 4 + 4 * q;
        uint32_t v;
        if(x >= 0 && x + 3 < pitch) // X0 and the pitch are multiples of 4: aligned dword inside the row
            v = *reinterpret_cast<const uint32_t*>(src + (size_t)y * pitch + x);
        else
            v = 0; // outside the level: only ever read by pixels that are not tested
        reinterpret_cast<uint32_t*>(tile)[r * (kSP / 4) + q] = v;
    }
    for(int i = tid; i < 66 * kSc / 4; i += 256)
        reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if(tid == 0)
        n_list = 0;
    __syncthreads();

    // compass test over the 66x66 scored region: a 9-arc contains two adjacent compass points of one polarity
    for(int i = tid; i < 66 * 66; i += 256)
    {
        const int ly = i / 66, lx = i - ly * 66;
        const int x = X0 - 1 + lx, y = Y0 - 1 + ly;
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
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if((tid & 63) == leader)
                base = atomicAdd(&n_list, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader);
            if(keep)
                list[base + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)i;
        }
    }
    __syncthreads();
    const uint32_t n = n_list;
    for(uint32_t j = tid; 2 * j < n; j += 256) // two listed pixels per lane (arc_score.hpp)
    {
        const int ia = list[2 * j], ib = 2 * j + 1 < n ? list[2 * j + 1] : ia;
        const int ay = ia / 66, ax = ia - ay * 66, by = ib / 66, bx = ib - by * 66;
        const uint8_t* pa = &tile[(ay + 3) * kSP + ax + 3];
        const uint8_t* pb = &tile[(by + 3) * kSP + bx + 3];
        const uint32_t vv = ((uint32_t)pa[0] + kArcBias) | (((uint32_t)pb[0] + kArcBias) << 16);
        uint32_t e[16];
#define MSLAM_E(k, off) e[k] = vv - ((uint32_t)pa[off] | ((uint32_t)pb[off] << 16))
        MSLAM_E(0, 3 * kSP);
        MSLAM_E(1, 3 * kSP + 1);
        MSLAM_E(2, 2 * kSP + 2);
        MSLAM_E(3, kSP + 3);
        MSLAM_E(4, 3);
        MSLAM_E(5, -kSP + 3);
        MSLAM_E(6, -2 * kSP + 2);
        MSLAM_E(7, -3 * kSP + 1);
        MSLAM_E(8, -3 * kSP);
        MSLAM_E(9, -3 * kSP - 1);
        MSLAM_E(10, -2 * kSP - 2);
        MSLAM_E(11, -kSP - 3);
        MSLAM_E(12, -3);
        MSLAM_E(13, kSP - 3);
        MSLAM_E(14, 2 * kSP - 2);
        MSLAM_E(15, 3 * kSP - 1);
#undef MSLAM_E
        int sa, sb;
        arc_score2(e, sa, sb);
        if(sa >= thr && sa > 0)
            sc[ay * kSc + ax] = (uint8_t)sa;
        if(sb >= thr && sb > 0)
            sc[by * kSc + bx] = (uint8_t)sb;
    }
    __syncthreads();
    // 3x3 strict NMS over the listed pixels of the tile's own 64x64 block + runByImageBorder(edge)
    const int e = a.edge;
    uint32_t* out = a.cand + (frame * g.n_levels + level) * (size_t)a.cand_cap;
    uint32_t* out_cnt = a.cand_cnt + frame * g.n_levels + level;
    for(uint32_t i0 = 0; i0 < n; i0 += 256)
    {
        const uint32_t i = i0 + tid;
        bool is_kp = false;
        uint32_t packed = 0;
        if(i < n)
        {
            const int idx = list[i];
            const int ly = idx / 66, lx = idx - ly * 66;
            const int x = X0 - 1 + lx, y = Y0 - 1 + ly;
            const uint8_t* q = &sc[ly * kSc + lx];
            const int s = q[0];
            if(s != 0 && lx >= 1 && lx <= 64 && ly >= 1 && ly <= 64 && x >= e && x < w - e && y >= e && y < h - e)
            {
                const int m = max(max(max(q[-kSc - 1], q[-kSc]), max(q[-kSc + 1], q[-1])),
                                  max(max(q[1], q[kSc - 1]), max(q[kSc], q[kSc + 1])));
                is_kp = s > m;
                packed = pack_kp(x, y, s);
            }
        }
        const unsigned long long m = __ballot(is_kp);
        if(m)
        {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if((tid & 63) == leader)
                base = atomicAdd(out_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull));
            if(is_kp && pos < (uint32_t)a.cand_cap)
                out[pos] = packed;
        }
    }
}

void launch_fast_tiles(const uint8_t* d_pyr, const Geometry& g, int level, int thr, const CvSelectArgs& a, int frame0,
                       int n_frames, hipStream_t s)
{
    const LevelGeom& lv = g.lv[level];
    const int tiles_x = (lv.w + 63) / 64, tiles_y = (lv.h + 63) / 64;
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_tiles, dim3(tiles_x * tiles_y, n_frames), dim3(256), 0, s, d_pyr, gg, level, tiles_x, thr, a);
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_raw = (int)a.cand_cnt[slot];
    if(n_raw > a.cand_cap && tid == 0)
        atomicOr(a.flags, kFlagCandOverflow);
    const int n = min(n_raw, a.cand_cap);
    const int quota = a.quota[level];
    // The tile kernel appended the level's keypoints in arrival order; FAST's own order — the order everything
    // downstream is defined in — is raster (y, then x), which is ascending order of the packed words.  Bitonic sort
    // in place (global memory, visible inside the workgroup after a barrier; a level has ~10^3 keypoints).
    {
        uint32_t* keys = a.cand + slot * (size_t)a.cand_cap;
        int np2 = 1;
        while(np2 < n)
            np2 <<= 1;
        // normalised bitonic network (every comparator ascending; the first step of a merge mirrors the upper half), so a
        // length that is not a power of two only needs comparators reaching beyond n to be skipped
        for(int size = 2; size <= np2; size <<= 1)
            for(int stride = size >> 1; stride > 0; stride >>= 1)
            {
                for(int t = tid; t < (np2 >> 1); t += 256)
                {
                    int lo, hi;
                    if(stride == (size >> 1))
                    {
                        const int grp = t / stride, off = t - grp * stride;
                        lo = grp * size + off;
                        hi = grp * size + size - 1 - off;
                    }
                    else
                    {
                        lo = (t / stride) * stride * 2 + (t % stride);
                        hi = lo + stride;
                    }
                    if(hi < n)
                    {
                        const uint32_t x = keys[lo], y = keys[hi];
                        if(x > y)
                        {
                            keys[lo] = y;
                            keys[hi] = x;
                        }
                    }
                }
                __syncthreads();
            }
    }
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

void launch_cv_select(const uint8_t* d_pyr, const Geometry& g, const CvSelectArgs& a, int frame0, int n_frames,
                      hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_cv_select, dim3(g.n_levels, n_frames), dim3(256), 0, s, d_pyr, gg, a);
}

} // namespace mslam
