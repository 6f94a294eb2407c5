// k_match.hip — brute-force 256-bit Hamming knn-2 matcher + ratio test.
//
// Replaces OrbOpenCvMatcher::Pimpl::match (reference orb_feature.cpp:84-117):
// BFMatcher(NORM_HAMMING).knnMatch(query = to, train = from, k = 2) then the ratio test and the
// (fromIndex = trainIdx, toIndex = queryIdx) output in query order.
//
// k_match_knn2: one lane owns one query descriptor (8 dwords in registers); train rows arrive through
// the scalar unit (see the kernel).  On equal distances the lower train index ranks first — exactly
// batchDistance's insertion rule (strict `<` on insertion while scanning train rows in ascending order).
// Not HBM-bound: inputs are 2*K*32 bytes against K^2 popcount-compares.
#include "common.hpp"
#include <climits>
#include <cstdlib>

namespace mslam
{

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    // v_bcnt_u32_b32 d, x, acc = popcount(x) + acc in ONE instruction; written as asm because the
    // compiler otherwise re-associates the eight partial sums into popcounts + separate adds
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// A train row is the same for every lane of a wave, so it is fetched with SCALAR loads (s_load_dwordx8
// through the scalar cache) and used as an SGPR operand of the per-lane xor: no LDS staging, no vector
// memory traffic in the loop.  Each lane owns QL queries; MW waves of a workgroup scan disjoint slices
// of the train rows for the same queries.  The top-2 of a query is kept as two packed keys
// (distance << 16 | train index): keys are unique and ordered by (distance, index), so
// best0' = min(best0, key), best1' = med3(best0, best1, key) is exactly batchDistance's insertion
// rule, and the partial results of the waves merge exactly.
template <int MW, int QL, int UNR>
__global__ __launch_bounds__(64 * MW) void k_match_knn2(MatchArgs a)
{
    __shared__ uint32_t part[MW][QL][2][64];

    // XCD-aware mapping: workgroups are handed to the 8 XCDs round-robin by linear id, and every workgroup
    // of a pair scans that pair's whole train set, so all workgroups of a pair are put on ONE XCD (id & 7):
    // the train set is then fetched into a single L2 instead of all eight.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / a.wg_per_pair) * 8 + xcd;
    if(pair >= a.n_pairs)
        return;
    const int n_from = min(a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed, 65535);
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int q0 = (slot % a.wg_per_pair) * (64 * QL);
    if(q0 >= n_to)
        return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t* __restrict__ from =
        reinterpret_cast<const uint32_t*>(a.from_desc + (long long)pair * a.from_stride);
    const uint8_t* to = a.to_desc + (long long)pair * a.to_stride;

    uint4 qa[QL], qb[QL];
    uint32_t best0[QL], best1[QL];
#pragma unroll
    for(int u = 0; u < QL; ++u)
    {
        const int q = q0 + u * 64 + lane;
        qa[u] = make_uint4(0, 0, 0, 0);
        qb[u] = qa[u];
        if(q < n_to)
        {
            const uint4* qp = reinterpret_cast<const uint4*>(to + (size_t)q * 32);
            qa[u] = qp[0];
            qb[u] = qp[1];
        }
        best0[u] = best1[u] = 0xFFFFFFFFu;
    }
    const int chunk = (n_from + MW - 1) / MW;
    const int j0 = wave * chunk, j1 = min(n_from, j0 + chunk);
    auto row = [&](int j) {
        const uint32_t* __restrict__ t = from + (size_t)j * 8; // wave-uniform address -> s_load_dwordx8
        const uint32_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5], t6 = t[6], t7 = t[7];
#pragma unroll
        for(int u = 0; u < QL; ++u)
        {
            uint32_t d = __popc(qa[u].x ^ t0);
            d = bcnt_acc(qa[u].y ^ t1, d);
            d = bcnt_acc(qa[u].z ^ t2, d);
            d = bcnt_acc(qa[u].w ^ t3, d);
            d = bcnt_acc(qb[u].x ^ t4, d);
            d = bcnt_acc(qb[u].y ^ t5, d);
            d = bcnt_acc(qb[u].z ^ t6, d);
            d = bcnt_acc(qb[u].w ^ t7, d);
            const uint32_t key = (d << 16) | (uint32_t)j;
            best1[u] = med3_u32(best0[u], best1[u], key); // best0 <= best1: the median is the new runner-up
            best0[u] = min(best0[u], key);
        }
    };
    int j = j0;
    for(; j + UNR <= j1; j += UNR)
    {
#pragma unroll
        for(int k = 0; k < UNR; ++k)
            row(j + k);
    }
    for(; j < j1; ++j)
        row(j);
#pragma unroll
    for(int u = 0; u < QL; ++u)
    {
        part[wave][u][0][lane] = best0[u];
        part[wave][u][1][lane] = best1[u];
    }
    __syncthreads();
    // wave u merges the partial top-2 of query slice u
    for(int u = wave; u < QL; u += MW)
    {
        const int q = q0 + u * 64 + lane;
        if(q < n_to)
        {
            uint32_t b0 = 0xFFFFFFFFu, b1 = 0xFFFFFFFFu;
#pragma unroll
            for(int w = 0; w < MW; ++w)
#pragma unroll
                for(int k = 0; k < 2; ++k)
                {
                    const uint32_t key = part[w][u][k][lane];
                    b1 = min(max(b0, key), b1);
                    b0 = min(b0, key);
                }
            const size_t o = (size_t)pair * a.cap + q;
            a.idx0[o] = b0 == 0xFFFFFFFFu ? -1 : (int32_t)(b0 & 0xFFFFu);
            a.idx1[o] = b1 == 0xFFFFFFFFu ? -1 : (int32_t)(b1 & 0xFFFFu);
            a.dist0[o] = b0 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b0 >> 16);
            a.dist1[o] = b1 == 0xFFFFFFFFu ? INT_MAX : (int32_t)(b1 >> 16);
        }
    }
}

template <int MW, int QL, int UNR>
static void launch_variant(MatchArgs a, int n_pairs, hipStream_t s)
{
    a.n_pairs = n_pairs;
    a.wg_per_pair = (a.cap + 64 * QL - 1) / (64 * QL);
    const unsigned grid = (unsigned)((n_pairs + 7) / 8) * 8u * (unsigned)a.wg_per_pair;
    hipLaunchKernelGGL((k_match_knn2<MW, QL, UNR>), dim3(grid), dim3(64 * MW), 0, s, a);
}


// ---------------------------------------------------------------------------------------------------
// MFMA form of the same search.  With every descriptor bit b expanded to the i8 value 2b-1, the i8 dot
// product of two descriptors is 256 - 2*hamming, exactly (integers, i32 accumulate), so a 32x32 tile of
// distances is 8 v_mfma_i32_32x32x32_i8 instead of 32*32*16 xor/popcount lane-ops.  Queries are the B
// operand: the accumulator then has ONE query per lane column and 16 train rows in the lane's 16
// registers, so the running top-2 of a query stays in its lane (2 lanes per query, merged at the end).
//   key of (query, train j) = (dot + 257) << 16 | age,   age = 32*(tiles scanned after j's tile) + 31 - (j & 31)
// larger key = smaller distance, then smaller train index: the order of batchDistance's insertion rule.
// The accumulator is initialised to 257 so that valid keys are >= 1 << 16 and 0 means "no neighbour".
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MM_TROW = 272; // bytes of one expanded train row in LDS: 256 + 16 so that a fragment read is conflict-free

__device__ __forceinline__ uint32_t lshl_add(uint32_t x, uint32_t sh, uint32_t y)
{
    return (x << sh) + y; // v_lshl_add_u32
}

template <int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_match_knn2_mfma(MatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][32 * MM_TROW];
    __shared__ uint2 lut[256]; // byte -> its 8 bits as i8 +1 / -1

    // XCD-aware mapping, as in k_match_knn2: every workgroup of a pair runs on the XCD (id & 7)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / a.wg_per_pair) * 8 + xcd;
    if(pair >= a.n_pairs)
        return;
    const int n_from = min(a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed, 65535);
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int q0 = (slot % a.wg_per_pair) * (128 * QT);
    if(q0 >= n_to)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t* __restrict__ from =
        reinterpret_cast<const uint32_t*>(a.from_desc + (long long)pair * a.from_stride);
    const uint8_t* to = a.to_desc + (long long)pair * a.to_stride;

    {
        auto spread = [](uint32_t x4) { // 4 bits -> 4 bytes of +1 / -1
            const uint32_t nb = ((x4 ^ 15u) * 0x00204081u) & 0x01010101u; // 1 where the bit is 0
            return ((nb << 8) - nb) | 0x01010101u;                         // 0xFF there, 0x01 elsewhere
        };
        lut[tid] = make_uint2(spread(tid & 15), spread(tid >> 4));
    }
    __syncthreads();

    // B fragments: lane (r, h) holds bits [32 s + 16 h, +16) of query r for k-step s
    v4i b[QT][8];
#pragma unroll
    for(int u = 0; u < QT; ++u)
    {
        const int q = q0 + (wave * QT + u) * 32 + r;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if(q < n_to)
        {
            const uint4* qp = reinterpret_cast<const uint4*>(to + (size_t)q * 32);
            lo = qp[0];
            hi = qp[1];
        }
        const uint32_t dw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for(int s = 0; s < 8; ++s)
        {
            const uint32_t half = (dw[s] >> (16 * h)) & 0xFFFFu;
            const uint2 e0 = lut[half & 255u], e1 = lut[half >> 8];
            b[u][s] = v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
        }
    }

    // train tile staging: thread tid expands dword (tid & 7) of row (tid >> 3) of the tile
    const int tr = tid >> 3, tw = tid & 7;
    const int n_tiles = (n_from + 31) >> 5;
    if(n_tiles == 0)
    {
        // no train rows at all: every query gets "no neighbour"
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            const int q = q0 + (wave * QT + u) * 32 + r;
            if(h == 0 && q < n_to)
            {
                const size_t o = (size_t)pair * a.cap + q;
                a.idx0[o] = a.idx1[o] = -1;
                a.dist0[o] = a.dist1[o] = INT_MAX;
            }
        }
        return;
    }
    // rows past the end re-read the last row (no branch); they are only ever used in the last tile, where
    // their keys are masked
    auto fetch = [&](int t) -> uint32_t { return from[(size_t)min(t * 32 + tr, n_from - 1) * 8 + tw]; };
    uint32_t d_next;
    auto stage = [&](int buf, uint32_t d) {
        const uint2 e0 = lut[d & 255u], e1 = lut[(d >> 8) & 255u], e2 = lut[(d >> 16) & 255u], e3 = lut[d >> 24];
        uint4* dst = reinterpret_cast<uint4*>(&tile[buf][tr * MM_TROW + tw * 32]);
        dst[0] = make_uint4(e0.x, e0.y, e1.x, e1.y);
        dst[1] = make_uint4(e2.x, e2.y, e3.x, e3.y);
    };

    uint32_t T[16]; // age of register i's train row inside its tile: 31 - row
    v16i cinit, zero;
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        T[i] = 31u - (uint32_t)((i & 3) + 8 * (i >> 2) + 4 * h);
        cinit[i] = 257;
        zero[i] = 0;
    }
    uint32_t best0[QT], best1[QT];
#pragma unroll
    for(int u = 0; u < QT; ++u)
        best0[u] = best1[u] = 0u;

    // running top-2 of one query over the 16 train rows a lane holds of one tile
    auto top2 = [&](const v16i& acc, uint32_t& b0, uint32_t& b1) {
        b0 += 32u; // everything found so far is one tile older
        b1 += 32u;
#pragma unroll
        for(int i = 0; i < 16; ++i)
        {
            const uint32_t g = lshl_add((uint32_t)acc[i], 16, T[i]);
            b1 = med3_u32(b0, b1, g); // b0 >= b1: the median is the new runner-up
            b0 = max(b0, g);
        }
    };
    // One tile step, software-pipelined by hand: the MFMAs of tile t run beside the top-2 VALU work of
    // tile t-1 (independent registers), which the matrix pipe otherwise waits for (8 MFMAs = 256 cycles,
    // top-2 of their 16 results = 50 VALU issues = 200 cycles).  Branch-free so it stays one block.
    auto step = [&](int t, const v16i (&old_acc)[QT], v16i (&new_acc)[QT]) {
        const int buf = t & 1;
        stage(buf ^ 1, d_next);
        d_next = fetch(t + 2);
        v4i af[8];
#pragma unroll
        for(int s = 0; s < 8; ++s)
            af[s] = *reinterpret_cast<const v4i*>(&tile[buf][r * MM_TROW + s * 32 + h * 16]);
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            v16i acc = cinit;
#pragma unroll
            for(int s = 0; s < 8; ++s)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], b[u][s], acc, 0, 0, 0);
            new_acc[u] = acc;
        }
#pragma unroll
        for(int u = 0; u < QT; ++u)
            top2(old_acc[u], best0[u], best1[u]);
        __syncthreads();
    };

    v16i accP[QT], accQ[QT];
#pragma unroll
    for(int u = 0; u < QT; ++u)
        accP[u] = zero; // "tile -1": keys below 1 << 16 never beat a real neighbour
    stage(0, fetch(0));
    d_next = fetch(1);
    __syncthreads();
    int t = 0;
    for(; t + 1 < n_tiles; t += 2)
    {
        step(t, accP, accQ);
        step(t + 1, accQ, accP);
    }
    if(t < n_tiles)
    {
        step(t, accP, accQ);
#pragma unroll
        for(int u = 0; u < QT; ++u)
            accP[u] = accQ[u];
    }
    // the last tile (possibly partial: rows past n_from are masked out here)
    if(n_tiles > 0)
    {
        const int base = (n_tiles - 1) * 32;
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            best0[u] += 32u;
            best1[u] += 32u;
#pragma unroll
            for(int i = 0; i < 16; ++i)
            {
                uint32_t g = lshl_add((uint32_t)accP[u][i], 16, T[i]);
                if(base + (int)(31u - T[i]) >= n_from)
                    g = 0u;
                best1[u] = med3_u32(best0[u], best1[u], g);
                best0[u] = max(best0[u], g);
            }
        }
    }

#pragma unroll
    for(int u = 0; u < QT; ++u)
    {
        // the two lanes of a query (h = 0, 1) saw disjoint train rows: merge their top-2
        const uint32_t p0 = (uint32_t)__shfl_xor((int)best0[u], 32), p1 = (uint32_t)__shfl_xor((int)best1[u], 32);
        const uint32_t m0 = max(best0[u], p0);
        const uint32_t m1 = max(min(best0[u], p0), max(best1[u], p1));
        const int q = q0 + (wave * QT + u) * 32 + r;
        if(h == 0 && q < n_to)
        {
            const size_t o = (size_t)pair * a.cap + q;
            const int last = 32 * n_tiles - 1;
            a.idx0[o] = (m0 >> 16) ? last - (int)(m0 & 0xFFFFu) : -1;
            a.idx1[o] = (m1 >> 16) ? last - (int)(m1 & 0xFFFFu) : -1;
            a.dist0[o] = (m0 >> 16) ? (int32_t)((513u - (m0 >> 16)) >> 1) : INT_MAX;
            a.dist1[o] = (m1 >> 16) ? (int32_t)((513u - (m1 >> 16)) >> 1) : INT_MAX;
        }
    }
}

template <int QT>
static void launch_mfma(MatchArgs a, int n_pairs, hipStream_t s)
{
    a.n_pairs = n_pairs;
    a.wg_per_pair = (a.cap + 128 * QT - 1) / (128 * QT);
    const unsigned grid = (unsigned)((n_pairs + 7) / 8) * 8u * (unsigned)a.wg_per_pair;
    hipLaunchKernelGGL((k_match_knn2_mfma<QT>), dim3(grid), dim3(256), 0, s, a);
}

void launch_match_knn2(const MatchArgs& a, int n_pairs, hipStream_t s)
{
    static const int variant = getenv("MSLAM_MATCH_VARIANT") ? atoi(getenv("MSLAM_MATCH_VARIANT")) : 2;
    if(variant == 1)
        return launch_mfma<1>(a, n_pairs, s);
    if(variant == 2)
        return launch_mfma<2>(a, n_pairs, s);
    // 8 waves x 1 query per lane, 8 rows per scalar-load batch: measured fastest of the variants tried
    // (the loop is bound by integer VALU issue: 8 xor + 8 bcnt + 4 top-2 ops per pair).
    launch_variant<8, 1, 8>(a, n_pairs, s);
}

// ratio test (orb_feature.cpp:99-105) + ordered compaction (:110-114); one workgroup per pair
__global__ __launch_bounds__(256) void k_ratio_compact(RatioArgs a)
{
    __shared__ uint32_t wcnt[4];
    const int pair = blockIdx.x;
    const int n_from = a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed;
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t o = (size_t)pair * a.cap;
    uint32_t base = 0;
    if(n_from >= 2) // fewer than two train rows: the reference indexes match[1] out of bounds (:101)
    {
        for(int q0 = 0; q0 < n_to; q0 += 256)
        {
            const int q = q0 + tid;
            bool ok = false;
            int fi = -1;
            if(q < n_to)
            {
                const int d0 = a.dist0[o + q], d1 = a.dist1[o + q];
                fi = a.idx0[o + q];
                ok = d1 <= 256 && d0 < a.thr[d1];
            }
            const unsigned long long b = __ballot(ok);
            if(lane == 0)
                wcnt[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for(int w = 0; w < 4; ++w)
            {
                if(w < wave)
                    pre += wcnt[w];
                tot += wcnt[w];
            }
            if(ok)
            {
                const uint32_t pos = base + pre + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                a.from_idx[o + pos] = fi;
                a.to_idx[o + pos] = q;
            }
            base += tot;
            __syncthreads();
        }
    }
    if(tid == 0)
        a.n_out[pair] = (int32_t)base;
}

void launch_ratio_compact(const RatioArgs& a, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_ratio_compact, dim3(n_pairs), dim3(256), 0, s, a);
}

} // namespace mslam
