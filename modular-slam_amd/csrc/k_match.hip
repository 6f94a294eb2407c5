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

    const int pair = blockIdx.y;
    const int n_from = min(a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed, 65535);
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int q0 = blockIdx.x * (64 * QL);
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
static void launch_variant(const MatchArgs& a, int n_pairs, hipStream_t s)
{
    dim3 grid((a.cap + 64 * QL - 1) / (64 * QL), n_pairs);
    hipLaunchKernelGGL((k_match_knn2<MW, QL, UNR>), grid, dim3(64 * MW), 0, s, a);
}

void launch_match_knn2(const MatchArgs& a, int n_pairs, hipStream_t s)
{
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
