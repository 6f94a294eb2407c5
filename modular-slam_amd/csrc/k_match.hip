// k_match.hip — brute-force 256-bit Hamming knn-2 matcher + ratio test.
//
// Replaces OrbOpenCvMatcher::Pimpl::match (reference orb_feature.cpp:84-117):
// BFMatcher(NORM_HAMMING).knnMatch(query = to, train = from, k = 2) then the ratio test and the
// (fromIndex = trainIdx, toIndex = queryIdx) output in query order.
//
// k_match_knn2: one lane owns one query descriptor (4 x u64 in registers).  Train descriptors are
// staged through LDS in 256-row tiles; every lane of a wave reads the SAME train row, which the LDS
// serves as a broadcast (no bank conflicts), so the inner loop is 4 xor + 4 popcount + the top-2
// update.  Scan order is ascending train index and both comparisons are strict, so on equal
// distances the lower train index ranks first — exactly batchDistance's insertion rule.
// Not HBM-bound: inputs are 2*K*32 bytes against K^2 popcount-compares.
#include "common.hpp"
#include <climits>

namespace mslam
{

constexpr int kMT = 256; // queries per workgroup == train rows per LDS tile

__global__ __launch_bounds__(kMT) void k_match_knn2(MatchArgs a)
{
    __shared__ uint4 tile[kMT * 2]; // 256 descriptors x 32 B

    const int pair = blockIdx.y;
    const int n_from = a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed;
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int q0 = blockIdx.x * kMT;
    if(q0 >= n_to)
        return;
    const int tid = threadIdx.x;
    const int q = q0 + tid;
    const uint8_t* from = a.from_desc + (long long)pair * a.from_stride;
    const uint8_t* to = a.to_desc + (long long)pair * a.to_stride;

    unsigned long long qd0 = 0, qd1 = 0, qd2 = 0, qd3 = 0;
    if(q < n_to)
    {
        const uint4* qp = reinterpret_cast<const uint4*>(to + (size_t)q * 32);
        const uint4 lo = qp[0], hi = qp[1];
        qd0 = lo.x | ((unsigned long long)lo.y << 32);
        qd1 = lo.z | ((unsigned long long)lo.w << 32);
        qd2 = hi.x | ((unsigned long long)hi.y << 32);
        qd3 = hi.z | ((unsigned long long)hi.w << 32);
    }
    int best0 = INT_MAX, best1 = INT_MAX, i0 = -1, i1 = -1;

    for(int base = 0; base < n_from; base += kMT)
    {
        const int rows = min(kMT, n_from - base);
        __syncthreads();
        // 512 uint4 per tile, 2 per thread, coalesced
        for(int k = tid; k < rows * 2; k += kMT)
            tile[k] = reinterpret_cast<const uint4*>(from + (size_t)base * 32)[k];
        __syncthreads();
        const unsigned long long* t64 = reinterpret_cast<const unsigned long long*>(tile);
#pragma unroll 4
        for(int j = 0; j < rows; ++j)
        {
            const int d = __popcll(qd0 ^ t64[4 * j]) + __popcll(qd1 ^ t64[4 * j + 1]) + __popcll(qd2 ^ t64[4 * j + 2]) +
                          __popcll(qd3 ^ t64[4 * j + 3]);
            if(d < best1)
            {
                if(d < best0)
                {
                    best1 = best0;
                    i1 = i0;
                    best0 = d;
                    i0 = base + j;
                }
                else
                {
                    best1 = d;
                    i1 = base + j;
                }
            }
        }
    }
    if(q < n_to)
    {
        const size_t o = (size_t)pair * a.cap + q;
        a.idx0[o] = i0;
        a.idx1[o] = i1;
        a.dist0[o] = best0;
        a.dist1[o] = best1;
    }
}

void launch_match_knn2(const MatchArgs& a, int n_pairs, hipStream_t s)
{
    dim3 grid((a.cap + kMT - 1) / kMT, n_pairs);
    hipLaunchKernelGGL(k_match_knn2, grid, dim3(kMT), 0, s, a);
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
