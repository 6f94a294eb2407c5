// k_match.hip — brute-force 256-bit Hamming knn-2 matcher + ratio test.
//
// Replaces OrbOpenCvMatcher::Pimpl::match (reference orb_feature.cpp:84-117):
// BFMatcher(NORM_HAMMING).knnMatch(query = to, train = from, k = 2) then the ratio test and the
// (fromIndex = trainIdx, toIndex = queryIdx) output in query order.
//
// Two kernels with identical results.  k_match_knn2_fp4 (the one that normally runs) computes the
// distances on the matrix cores; k_match_knn2 is the xor/popcount form (one lane owns one query, train
// rows arrive through the scalar unit) kept for train sets beyond the matrix-core kernel's index range.
// On equal distances the lower train index ranks first — exactly batchDistance's insertion rule (strict
// `<` on insertion while scanning train rows in ascending order).
// Not HBM-bound: inputs are 2*K*32 bytes against K^2 distance evaluations.
#include "common.hpp"
#include <climits>
#include <cstdlib>
#include <cstring>

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
// Matrix-core form of the same search.  With every descriptor bit b expanded to the FP4 (E2M1) value 2b-1,
// the dot product of two descriptors is 256 - 2*hamming, and v_mfma_f32_32x32x64_f8f6f4 computes it
// exactly (products +-1, f32 accumulation): a 32x32 tile of distances is 4 MFMAs instead of 32*32*16 xor/popcount
// lane-operations.  The accumulator starts at 258 + (31 - row) * 2^-14, so what the matrix core delivers IS the sort key
//   key(query, train j) = (dot + 258) + age * 2^-14,   age = 32 * (tiles scanned after j's tile) + 31 - (j & 31)
// — below 2^10 with 14 fraction bits: 24 significant bits, every partial sum exact in f32 — larger key = smaller distance,
// then smaller train index (batchDistance's insertion order), and the VALU only does the running top-2 on the f32 bit
// patterns (positive floats order like unsigned integers).  Keys below 2 mean "no neighbour" (1023 tiles of ageing stay
// below it).  key * 2^14 is the integer (dot / 2 + 129) << 15 | age (the dot product is even).  The age field limits this
// kernel to 32736 train rows; larger sets take the VALU kernel above.
// (Rounds 2-5 put the 2^14 into the instruction's block scale — v_mfma_scale_..., key = (dot / 2 + 129) * 2^15 + age.  The
// scaled form is two issue slots, v_mfma_ld_scale + the MFMA; with both scale operands the constant 0 the compiler selects
// the unscaled instruction, and the fraction bits carry the age instead.)
// Queries are the B operand: the accumulator then has ONE query per lane column and 16 train rows in the
// lane's 16 registers, so a query's top-2 stays in its lane (two lanes per query, merged at the end).
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int MM_TROW = 144;          // bytes of one expanded train row in LDS: 128 + 16 (conflict-free fragment reads)
constexpr int MM_MAX_TRAIN = 32736;   // 32 * 1023: the age field is 15 bits (keys stay below 2^24: exact in f32)
constexpr float MM_AGE_UNIT = 1.f / 16384.f;   // one unit of the key's age field (2^-14: the key is (dot + 258) + age * 2^-14)
constexpr float MM_TILE_AGE = 32.f * MM_AGE_UNIT; // what a key ages per train tile scanned after its own
__device__ __forceinline__ uint32_t mm_key_int(uint32_t key_bits)
{
    // the key as the integer (dot / 2 + 129) << 15 | age: an exact power-of-two multiply of a value below 2^10 with 14 fraction bits
    return (uint32_t)(__uint_as_float(key_bits) * 16384.f);
}

// SKIP (large train sets): before the 34-instruction top-2 update of a (train tile, query tile) block the wave checks whether
// ANY of its keys beats the lane's current runner-up (8 v_max3 + a compare); if none does — the common case once a few
// hundred rows have been seen — the update is skipped and the ageing of the bests (+32 per tile) is owed as a wave-uniform
// scalar.  Exact: a key that does not beat the runner-up cannot enter the top-2, and equal keys do not exist (the age
// fields differ).  Costs 10 instructions per block where it does not fire, so small train sets (cfg2) run without it.
// PIPE (QT = 4, no SKIP: the batched cfg2 / cfg4 shape): the train-tile loop in the hand-scheduled form described at the loop.
template <int QT, bool SKIP = false, bool PIPE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_match_knn2_fp4(MatchArgs a)
{
    static_assert(!PIPE || (QT == 4 && !SKIP), "the scheduled loop is written for four query tiles per wave without tile skipping");
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][32 * MM_TROW];
    __shared__ uint32_t lut[256]; // byte -> its 8 bits as FP4 +1.0 (0x2) / -1.0 (0xA)

    // XCD-aware mapping, as in k_match_knn2: every workgroup of a pair runs on the XCD (id & 7)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / a.wg_per_pair) * 8 + xcd;
    if(pair >= a.n_pairs)
        return;
    const int n_from = min(a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed, MM_MAX_TRAIN);
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    const int q0 = (slot % a.wg_per_pair) * (128 * QT);
    if(q0 >= n_to)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const uint32_t* __restrict__ from =
        reinterpret_cast<const uint32_t*>(a.from_desc + (long long)pair * a.from_stride);
    const uint8_t* to = a.to_desc + (long long)pair * a.to_stride;
    const int n_tiles = (n_from + 31) >> 5;
    // a.n_slices > 1 (a handful of pairs: the synchronous single-frame calls): blockIdx.y scans only its share of the train
    // tiles and leaves its top-2 KEYS in a.partial; k_match_merge picks the overall top-2 (the keys carry distance and train
    // index, so they compare across slices once their age field counts from the end of the whole train set)
    const int t_begin = a.n_slices > 1 ? (int)((long long)n_tiles * blockIdx.y / a.n_slices) : 0;
    const int t_end = a.n_slices > 1 ? (int)((long long)n_tiles * (blockIdx.y + 1) / a.n_slices) : n_tiles;

    if(a.n_slices > 1 && n_tiles != 0 && t_begin >= t_end)
    {
        // a slice without tiles (fewer train tiles than slices: the captured single-pair launch always has all of them): its
        // top-2 keys are "no neighbour" (0: below every real key), which k_match_merge's maxima ignore
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            const int q = q0 + (wave * QT + u) * 32 + r;
            if(h == 0 && q < n_to)
            {
                uint32_t* part = a.partial + ((size_t)pair * a.n_slices + blockIdx.y) * 2 * a.cap;
                part[q] = 0u;
                part[a.cap + q] = 0u;
            }
        }
        return;
    }
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
    {
        uint32_t e = 0;
#pragma unroll
        for(int i = 0; i < 8; ++i)
            e |= (((uint32_t)tid >> i) & 1u ? 0x2u : 0xAu) << (4 * i);
        lut[tid] = e;
    }
    __syncthreads();
    auto expand = [&](uint32_t d) -> uint4 { // 32 descriptor bits -> 32 FP4 values
        return make_uint4(lut[d & 255u], lut[(d >> 8) & 255u], lut[(d >> 16) & 255u], lut[d >> 24]);
    };

    // B fragments: lane (r, h) holds bits [64 s + 32 h, +32) of query r for k-step s
    v8i b[QT][4];
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
        const uint32_t dw[4] = {h ? lo.y : lo.x, h ? lo.w : lo.z, h ? hi.y : hi.x, h ? hi.w : hi.z};
#pragma unroll
        for(int s = 0; s < 4; ++s)
        {
            const uint4 e = expand(dw[s]);
            b[u][s] = v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
        }
    }

    // train tile staging: thread tid expands dword (tid & 7) of row (tid >> 3) of the tile.  Rows past the
    // end re-read the last row (no branch); they only ever land in the last tile, where their keys are masked.
    const int tr = tid >> 3, tw = tid & 7;
    auto fetch = [&](int t) -> uint32_t { return from[(size_t)min(t * 32 + tr, n_from - 1) * 8 + tw]; };
    auto stage = [&](int buf, uint32_t d) {
        *reinterpret_cast<uint4*>(&tile[buf][tr * MM_TROW + tw * 16]) = expand(d);
    };
    auto read_frags = [&](int buf, v8i (&af)[4]) {
#pragma unroll
        for(int s = 0; s < 4; ++s)
        {
            const uint4 e = *reinterpret_cast<const uint4*>(&tile[buf][r * MM_TROW + (2 * s + h) * 16]);
            af[s] = v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
        }
    };

    v16f cinit;
    uint32_t zero[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
    {
        cinit[i] = 258.f + (float)(31 - ((i & 3) + 8 * (i >> 2) + 4 * h)) * MM_AGE_UNIT;
        zero[i] = 0u;
    }
    uint32_t best0[QT], best1[QT];
    float owed[QT]; // SKIP: ageing not yet added to (best0, best1): 32 per skipped block, wave-uniform
#pragma unroll
    for(int u = 0; u < QT; ++u)
    {
        best0[u] = best1[u] = 0u;
        owed[u] = 0.f;
    }

    auto dots = [&](const v8i (&af)[4], int u, uint32_t (&key)[16]) {
        v16f acc = cinit;
#pragma unroll
        for(int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[s], b[u][s], acc, 4, 4, 0, 0, 0, 0);
#pragma unroll
        for(int i = 0; i < 16; ++i)
            key[i] = __float_as_uint(acc[i]);
    };
    // running top-2 of one query over the 16 train rows a lane holds of one tile
    auto top2 = [&](const uint32_t (&key)[16], uint32_t& b0, uint32_t& b1, float& lag) {
        if(SKIP)
        {
            uint32_t mx = max(max(key[0], key[1]), key[2]);
#pragma unroll
            for(int i = 3; i < 15; i += 2)
                mx = max(max(mx, key[i]), key[i + 1]);
            mx = max(mx, key[15]);
            const float age = lag + MM_TILE_AGE; // what the bests have aged by the time these keys compete
            // keys are positive floats holding integers: they order like their bit patterns
            if(__ballot(mx > __float_as_uint(__uint_as_float(b1) + age)) == 0ull)
            {
                lag = age;
                return;
            }
            b0 = __float_as_uint(__uint_as_float(b0) + age);
            b1 = __float_as_uint(__uint_as_float(b1) + age);
            lag = 0.f;
        }
        else
        {
        b0 = __float_as_uint(__uint_as_float(b0) + MM_TILE_AGE); // everything found so far is one tile older
        b1 = __float_as_uint(__uint_as_float(b1) + MM_TILE_AGE);
        }
        // Running top-2 in bundles of four keys, 5 instructions per bundle instead of 8 (round 5: this update, not the matrix
        // pipe, is what the kernel waits for).  With b0 >= b1 and two new keys x, y: the best of {b0, b1, x, y} is
        // max3(b0, x, y), the runner-up max(b1, med3(b0, x, y)) — b1 <= b0 can never exceed the median of a triple that
        // contains b0 unless it IS the runner-up — and the max with b1 can wait: two bundles' medians go into one max3.
        // Exact: the top-2 of a set does not depend on the order its elements arrive in, and keys are distinct.
        // (The builtins, not inline asm: the compiler must see these reads of an MFMA result to keep the MFMA -> VALU wait states.)
#pragma unroll
        for(int i = 0; i < 16; i += 4)
        {
            const uint32_t m01 = __float_as_uint(
                __builtin_amdgcn_fmed3f(__uint_as_float(b0), __uint_as_float(key[i]), __uint_as_float(key[i + 1])));
            b0 = max(max(b0, key[i]), key[i + 1]);
            const uint32_t m23 = __float_as_uint(
                __builtin_amdgcn_fmed3f(__uint_as_float(b0), __uint_as_float(key[i + 2]), __uint_as_float(key[i + 3])));
            b0 = max(max(b0, key[i + 2]), key[i + 3]);
            b1 = max(max(b1, m01), m23);
        }
    };

    // Software pipeline, by hand: the MFMAs of one query tile run beside the top-2 VALU work of the previous
    // one (different registers); the last query tile of a train tile is finished under the next train tile.
    uint32_t keyA[16], keyB[16], pend[16];
#pragma unroll
    for(int i = 0; i < 16; ++i)
        pend[i] = 0u; // "tile -1": keys below 2^15 never beat a real neighbour
    uint32_t d_next;
    if constexpr(PIPE)
    {
    // ---- the scheduled loop (round 6) ---------------------------------------------------------------------------------
    // A 32x32x64 FP4 MFMA keeps the SIMD's matrix core busy for 32 cycles but takes ONE issue slot; the 5-instruction top-2
    // bundle of four keys takes 20 cycles of vector issue.  The compiler, left alone, emits the sixteen MFMAs of a train tile
    // back to back (the wave then sits in front of the busy matrix core for 512 cycles) and the ~100 vector instructions of
    // the four top-2 updates after them (the matrix core idles for 430): the two pipes alternate instead of overlapping, and
    // the second wave of the SIMD, which started in phase, does the same at the same time — 957 cycles per tile and wave
    // measured against 512 of matrix-core time.  Here the order is fixed by hand (a scheduling barrier after every slot):
    // every slot is ONE MFMA plus ONE top-2 bundle of keys that were finished half a tile earlier —
    //   slots 1-8  : MFMAs of query tiles 0, 1 on train tile t   |  top-2 of query tiles 2, 3 on train tile t - 1
    //   slots 9-16 : MFMAs of query tiles 2, 3 on train tile t   |  top-2 of query tiles 0, 1 on train tile t
    // so the vector work (<= 7 instructions = 28 cycles per slot) runs in the shadow of the matrix core.  The four
    // accumulators are the key registers (no second set); a chain's dependent MFMA follows its producer by two slots.  The
    // staging of tile t + 1 (table look-ups, LDS write, the global fetch of tile t + 2) rides in slots 1 - 5; the barrier
    // stands after slot 10 and the fragments of tile t + 1 replace those of tile t as soon as their last MFMA has issued.
    if(t_end > t_begin)
    {
        v16f acc0, acc1, acc2, acc3;
#pragma unroll
        for(int i = 0; i < 16; ++i)
            acc0[i] = acc1[i] = acc2[i] = acc3[i] = 0.f; // "tile -1": keys below 2^15 never beat a real neighbour
        stage(t_begin & 1, fetch(t_begin));
        d_next = fetch(t_begin + 1);
        __syncthreads();
        v8i af[4];
        read_frags(t_begin & 1, af);
        const int n_full = t_end - 1; // the slice's last tile (the only one that can be partial) is peeled: its keys need masking
#define MM_SLOT() __builtin_amdgcn_sched_barrier(0)
#define MM_MFMA(ACC, U, S, C) ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[S], b[U][S], C, 4, 4, 0, 0, 0, 0)
#define MM_AGE(U)                                                                                                      \
    best0[U] = __float_as_uint(__uint_as_float(best0[U]) + MM_TILE_AGE);                                                      \
    best1[U] = __float_as_uint(__uint_as_float(best1[U]) + MM_TILE_AGE)
#define MM_BUNDLE(ACC, U, I)                                                                                           \
    {                                                                                                                  \
        const uint32_t m01 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(best0[U]), ACC[I], ACC[I + 1]));  \
        best0[U] = max(max(best0[U], __float_as_uint(ACC[I])), __float_as_uint(ACC[I + 1]));                           \
        const uint32_t m23 = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(best0[U]), ACC[I + 2], ACC[I + 3])); \
        best0[U] = max(max(best0[U], __float_as_uint(ACC[I + 2])), __float_as_uint(ACC[I + 3]));                       \
        best1[U] = max(max(best1[U], m01), m23);                                                                       \
        asm volatile("" : "+v"(best1[U])); /* (keeps the runner-up's max3 in this slot: it would be sunk to the loop's end) */ \
    }
#define MM_FRAG(BUF, S)                                                                                                \
    {                                                                                                                  \
        const uint4 e = *reinterpret_cast<const uint4*>(&tile[BUF][r * MM_TROW + (2 * S + h) * 16]);                   \
        af[S] = v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};                                               \
    }
#pragma unroll 1
        for(int t = t_begin; t < n_full; ++t)
        {
            const int nb = (t & 1) ^ 1;
            // slots 1 - 8
            MM_AGE(2);
            MM_BUNDLE(acc2, 2, 0);
            const uint4 ex = expand(d_next); // tile t + 1: the four table look-ups
            MM_SLOT();
            MM_MFMA(acc0, 0, 0, cinit);
            MM_SLOT();
            MM_BUNDLE(acc2, 2, 4);
            MM_SLOT();
            MM_MFMA(acc1, 1, 0, cinit);
            MM_SLOT();
            MM_BUNDLE(acc2, 2, 8);
            *reinterpret_cast<uint4*>(&tile[nb][tr * MM_TROW + tw * 16]) = ex;
            MM_SLOT();
            MM_MFMA(acc0, 0, 1, acc0);
            MM_SLOT();
            MM_BUNDLE(acc2, 2, 12);
            MM_SLOT();
            MM_MFMA(acc1, 1, 1, acc1);
            MM_SLOT();
            MM_AGE(3);
            MM_BUNDLE(acc3, 3, 0);
            d_next = fetch(t + 2);
            MM_SLOT();
            MM_MFMA(acc0, 0, 2, acc0);
            MM_SLOT();
            MM_BUNDLE(acc3, 3, 4);
            MM_SLOT();
            MM_MFMA(acc1, 1, 2, acc1);
            MM_SLOT();
            MM_BUNDLE(acc3, 3, 8);
            MM_SLOT();
            MM_MFMA(acc0, 0, 3, acc0);
            MM_SLOT();
            MM_BUNDLE(acc3, 3, 12);
            MM_SLOT();
            MM_MFMA(acc1, 1, 3, acc1);
            MM_SLOT();
            // slots 9 - 16
            MM_AGE(0);
            MM_BUNDLE(acc0, 0, 0);
            MM_SLOT();
            MM_MFMA(acc2, 2, 0, cinit);
            MM_SLOT();
            MM_BUNDLE(acc0, 0, 4);
            MM_SLOT();
            MM_MFMA(acc3, 3, 0, cinit);
            MM_SLOT();
            __syncthreads(); // tile t + 1 is staged; every wave has issued its last read of the fragments' first quarter
            MM_FRAG(nb, 0);
            MM_BUNDLE(acc0, 0, 8);
            MM_SLOT();
            MM_MFMA(acc2, 2, 1, acc2);
            MM_SLOT();
            MM_BUNDLE(acc0, 0, 12);
            MM_SLOT();
            MM_MFMA(acc3, 3, 1, acc3);
            MM_SLOT();
            MM_FRAG(nb, 1);
            MM_AGE(1);
            MM_BUNDLE(acc1, 1, 0);
            MM_SLOT();
            MM_MFMA(acc2, 2, 2, acc2);
            MM_SLOT();
            MM_BUNDLE(acc1, 1, 4);
            MM_SLOT();
            MM_MFMA(acc3, 3, 2, acc3);
            MM_SLOT();
            MM_FRAG(nb, 2);
            MM_BUNDLE(acc1, 1, 8);
            MM_SLOT();
            MM_MFMA(acc2, 2, 3, acc2);
            MM_SLOT();
            MM_BUNDLE(acc1, 1, 12);
            MM_SLOT();
            MM_MFMA(acc3, 3, 3, acc3);
            MM_SLOT();
            MM_FRAG(nb, 3);
            MM_SLOT();
        }
        if(n_full > t_begin)
        {
            // query tiles 2, 3 of the last full tile
            MM_AGE(2);
            MM_BUNDLE(acc2, 2, 0);
            MM_BUNDLE(acc2, 2, 4);
            MM_BUNDLE(acc2, 2, 8);
            MM_BUNDLE(acc2, 2, 12);
            MM_AGE(3);
            MM_BUNDLE(acc3, 3, 0);
            MM_BUNDLE(acc3, 3, 4);
            MM_BUNDLE(acc3, 3, 8);
            MM_BUNDLE(acc3, 3, 12);
        }
#undef MM_SLOT
#undef MM_MFMA
#undef MM_AGE
#undef MM_BUNDLE
#undef MM_FRAG
        {
            // the peeled last tile (its fragments are in af): keys of rows past the end are masked
            const int base = n_full * 32;
#pragma unroll
            for(int u = 0; u < QT; ++u)
            {
                dots(af, u, keyA);
#pragma unroll
                for(int i = 0; i < 16; ++i)
                    if(base + (i & 3) + 8 * (i >> 2) + 4 * h >= n_from)
                        keyA[i] = 0u;
                top2(keyA, best0[u], best1[u], owed[u]);
            }
        }
    }
    }
    else
    if(t_end > t_begin) // (an empty slice keeps "no neighbour": more slices than tiles)
    {
    stage(t_begin & 1, fetch(t_begin));
    d_next = fetch(t_begin + 1);
    __syncthreads();
    const int n_full = t_end - 1; // the slice's last tile (the only one that can be partial) is peeled: its keys need masking
    for(int t = t_begin; t < n_full; ++t)
    {
        const int buf = t & 1;
        stage(buf ^ 1, d_next);
        d_next = fetch(t + 2);
        v8i af[4];
        read_frags(buf, af);
        static_assert(QT == 2 || QT == 4, "pipeline below is written for 2 or 4 query tiles per wave");
        dots(af, 0, keyA);
        top2(pend, best0[QT - 1], best1[QT - 1], owed[QT - 1]);
        dots(af, 1, keyB);
        top2(keyA, best0[0], best1[0], owed[0]);
        if(QT == 4)
        {
            dots(af, 2, keyA);
            top2(keyB, best0[1], best1[1], owed[1]);
            dots(af, 3, keyB);
            top2(keyA, best0[2], best1[2], owed[2]);
        }
#pragma unroll
        for(int i = 0; i < 16; ++i)
            pend[i] = keyB[i];
        __syncthreads();
    }
    if(n_full > t_begin)
        top2(pend, best0[QT - 1], best1[QT - 1], owed[QT - 1]);
    {
        v8i af[4];
        read_frags(n_full & 1, af);
        const int base = n_full * 32;
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            dots(af, u, keyA);
#pragma unroll
            for(int i = 0; i < 16; ++i)
                if(base + (i & 3) + 8 * (i >> 2) + 4 * h >= n_from)
                    keyA[i] = 0u;
            top2(keyA, best0[u], best1[u], owed[u]);
        }
    }
    } // t_end > t_begin

    if(SKIP)
    {
#pragma unroll
        for(int u = 0; u < QT; ++u)
        {
            best0[u] = __float_as_uint(__uint_as_float(best0[u]) + owed[u]);
            best1[u] = __float_as_uint(__uint_as_float(best1[u]) + owed[u]);
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
        if(a.n_slices > 1)
        {
            if(h == 0 && q < n_to)
            {
                // age counted from the end of the whole train set: + 32 per tile behind this slice (exact: integers < 2^24);
                // "no neighbour" keys stay below 2^15 (at most 1023 tiles)
                const float behind = MM_TILE_AGE * (float)(n_tiles - t_end);
                uint32_t* part = a.partial + ((size_t)pair * a.n_slices + blockIdx.y) * 2 * a.cap;
                part[q] = __float_as_uint(__uint_as_float(m0) + behind);
                part[a.cap + q] = __float_as_uint(__uint_as_float(m1) + behind);
            }
            continue;
        }
        if(h == 0 && q < n_to)
        {
            const size_t o = (size_t)pair * a.cap + q;
            const int last = 32 * n_tiles - 1;
            const uint32_t k0 = mm_key_int(m0), k1 = mm_key_int(m1); // exact integers
            // key >> 15 = dot / 2 + 129 = 257 - hamming
            a.idx0[o] = (k0 >> 15) ? last - (int)(k0 & 32767u) : -1;
            a.idx1[o] = (k1 >> 15) ? last - (int)(k1 & 32767u) : -1;
            a.dist0[o] = (k0 >> 15) ? (int32_t)(257u - (k0 >> 15)) : INT_MAX;
            a.dist1[o] = (k1 >> 15) ? (int32_t)(257u - (k1 >> 15)) : INT_MAX;
        }
    }
}

// overall top-2 of a query from the slices' top-2 keys (k_match_knn2_fp4 with n_slices > 1), decoded like the kernel's own end
__global__ __launch_bounds__(256) void k_match_merge(MatchArgs a)
{
    const int pair = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n_from = min(a.from_cnt ? a.from_cnt[pair] : a.n_from_fixed, MM_MAX_TRAIN);
    const int n_to = min(a.to_cnt ? a.to_cnt[pair] : a.n_to_fixed, a.cap);
    if(q >= n_to)
        return;
    const int n_tiles = (n_from + 31) >> 5;
    uint32_t b0 = 0u, b1 = 0u;
    for(int s = 0; s < (n_tiles ? a.n_slices : 0); ++s) // (no train rows: the slices wrote nothing)
    {
        const uint32_t* part = a.partial + ((size_t)pair * a.n_slices + s) * 2 * a.cap;
        const uint32_t k0 = part[q], k1 = part[a.cap + q]; // k0 >= k1
        b1 = max(min(b0, k0), max(b1, k1));
        b0 = max(b0, k0);
    }
    const size_t o = (size_t)pair * a.cap + q;
    const int last = 32 * n_tiles - 1;
    const uint32_t k0 = mm_key_int(b0), k1 = mm_key_int(b1); // exact integers
    a.idx0[o] = (k0 >> 15) ? last - (int)(k0 & 32767u) : -1;
    a.idx1[o] = (k1 >> 15) ? last - (int)(k1 & 32767u) : -1;
    a.dist0[o] = (k0 >> 15) ? (int32_t)(257u - (k0 >> 15)) : INT_MAX;
    a.dist1[o] = (k1 >> 15) ? (int32_t)(257u - (k1 >> 15)) : INT_MAX;
}

template <int QT, bool SKIP = false, bool PIPE = false>
static void launch_fp4(MatchArgs a, int n_pairs, hipStream_t s)
{
    a.n_pairs = n_pairs;
    a.wg_per_pair = (a.cap + 128 * QT - 1) / (128 * QT);
    const unsigned grid = (unsigned)((n_pairs + 7) / 8) * 8u * (unsigned)a.wg_per_pair;
    if(a.n_slices > 1 && a.partial)
    {
        hipLaunchKernelGGL((k_match_knn2_fp4<QT, SKIP, PIPE>), dim3(grid, a.n_slices), dim3(256), 0, s, a);
        // (merging inside k_ratio_compact instead of a launch of its own measured slower: 51.6 vs 49.4 us per call — that kernel
        // is one workgroup walking the queries in order)
        hipLaunchKernelGGL(k_match_merge, dim3((a.cap + 255) / 256, n_pairs), dim3(256), 0, s, a);
        return;
    }
    a.n_slices = 1;
    hipLaunchKernelGGL((k_match_knn2_fp4<QT, SKIP, PIPE>), dim3(grid), dim3(256), 0, s, a);
}

int launch_match_knn2(const MatchArgs& a, int n_pairs, hipStream_t s)
{
    // matrix-core kernel (4 query tiles per wave: 0.117 ms per 250 x 1900^2 pairs against 0.48 ms for the
    // VALU kernel) whenever the train side fits its 14-bit age field; the VALU kernel otherwise
    // (8 waves x 1 query per lane, 8 rows per scalar-load batch: the fastest of its variants).
    // a.popcount_only (mslam_hip_set_matcher / MSLAM_HIP_MATCHER=popcount at context creation) selects the
    // xor/popcount kernel everywhere (the form BASELINE.json's north_star describes; same results, 0.48 instead of
    // 0.12 ms per 250 x 1900^2 pairs)
    const int max_train = a.from_cnt ? a.cap_from : a.n_from_fixed;
    if(max_train <= MM_MAX_TRAIN && !a.popcount_only)
    {
        // a handful of pairs (the synchronous single-frame calls): two query tiles per wave, twice as many waves —
        // the kernel's latency is what counts there (39 -> 13 us for one 1900 x 1900 pair), not its throughput
        const int skip_from = [] { const char* e = getenv("MSLAM_HIP_MATCH_SKIP_FROM"); return e ? atoi(e) : 6000; }(); // (read per launch: a test switches it)
        // MSLAM_HIP_MATCH_PIPE=0 (read per launch: A/B runs and a test switch it) selects the round-5 loops: the compiler-
        // scheduled one, and from `skip_from` train rows on the tile-skipping one (cfg5: 0.64 ms per 64 x 10058^2 pairs; the
        // scheduled loop does the same work in 0.57 ms without skipping anything, so it takes every size by default)
        const bool pipe = [] { const char* e = getenv("MSLAM_HIP_MATCH_PIPE"); return !e || atoi(e) != 0; }();
        if(n_pairs <= 4)
            launch_fp4<2>(a, n_pairs, s);
        else if(pipe)
            launch_fp4<4, false, true>(a, n_pairs, s);
        else if(max_train >= skip_from) // long scans (cfg5: 10 k train rows): most late blocks cannot change a top-2
            launch_fp4<4, true>(a, n_pairs, s);
        else
            launch_fp4<4>(a, n_pairs, s);
        return 1;
    }
    launch_variant<8, 1, 8>(a, n_pairs, s);
    return 2;
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

// The single synchronous call (mslam_hip_match): merge of the slices' top-2 keys (k_match_merge), ratio test and ordered
// compaction (k_ratio_compact) in ONE launch of one 1024-thread workgroup — two launches of a few microseconds each were a
// quarter of the call's GPU time.  Pair 0 only; the raw top-2 arrays are not written (mslam_hip_match does not return them).
__global__ __launch_bounds__(1024) void k_merge_ratio(MatchArgs a, RatioArgs r)
{
    __shared__ uint32_t wcnt[16];
    const int n_from = min(a.from_cnt ? a.from_cnt[0] : a.n_from_fixed, MM_MAX_TRAIN);
    const int n_to = min(a.to_cnt ? a.to_cnt[0] : a.n_to_fixed, a.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (n_from + 31) >> 5;
    const int last = 32 * n_tiles - 1;
    uint32_t base = 0;
    if(n_from >= 2) // fewer than two train rows: the reference indexes match[1] out of bounds (orb_feature.cpp:101)
    {
        for(int q0 = 0; q0 < n_to; q0 += 1024)
        {
            const int q = q0 + tid;
            bool ok = false;
            int fi = -1;
            if(q < n_to)
            {
                uint32_t b0 = 0u, b1 = 0u;
                for(int s = 0; s < a.n_slices; ++s)
                {
                    const uint32_t* part = a.partial + (size_t)s * 2 * a.cap;
                    const uint32_t k0 = part[q], k1 = part[a.cap + q]; // k0 >= k1
                    b1 = max(min(b0, k0), max(b1, k1));
                    b0 = max(b0, k0);
                }
                const uint32_t k0 = mm_key_int(b0), k1 = mm_key_int(b1); // exact integers
                // key >> 15 = 257 - hamming; a key without a neighbour decodes to distance INT_MAX, which fails the test
                fi = (k0 >> 15) ? last - (int)(k0 & 32767u) : -1;
                const int d0 = (k0 >> 15) ? (int)(257u - (k0 >> 15)) : INT_MAX, d1 = (k1 >> 15) ? (int)(257u - (k1 >> 15)) : INT_MAX;
                ok = d1 <= 256 && d0 < r.thr[d1];
            }
            const unsigned long long b = __ballot(ok);
            if(lane == 0)
                wcnt[wave] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for(int w = 0; w < 16; ++w)
            {
                const uint32_t x = wcnt[w];
                pre += w < wave ? x : 0u;
                tot += x;
            }
            if(ok)
            {
                const uint32_t pos = base + pre + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
                r.from_idx[pos] = fi;
                r.to_idx[pos] = q;
            }
            base += tot;
            __syncthreads();
        }
    }
    if(tid == 0)
        r.n_out[0] = (int32_t)base;
}

// matcher + merge + ratio test of ONE pair with sliced train tiles, for mslam_hip_match; returns false when the arguments
// do not fit this form (the caller then takes launch_match_knn2 + launch_ratio_compact)
bool launch_match_ratio_single(const MatchArgs& a, const RatioArgs& r, hipStream_t s)
{
    const int max_train = a.from_cnt ? a.cap_from : a.n_from_fixed;
    if(max_train > MM_MAX_TRAIN || a.popcount_only || a.n_slices <= 1 || !a.partial)
        return false;
    MatchArgs m = a;
    m.n_pairs = 1;
    m.wg_per_pair = (m.cap + 128 * 2 - 1) / (128 * 2);
    const unsigned grid = 8u * (unsigned)m.wg_per_pair;
    hipLaunchKernelGGL((k_match_knn2_fp4<2, false>), dim3(grid, m.n_slices), dim3(256), 0, s, m);
    hipLaunchKernelGGL(k_merge_ratio, dim3(1), dim3(1024), 0, s, m, r);
    return true;
}

void launch_ratio_compact(const RatioArgs& a, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_ratio_compact, dim3(n_pairs), dim3(256), 0, s, a);
}

} // namespace mslam
