// k_bow.hip — DBoW3 bag of words on the GPU: vocabulary-tree descent, BoW vectors, L1 scoring, database.
//
// Replaces what the reference wires OrbRelocalizer for (orb_relocalizer.cpp:26-50) and the DBoW3
// library code behind it (rmsalinas/DBow3 master; only Vocabulary.cpp/.h text is in the reference tree,
// inside conan_recipes/dbow3/dbow3.patch):
//   Vocabulary::fromStream                              dbow3.patch:2544-2651   (host parser below)
//   Vocabulary::transform(feature, word_id, weight)     dbow3.patch:1760-1860   (k_bow_descend)
//   Vocabulary::transform(features, BowVector)          dbow3.patch:1432-1530   (k_bow_vector)
//   BowVector::addWeight / addIfNotExist / normalize, L1Scoring::score, Database::add / queryL1
//   (DBoW3 sources not in the tree; restated from the published algorithm)               (k_bow_vector, k_bow_score)
//
// Bit-exactness notes.  Word assignment is integer-only (256-bit Hamming argmin, strict `<` so the
// first child wins ties).  BoW values and scores are f64 and every sum is evaluated in the order the
// reference's std::map iteration / feature loop fixes: a word hit c times is w+w+...+w accumulated
// sequentially, the L1 norm is a sequential sum over ascending word ids, a score is a sequential sum
// over the common words in ascending order.  Those chains are short (<= keypoints per frame) and run
// one per frame / per (frame, entry) pair, so thousands of them run in parallel across the batch.
#include "context.hpp"

#include <algorithm>
#include <new>
#include <cmath>
#include <cstring>

namespace mslam
{
int qlz_decode_stream(const uint8_t* src, size_t size, uint32_t n_packets, std::vector<uint8_t>& out); // quicklz_decode.hip

constexpr int kBowGroup = 16; // lanes cooperating on one descriptor: one child each, <= 16 per step

struct BowState
{
    int k = 0, L = 0, scoring = 0, weighting = 0;
    uint32_t n_nodes = 0, n_words = 0;
    int max_children = 0;
    // device tree, indexed by "slot": the children of a node occupy consecutive slots in stream order
    uint4* d_desc = nullptr;        // [n_nodes][2]
    uint32_t* d_first = nullptr;    // first child slot (0 = leaf)
    uint32_t* d_nchild = nullptr;
    uint32_t* d_word = nullptr;     // word id of a leaf
    double* d_weight = nullptr;
    // per-feature scratch for a batch
    uint32_t* d_fword = nullptr;    // [B][cap]
    double* d_fweight = nullptr;    // [B][cap]
    // BoW vectors of the current batch (slot B = the host-pointer query)
    uint32_t* d_bwords = nullptr;   // [B+1][cap]
    double* d_bvalues = nullptr;    // [B+1][cap]
    int32_t* d_bn = nullptr;        // [B+1]
    // database ring: R live entries + B in flight
    int R = 64, RP = 0;
    uint32_t* d_rwords = nullptr;   // [RP][cap]
    double* d_rvalues = nullptr;
    int32_t* d_rn = nullptr;        // [RP]
    long long next_id = 0;
    // scoring outputs
    double* d_contrib = nullptr;    // [B+1][R][cap] matched terms of every (query, entry) pair, ascending word order
    int32_t* d_match_cnt = nullptr; // [B+1][R]
    double* d_scores = nullptr;     // [B+1][R]   score against entry (id_t - 1 - j), -1 when absent / no common word
    int32_t* d_best_entry = nullptr; // [B+1]
    double* d_best_score = nullptr;
    uint8_t* d_hdesc = nullptr;     // host-pointer query descriptors [cap][32]
    // flat mode (SURVEY.md §8d bow_flat): the leaves' descriptors / weights in word-id order, per-feature best keys
    uint32_t* d_leaf_desc = nullptr; // [n_words][8]
    double* d_leaf_weight = nullptr; // [n_words]
    uint32_t* d_fbest = nullptr;     // [B+1][cap] (distance << 20) | word
    int flat = 0;                    // MSLAM_BOW_ASSIGN_*
    // inverted file of the database (DBoW3 Database: m_ifile, word -> (entry, value) rows).  Postings live in an
    // append-only log, entry after entry; the rows of a word are a linked list threaded through the log
    // (ix_prev), newest first, anchored at ix_head[word].  Index 0 is the null posting.
    uint32_t* ix_head = nullptr;     // [n_words]
    uint32_t* ix_entry = nullptr;    // [ix_cap_postings + 1]
    uint32_t* ix_prev = nullptr;
    double* ix_value = nullptr;
    uint32_t* ix_size = nullptr;     // device counter: postings used
    uint8_t* ix_removed = nullptr;   // [ix_max_entries]
    // query scratch
    uint32_t* ix_cnt = nullptr;      // [ix_max_entries] common words per entry
    uint32_t* ix_ofs = nullptr;      // [ix_max_entries + 1]
    uint32_t* ix_fill = nullptr;     // [ix_max_entries]
    uint32_t* ix_keys = nullptr;     // [ix_cap_terms] rank of the query word
    double* ix_vals = nullptr;       // [ix_cap_terms] the L1 term
    double* ix_scores = nullptr;     // [ix_max_entries]
    uint32_t* ix_total = nullptr;    // device: number of terms of the last query (for the capacity check)
    long long ix_max_entries = 0;
    size_t ix_cap_postings = 0, ix_cap_terms = 0;
    int cap = 0, B = 0;
};

static void bow_free(BowState* b)
{
    void* bufs[] = {b->d_desc,   b->d_first,   b->d_nchild, b->d_word,   b->d_weight,     b->d_fword,      b->d_fweight,
                    b->d_bwords, b->d_bvalues, b->d_bn,     b->d_rwords, b->d_rvalues,    b->d_rn,         b->d_scores,
                    b->d_best_entry, b->d_best_score, b->d_hdesc, b->d_contrib, b->d_match_cnt,
                    b->d_leaf_desc,  b->d_leaf_weight, b->d_fbest, b->ix_head, b->ix_entry, b->ix_prev, b->ix_value,
                    b->ix_size, b->ix_removed, b->ix_cnt, b->ix_ofs, b->ix_fill, b->ix_keys, b->ix_vals, b->ix_scores,
                    b->ix_total};
    for(void* p : bufs)
        if(p)
            (void)hipFree(p);
}
void bow_destroy(BowState* b)
{
    if(!b)
        return;
    bow_free(b);
    delete b;
}

// ---- kernels ------------------------------------------------------------------------------------------

// 16 lanes walk kBowDepth descriptors down the tree side by side; lane c scores child c of each one's current node.
// A level is one dependent memory round trip (from the fifth level on the nodes of a 10^6-word vocabulary come from the
// memory side: 1-2 us), so what sets the kernel's time is how many descents are in flight: with one per lane group it
// ran at that latency (0.40 ms per 1000 frames at full occupancy); kBowDepth of them issue their loads together.
// (2: 0.69 -> 0.48 ms in place, cfg3 step -0.8 %; 4 and 8 cost more in registers / occupancy than they add: 0.82 / 1.77 ms.)
#ifndef MSLAM_BOW_DEPTH
#define MSLAM_BOW_DEPTH 2
#endif
constexpr int kBowDepth = MSLAM_BOW_DEPTH;
__global__ __launch_bounds__(256) void k_bow_descend(const uint8_t* __restrict__ desc, long long desc_stride,
                                                     const int32_t* __restrict__ counts, int n_fixed, int cap,
                                                     const uint4* __restrict__ tdesc, const uint32_t* __restrict__ first,
                                                     const uint32_t* __restrict__ nchild,
                                                     const uint32_t* __restrict__ word, const double* __restrict__ weight,
                                                     uint32_t* __restrict__ out_word, double* __restrict__ out_weight)
{
    const int frame = blockIdx.y;
    const int n = min(counts ? counts[frame] : n_fixed, cap);
    const int sub = threadIdx.x & (kBowGroup - 1);
    const int i0 = ((blockIdx.x * 256 + threadIdx.x) / kBowGroup) * kBowDepth;
    if(i0 >= n)
        return; // whole 16-lane group leaves together
    uint4 qa[kBowDepth], qb[kBowDepth];
    uint32_t node[kBowDepth], fc[kBowDepth], nc[kBowDepth];
    const uint32_t fc0 = first[0], nc0 = nchild[0];
#pragma unroll
    for(int d = 0; d < kBowDepth; ++d)
    {
        const int i = min(i0 + d, n - 1); // (a group's surplus slots repeat its last descriptor; nothing is stored for them)
        const uint4* q = reinterpret_cast<const uint4*>(desc + (long long)frame * desc_stride + (size_t)i * 32);
        qa[d] = q[0], qb[d] = q[1];
        node[d] = 0, fc[d] = fc0, nc[d] = nc0;
    }
    // One dependent memory round trip per tree level: lane c fetches child c's descriptor AND child c's own
    // (first child, child count) together; after the argmin the winner's pair comes from the winning lane by
    // shuffle instead of from memory (first[] -> children -> nchild[] used to be three round trips per level).
    for(;;)
    {
        bool any = false;
#pragma unroll
        for(int d = 0; d < kBowDepth; ++d)
            any = any || nc[d] != 0; // (uniform over the group)
        if(!any)
            break;
        uint32_t best[kBowDepth], best_fc[kBowDepth], best_nc[kBowDepth];
        uint4 ta[kBowDepth], tb[kBowDepth];
        uint32_t cf[kBowDepth], cn[kBowDepth];
        // every descent's loads first ...
#pragma unroll
        for(int d = 0; d < kBowDepth; ++d)
        {
            best[d] = 0xFFFFFFFFu; // (distance << 16) | child index: min == first child with the least distance
            best_fc[d] = 0, best_nc[d] = 0;
            ta[d] = tb[d] = make_uint4(0, 0, 0, 0);
            cf[d] = cn[d] = 0;
            if((uint32_t)sub < nc[d])
            {
                const uint4* t = tdesc + (size_t)(fc[d] + sub) * 2;
                ta[d] = t[0], tb[d] = t[1];
                cf[d] = first[fc[d] + sub], cn[d] = nchild[fc[d] + sub];
            }
        }
        // ... then their scores (nodes with more than 16 children take further rounds, one descent at a time)
#pragma unroll
        for(int d = 0; d < kBowDepth; ++d)
        {
            if(nc[d] == 0)
                continue;
            for(uint32_t c0 = 0; c0 < nc[d]; c0 += kBowGroup)
            {
                const uint32_t c = c0 + sub;
                if(c < nc[d])
                {
                    if(c0 != 0)
                    {
                        const uint4* t = tdesc + (size_t)(fc[d] + c) * 2;
                        ta[d] = t[0], tb[d] = t[1];
                        cf[d] = first[fc[d] + c], cn[d] = nchild[fc[d] + c];
                    }
                    const uint32_t dist = __popc(qa[d].x ^ ta[d].x) + __popc(qa[d].y ^ ta[d].y) + __popc(qa[d].z ^ ta[d].z) +
                                          __popc(qa[d].w ^ ta[d].w) + __popc(qb[d].x ^ tb[d].x) + __popc(qb[d].y ^ tb[d].y) +
                                          __popc(qb[d].z ^ tb[d].z) + __popc(qb[d].w ^ tb[d].w);
                    const uint32_t key = (dist << 16) | c;
                    if(key < best[d])
                    {
                        best[d] = key;
                        best_fc[d] = cf[d];
                        best_nc[d] = cn[d];
                    }
                }
            }
            // all-reduce over the group's 16 lanes = one DPP row: four rotate-and-min steps on the vector ALU (as __shfl_xor
            // these were trips through the LDS crossbar, six per level with the two winner lookups, on a kernel that runs at
            // the latency of its dependent chain).  Keys are unique (they carry the child index): exactly one lane holds the
            // winner, and its (first child, child count) reach the others by the same rotations of a masked value.
            static_assert(kBowGroup == 16, "the DPP row rotations below reduce over 16 lanes");
            auto ror = [](uint32_t v, auto sh) {
                constexpr int S = decltype(sh)::value;
                return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + S, 0xF, 0xF, false); // row_ror:S
            };
            using std::integral_constant;
            uint32_t all = best[d];
            all = min(all, ror(all, integral_constant<int, 1>{}));
            all = min(all, ror(all, integral_constant<int, 2>{}));
            all = min(all, ror(all, integral_constant<int, 4>{}));
            all = min(all, ror(all, integral_constant<int, 8>{}));
            const bool winner = best[d] == all;
            uint32_t wf = winner ? best_fc[d] : 0u, wn = winner ? best_nc[d] : 0u;
            wf |= ror(wf, integral_constant<int, 1>{}), wn |= ror(wn, integral_constant<int, 1>{});
            wf |= ror(wf, integral_constant<int, 2>{}), wn |= ror(wn, integral_constant<int, 2>{});
            wf |= ror(wf, integral_constant<int, 4>{}), wn |= ror(wn, integral_constant<int, 4>{});
            wf |= ror(wf, integral_constant<int, 8>{}), wn |= ror(wn, integral_constant<int, 8>{});
            node[d] = fc[d] + (all & 0xFFFFu);
            fc[d] = wf;
            nc[d] = wn;
        }
    }
    if(sub < kBowDepth && i0 + sub < n)
    {
        uint32_t nd = node[0];
#pragma unroll
        for(int d = 1; d < kBowDepth; ++d)
            nd = sub == d ? node[d] : nd;
        const size_t o = (size_t)frame * cap + i0 + sub;
        out_word[o] = word[nd];
        out_weight[o] = weight[nd];
    }
}

// ---- flat word assignment (north_star: "a batched descriptor-vs-vocabulary Hamming kernel"; SURVEY.md §8d bow_flat) ----
// word(d) = argmin over ALL leaves of the 256-bit Hamming distance, ties to the lower word id — the exhaustive
// search the tree descent approximates.  Same structure as the xor/popcount matcher (k_match.hip): one lane owns one
// descriptor, a leaf row is wave-uniform and arrives through the scalar unit (s_load_dwordx8), 8 xor + 8 v_bcnt per
// pair; the running minimum is a packed key (distance << 20 | word), unique per word.  The vocabulary is cut into
// slices (gridDim.y) so that 10^6 leaves fill the chip even for one frame; slices merge with atomicMin on the key,
// which is order-independent.  Not HBM-bound: n x V distance evaluations against (n + V) x 32 bytes.
constexpr int kFlatWaves = 4;

__device__ __forceinline__ uint32_t bow_bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

__global__ __launch_bounds__(64 * kFlatWaves) void k_bow_flat(const uint8_t* __restrict__ desc, long long desc_stride,
                                                               const int32_t* __restrict__ counts, int n_fixed, int cap,
                                                               const uint32_t* __restrict__ leaves, int n_words,
                                                               int slice, uint32_t* __restrict__ best)
{
    const int frame = blockIdx.z;
    const int n = min(counts ? counts[frame] : n_fixed, cap);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 64 + lane;
    if(blockIdx.x * 64 >= n)
        return;
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if(q < n)
    {
        const uint4* qp = reinterpret_cast<const uint4*>(desc + (long long)frame * desc_stride + (size_t)q * 32);
        qa = qp[0];
        qb = qp[1];
    }
    const int s0 = blockIdx.y * slice, s1 = min(n_words, s0 + slice);
    const int chunk = (s1 - s0 + kFlatWaves - 1) / kFlatWaves;
    const int j0 = s0 + wave * chunk, j1 = min(s1, j0 + chunk);
    uint32_t b0 = 0xFFFFFFFFu;
    auto row = [&](int j) {
        const uint32_t* __restrict__ t = leaves + (size_t)j * 8; // wave-uniform address -> s_load_dwordx8
        uint32_t d = __popc(qa.x ^ t[0]);
        d = bow_bcnt_acc(qa.y ^ t[1], d);
        d = bow_bcnt_acc(qa.z ^ t[2], d);
        d = bow_bcnt_acc(qa.w ^ t[3], d);
        d = bow_bcnt_acc(qb.x ^ t[4], d);
        d = bow_bcnt_acc(qb.y ^ t[5], d);
        d = bow_bcnt_acc(qb.z ^ t[6], d);
        d = bow_bcnt_acc(qb.w ^ t[7], d);
        b0 = min(b0, (d << 20) | (uint32_t)j);
    };
    int j = j0;
    for(; j + 8 <= j1; j += 8)
    {
#pragma unroll
        for(int k = 0; k < 8; ++k)
            row(j + k);
    }
    for(; j < j1; ++j)
        row(j);
    if(q < n && b0 != 0xFFFFFFFFu)
        atomicMin(&best[(size_t)frame * cap + q], b0);
}

__global__ __launch_bounds__(256) void k_bow_flat_finish(const uint32_t* __restrict__ best, const int32_t* __restrict__ counts,
                                                         int n_fixed, int cap, const double* __restrict__ leaf_weight,
                                                         uint32_t* __restrict__ out_word, double* __restrict__ out_weight)
{
    const int frame = blockIdx.y;
    const int n = min(counts ? counts[frame] : n_fixed, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= n)
        return;
    const size_t o = (size_t)frame * cap + i;
    const uint32_t w = best[o] & 0xFFFFFu;
    out_word[o] = w;
    out_weight[o] = leaf_weight[w];
}

constexpr int VT = 512;

// One workgroup per frame: (word, feature index) pairs -> ascending unique words with accumulated,
// normalised values.
__global__ __launch_bounds__(VT) void k_bow_vector(const uint32_t* __restrict__ fword, const double* __restrict__ fweight,
                                                   const int32_t* __restrict__ counts, int n_fixed, int cap, int npow2,
                                                   int weighting, int scoring, uint32_t* __restrict__ bwords,
                                                   double* __restrict__ bvalues, int32_t* __restrict__ bn, int slot0)
{
    extern __shared__ unsigned long long keys[]; // npow2 sort keys, reused as f64 values afterwards
    __shared__ uint32_t wsum[VT / 64];
    __shared__ double sh_norm;

    const int frame = blockIdx.x;
    const int n = min(counts ? counts[frame] : n_fixed, cap);
    const int tid = threadIdx.x;
    const uint32_t* fw = fword + (size_t)frame * cap;
    const double* fv = fweight + (size_t)frame * cap;
    uint32_t* ow = bwords + (size_t)(slot0 + frame) * cap;
    double* ov = bvalues + (size_t)(slot0 + frame) * cap;

    // sort size: the smallest power of two (>= 64) holding this frame's n features
    int np2 = 64;
    while(np2 < n)
        np2 <<= 1;
    npow2 = min(npow2, np2);
    for(int i = tid; i < npow2; i += VT)
    {
        unsigned long long key = ~0ull;
        if(i < n && fv[i] > 0) // "not stopped" (dbow3.patch transform: if (w > 0))
            key = ((unsigned long long)fw[i] << 32) | (uint32_t)i;
        keys[i] = key;
    }
    __syncthreads();
    // bitonic sort, ascending: by word id, then by feature index (= the order of the reference's loop)
    for(int size = 2; size <= npow2; size <<= 1)
        for(int stride = size >> 1; stride > 0; stride >>= 1)
        {
            for(int t = tid; t < (npow2 >> 1); t += VT)
            {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if((a > b) == up)
                {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    // heads of equal-word runs -> output positions
    uint32_t running = 0;
    const bool tf = weighting == MSLAM_BOW_TF || weighting == MSLAM_BOW_TF_IDF;
    for(int base = 0; base < npow2; base += VT)
    {
        const int i = base + tid;
        const unsigned long long key = i < npow2 ? keys[i] : ~0ull;
        const bool valid = key != ~0ull;
        const bool head = valid && (i == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(key >> 32));
        // workgroup exclusive scan of the head flags
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long bal = __ballot(head);
        if(lane == 0)
            wsum[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for(int w = 0; w < VT / 64; ++w)
        {
            if(w < wave)
                pre += wsum[w];
            tot += wsum[w];
        }
        if(head)
        {
            const uint32_t pos = running + pre + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            const uint32_t wid = (uint32_t)(key >> 32);
            const double w = fv[(uint32_t)key];
            double v = w; // addWeight: first hit inserts w, later hits += w (same word => same weight)
            if(tf)
                for(int j = i + 1; j < npow2 && (uint32_t)(keys[j] >> 32) == wid && keys[j] != ~0ull; ++j)
                    v += w;
            ow[pos] = wid;
            ov[pos] = v;
        }
        running += tot;
        __syncthreads();
    }
    const uint32_t m = running;
    __syncthreads();
    // normalisation (ScoringObject::mustNormalize): L1 for L1/CHI/KL/BHATTACHARYYA, L2 for L2, none for DOT
    double* vals = reinterpret_cast<double*>(keys);
    for(uint32_t i = tid; i < m; i += VT)
        vals[i] = ov[i];
    __syncthreads();
    const bool must = scoring != MSLAM_BOW_DOT_PRODUCT;
    if(tid == 0)
    {
        double norm = 0.0;
        if(must)
        {
            if(scoring == MSLAM_BOW_L2_NORM)
            {
                for(uint32_t i = 0; i < m; ++i)
                    norm += vals[i] * vals[i];
                norm = sqrt(norm);
            }
            else
                for(uint32_t i = 0; i < m; ++i)
                    norm += fabs(vals[i]);
        }
        else if(tf && m > 0)
            norm = (double)m; // "unnecessary when normalizing": divide by the number of words
        sh_norm = norm;
        bn[slot0 + frame] = (int32_t)m;
    }
    __syncthreads();
    const double norm = sh_norm;
    if(norm > 0.0)
        for(uint32_t i = tid; i < m; i += VT)
            ov[i] = vals[i] / norm;
}

// copy the batch's vectors into their database ring slots
__global__ __launch_bounds__(256) void k_bow_commit(const uint32_t* __restrict__ bwords, const double* __restrict__ bvalues,
                                                    const int32_t* __restrict__ bn, int cap, long long base_id, int RP,
                                                    uint32_t* __restrict__ rwords, double* __restrict__ rvalues,
                                                    int32_t* __restrict__ rn)
{
    const int t = blockIdx.x;
    const int slot = (int)((base_id + t) % RP);
    const int n = bn[t];
    for(int i = threadIdx.x; i < n; i += 256)
    {
        rwords[(size_t)slot * cap + i] = bwords[(size_t)t * cap + i];
        rvalues[(size_t)slot * cap + i] = bvalues[(size_t)t * cap + i];
    }
    if(threadIdx.x == 0)
        rn[slot] = n;
}

constexpr int kScoreBlocksPerFrame = 2; // the query hash is rebuilt per block; two blocks per frame keep 2 blocks/CU busy
constexpr uint32_t kIdxBits = 14, kIdxMask = (1u << kIdxBits) - 1u; // k_bow_score's table word: index + 1 (cap <= 16383) | Bloom field << 14
constexpr int kScoreWaves = 16; // waves per workgroup; the workgroup owns one query frame, each wave a share of the entries

// L1Scoring::score of query vector `qslot + blockIdx.y` against the window of database entries that
// precede it.  The query's words go into an LDS hash table once per workgroup; each wave then streams
// one database entry through it (coalesced loads, one probe per element).  The per-element terms
// |v-w|-|v|-|w| are exact and order-free; their SUM must follow the reference's ascending-word order,
// so every 64-element chunk adds its matches in lane order (ballot + readlane), chunk after chunk.
__global__ __launch_bounds__(64 * kScoreWaves) void k_bow_score(
    const uint32_t* __restrict__ bwords, const double* __restrict__ bvalues, const int32_t* __restrict__ bn, int qslot,
    int cap, const uint32_t* __restrict__ rwords, const double* __restrict__ rvalues, const int32_t* __restrict__ rn,
    long long base_id, int per_frame_id, int R, int RP, int slots, double* __restrict__ contrib,
    int32_t* __restrict__ match_cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sm_raw[];
    double* qv = reinterpret_cast<double*>(sm_raw);                     // [cap] query values
    uint32_t* table = reinterpret_cast<uint32_t*>(qv + cap);            // [slots] hash table of (index + 1)
    uint32_t* qw = table + slots;                                       // [cap] query words

    const int t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long id_t = base_id + (per_frame_id ? t : 0); // entries with id < id_t are visible
    const uint32_t* w1 = bwords + (size_t)(qslot + t) * cap;
    const double* v1 = bvalues + (size_t)(qslot + t) * cap;
    const int n1 = bn[qslot + t];
    const uint32_t hmask = (uint32_t)slots - 1;

    for(int i = tid; i < slots; i += 64 * kScoreWaves)
        table[i] = 0;
    for(int i = tid; i < n1; i += 64 * kScoreWaves)
    {
        qw[i] = w1[i];
        qv[i] = v1[i];
    }
    __syncthreads();
    // A slot holds the index (+ 1) of the query word stored there in its low 14 bits and, in bits 14 .. 29, a 16-bit Bloom
    // field for the words whose HOME slot it is: a database word that is not in the query — most of them — is told so by
    // the first read of its probe (its bit of the home slot's field is clear; false positives ~ load / 16) instead of
    // walking to the next empty slot, and the divergent probe loop below only runs for words that are (almost surely) there.
    for(int i = tid; i < n1; i += 64 * kScoreWaves)
    {
        const uint32_t hw = qw[i] * 2654435761u;
        const uint32_t home = hw >> 7 & hmask;
        uint32_t h = home;
        for(;;)
        {
            const uint32_t e = table[h];
            if((e & kIdxMask) != 0u)
            {
                h = (h + 1) & hmask;
                continue;
            }
            if(atomicCAS(&table[h], e, e | ((uint32_t)i + 1u)) == e)
                break; // (a failed exchange: another thread took the slot or set one of its Bloom bits — look again)
        }
        atomicOr(&table[home], 1u << (kIdxBits + (hw >> 28)));
    }
    __syncthreads();

    for(int j = blockIdx.x * kScoreWaves + wave; j < R; j += gridDim.x * kScoreWaves)
    {
        const long long e_id = id_t - 1 - j;
        const size_t oj = (size_t)(qslot + t) * R + j;
        if(e_id < 0)
        {
            if(lane == 0)
                match_cnt[oj] = -1; // entry does not exist
            continue;
        }
        const int slot = (int)(e_id % RP);
        const uint32_t* w2 = rwords + (size_t)slot * cap;
        const double* v2 = rvalues + (size_t)slot * cap;
        const int n2 = rn[slot];
        // matched terms are written, in ascending word order, to contrib[t][j][k]
        double* out = contrib + oj * cap;
        uint32_t n_match = 0;
        uint32_t word_n = lane < n2 ? w2[lane] : 0u; // the next chunk's loads are issued before this chunk is probed
        double wi_n = lane < n2 ? v2[lane] : 0.0;
        for(int base = 0; base < n2; base += 64)
        {
            const int i = base + lane;
            const uint32_t word = word_n;
            const double wi = wi_n;
            if(i + 64 < n2)
            {
                word_n = w2[i + 64];
                wi_n = v2[i + 64];
            }
            double c = 0;
            bool found = false;
            if(i < n2)
            {
                const uint32_t hw = word * 2654435761u;
                uint32_t h = hw >> 7 & hmask;
                uint32_t e = table[h];
                if((e >> (kIdxBits + (hw >> 28))) & 1u) // else: no query word with this home slot and Bloom bit
                    for(;;)
                    {
                        const uint32_t idx = e & kIdxMask;
                        if(idx == 0)
                            break;
                        if(qw[idx - 1] == word)
                        {
                            const double vi = qv[idx - 1];
                            c = fabs(vi - wi) - fabs(vi) - fabs(wi);
                            found = true;
                            break;
                        }
                        h = (h + 1) & hmask;
                        e = table[h];
                    }
            }
            const unsigned long long m = __ballot(found);
            if(found)
                out[n_match + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = c;
            n_match += (uint32_t)__popcll(m);
        }
        if(lane == 0)
            match_cnt[oj] = (int32_t)n_match;
    }
}

// Sequential, ascending-word-order sum of the matched terms: one LANE per (query, entry), so the 64
// dependent f64 chains of a query run side by side; reads are coalesced (entry index is fastest).
__global__ __launch_bounds__(64) void k_bow_sum(const double* __restrict__ contrib, const int32_t* __restrict__ match_cnt,
                                                int qslot, int cap, int R, double* __restrict__ scores)
{
    const int t = blockIdx.y;
    const int j = blockIdx.x * 64 + threadIdx.x;
    if(j >= R)
        return;
    const size_t oj = (size_t)(qslot + t) * R + j;
    const int n = match_cnt[oj];
    const double* in = contrib + oj * cap; // this lane's own row: consecutive addresses, L1-friendly
    double s = 0;
    int k = 0;
    for(; k + 16 <= n; k += 16) // 16 loads in flight, then the 16 dependent adds in order
    {
        double v[16];
#pragma unroll
        for(int u = 0; u < 16; ++u)
            v[u] = in[k + u];
#pragma unroll
        for(int u = 0; u < 16; ++u)
            s += v[u];
    }
    for(; k < n; ++k)
        s += in[k];
    // Database::queryL1 only reports entries sharing a word with the query
    scores[oj] = n > 0 ? -s / 2.0 : -1.0;
}

// best entry per query: highest score, ties to the lower entry id (= the higher j)
__global__ __launch_bounds__(64) void k_bow_best(const double* __restrict__ scores, int qslot, long long base_id,
                                                 int per_frame_id, int R, int32_t* __restrict__ best_entry,
                                                 double* __restrict__ best_score)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const long long id_t = base_id + (per_frame_id ? t : 0);
    double bs = -1.0;
    long long bi = -1;
    for(int j = lane; j < R; j += 64)
    {
        const double s = scores[(size_t)(qslot + t) * R + j];
        const long long e = id_t - 1 - j;
        if(s >= 0 && e >= 0 && (bi < 0 || s > bs || (s == bs && e < bi)))
        {
            bs = s;
            bi = e;
        }
    }
    for(int o = 32; o > 0; o >>= 1)
    {
        const double s2 = __shfl_xor(bs, o);
        const long long i2 = __shfl_xor(bi, o);
        if(i2 >= 0 && (bi < 0 || s2 > bs || (s2 == bs && i2 < bi)))
        {
            bs = s2;
            bi = i2;
        }
    }
    if(lane == 0)
    {
        best_entry[qslot + t] = (int32_t)bi;
        best_score[qslot + t] = bi >= 0 ? bs : 0.0;
    }
}

// L1Scoring::score of two explicit vectors (host-pointer entry point)
__global__ void k_bow_score_pair(const uint32_t* __restrict__ w1, const double* __restrict__ v1, int n1,
                                 const uint32_t* __restrict__ w2, const double* __restrict__ v2, int n2,
                                 double* __restrict__ out)
{
    int a = 0, b = 0;
    double s = 0;
    while(a < n1 && b < n2)
    {
        const uint32_t x = w1[a], y = w2[b];
        if(x == y)
        {
            const double vi = v1[a], wi = v2[b];
            s += fabs(vi - wi) - fabs(vi) - fabs(wi);
            ++a, ++b;
        }
        else if(x < y)
            ++a;
        else
            ++b;
    }
    *out = -s / 2.0;
}

// local frame t against foreign vector [r][t]; one thread per (t, r)
__global__ __launch_bounds__(64) void k_bow_cross_score(const uint32_t* __restrict__ bwords,
                                                        const double* __restrict__ bvalues,
                                                        const int32_t* __restrict__ bn, int cap, int n_frames,
                                                        const uint32_t* __restrict__ fwords,
                                                        const double* __restrict__ fvalues, const int32_t* __restrict__ fn,
                                                        int n_sets, int fcap, int max_batch, double* __restrict__ scores)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if(idx >= n_frames * n_sets)
        return;
    const int t = idx / n_sets, r = idx - t * n_sets;
    const uint32_t* w1 = bwords + (size_t)t * cap;
    const double* v1 = bvalues + (size_t)t * cap;
    const int n1 = bn[t];
    const size_t fo = (size_t)r * max_batch + t;
    const uint32_t* w2 = fwords + fo * fcap;
    const double* v2 = fvalues + fo * fcap;
    const int n2 = min(fn[fo], fcap);
    int a = 0, b = 0;
    double s = 0;
    while(a < n1 && b < n2)
    {
        const uint32_t x = w1[a], y = w2[b];
        if(x == y)
        {
            const double vi = v1[a], wi = v2[b];
            s += fabs(vi - wi) - fabs(vi) - fabs(wi);
            ++a, ++b;
        }
        else if(x < y)
            ++a;
        else
            ++b;
    }
    scores[(size_t)t * n_sets + r] = -s / 2.0;
}

// ---- inverted file (DBoW3 Database::add / queryL1; sources not in the reference tree, published algorithm) -------------
// add: every (word, value) of the new entries is appended to the log and pushed on its word's list.  Several entries
// added by one launch may share words: atomicExch keeps every list intact whatever the order.
// (one 1024-thread workgroup: exclusive prefix sum of the entries' word counts, chunk by chunk; a single thread walking
// 1000 counts took 56 us of the cfg3 step)
__global__ __launch_bounds__(1024) void k_ix_alloc(const int32_t* __restrict__ bn, int n_entries, uint32_t* __restrict__ size,
                                                   uint32_t* __restrict__ starts /*[n_entries]*/, uint32_t cap_postings,
                                                   uint32_t* __restrict__ flags)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if(tid == 0)
        run = *size;
    __syncthreads();
    for(int base = 0; base < n_entries; base += 1024)
    {
        const int t = base + tid;
        const uint32_t v = t < n_entries ? (uint32_t)bn[t] : 0u;
        uint32_t inc = v; // inclusive scan inside the wave
#pragma unroll
        for(int o = 1; o < 64; o <<= 1)
        {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
            if(lane >= o)
                inc += u;
        }
        if(lane == 63)
            wave_tot[wave] = inc;
        __syncthreads();
        uint32_t before = run;
        for(int w = 0; w < wave; ++w)
            before += wave_tot[w];
        if(t < n_entries)
            starts[t] = before + inc - v;
        __syncthreads();
        if(tid == 1023)
            run = before + inc;
        __syncthreads();
    }
    if(tid == 0)
    {
        if(run > cap_postings)
            atomicOr(flags, kFlagDbFull); // nothing is appended by k_ix_append in that case
        else
            *size = run;
    }
}

__global__ __launch_bounds__(256) void k_ix_append(const uint32_t* __restrict__ bwords, const double* __restrict__ bvalues,
                                                   const int32_t* __restrict__ bn, int cap, long long base_id,
                                                   const uint32_t* __restrict__ starts, const uint32_t* __restrict__ size,
                                                   uint32_t cap_postings, uint32_t* __restrict__ head,
                                                   uint32_t* __restrict__ ix_entry, uint32_t* __restrict__ ix_prev,
                                                   double* __restrict__ ix_value)
{
    const int t = blockIdx.x;
    const int n = bn[t];
    const uint32_t start = starts[t];
    if((size_t)start + n > cap_postings || start + (uint32_t)n > *size)
        return; // the log is full (flagged by k_ix_alloc)
    for(int i = threadIdx.x; i < n; i += 256)
    {
        const uint32_t idx = start + i + 1; // 0 = null
        const uint32_t w = bwords[(size_t)t * cap + i];
        ix_entry[idx] = (uint32_t)(base_id + t);
        ix_value[idx] = bvalues[(size_t)t * cap + i];
        ix_prev[idx] = atomicExch(&head[w], idx);
    }
}

// query, pass 1 and 2: one thread per query word (rank i in ascending word order) walks that word's list.
// FILL = false: count the common words of every entry; FILL = true: store (rank, term) into the entry's segment.
template <bool FILL>
__global__ __launch_bounds__(256) void k_ix_walk(const uint32_t* __restrict__ qwords, const double* __restrict__ qvalues,
                                                 const int32_t* __restrict__ qn, const uint32_t* __restrict__ head,
                                                 const uint32_t* __restrict__ ix_entry, const uint32_t* __restrict__ ix_prev,
                                                 const double* __restrict__ ix_value, const uint8_t* __restrict__ removed,
                                                 long long id_limit, uint32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ ofs, uint32_t* __restrict__ fill,
                                                 uint32_t* __restrict__ keys, double* __restrict__ vals, uint32_t cap_terms)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= *qn)
        return;
    const double vi = qvalues[i];
    uint32_t idx = head[qwords[i]];
    while(idx != 0)
    {
        const uint32_t e = ix_entry[idx];
        if((long long)e < id_limit && !removed[e])
        {
            if(!FILL)
                atomicAdd(&cnt[e], 1u);
            else
            {
                const uint32_t p = ofs[e] + atomicAdd(&fill[e], 1u);
                if(p < cap_terms)
                {
                    const double wi = ix_value[idx];
                    keys[p] = (uint32_t)i;
                    vals[p] = fabs(vi - wi) - fabs(vi) - fabs(wi); // L1Scoring::score term
                }
            }
        }
        idx = ix_prev[idx];
    }
}

// exclusive scan of the per-entry counts (one workgroup; N is the number of entries ever added)
__global__ __launch_bounds__(1024) void k_ix_scan(const uint32_t* __restrict__ cnt, long long n, uint32_t* __restrict__ ofs,
                                                  uint32_t* __restrict__ total)
{
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    const long long per = (n + 1023) / 1024, b = tid * per, e = min(n, b + per);
    uint32_t s = 0;
    for(long long k = b; k < e; ++k)
        s += cnt[k];
    part[tid] = s;
    __syncthreads();
    for(int o = 1; o < 1024; o <<= 1)
    {
        const uint32_t t = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for(long long k = b; k < e; ++k)
    {
        ofs[k] = run;
        run += cnt[k];
    }
    if(tid == 1023)
    {
        ofs[n] = part[1023];
        *total = part[1023];
    }
}

// pass 3: one wave per entry.  The terms of an entry were stored in arrival order; the reference adds them in
// ascending word order (std::map iteration in Database::queryL1), i.e. ascending rank: pick the smallest rank
// above the last one, add its term, repeat.  Ranks are unique inside an entry.
__global__ __launch_bounds__(256) void k_ix_sum(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ofs,
                                                const uint32_t* __restrict__ keys, const double* __restrict__ vals,
                                                long long n, double* __restrict__ scores)
{
    const int lane = threadIdx.x & 63;
    for(long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += (long long)gridDim.x * 4)
    {
        const uint32_t c = cnt[e];
        if(c == 0)
        {
            if(lane == 0)
                scores[e] = -1.0; // no common word: Database::queryL1 does not report the entry
            continue;
        }
        const uint32_t* k = keys + ofs[e];
        const double* v = vals + ofs[e];
        double s = 0.0;
        long long last = -1;
        for(uint32_t it = 0; it < c; ++it)
        {
            unsigned long long best = ~0ull; // (rank << 32) | position
            for(uint32_t j = lane; j < c; j += 64)
            {
                const uint32_t r = k[j];
                if((long long)r > last)
                    best = min(best, ((unsigned long long)r << 32) | j);
            }
#pragma unroll
            for(int o = 32; o > 0; o >>= 1)
            {
                const unsigned long long t = (unsigned long long)__shfl_xor((long long)best, o);
                best = min(best, t);
            }
            s += v[(uint32_t)best];
            last = (long long)(best >> 32);
        }
        if(lane == 0)
            scores[e] = -s / 2.0;
    }
}

// ---- cross-stream exchange format ---------------------------------------------------------
// One "set" = the BoW vectors of one stream's batch as they travel in the all-gather: for every frame k_max x
// {u32 word, f32 value} (ascending words, zero padded), then the per-frame word counts:
//   uint2 vec[n_frames][k_max];  int32 count[n_frames];            set stride = n_frames * (2 k_max + 1) dwords
// k_bow_pack writes the local batch in that form; k_bow_cross_packed scores frame t of set `self` against frame t
// of every set (L1Scoring::score on the f32-rounded values, summed in ascending word order like the reference's
// map iteration), entirely from the gathered buffer, so it may run on any stream after the collective.
__global__ __launch_bounds__(256) void k_bow_pack(const uint32_t* __restrict__ bwords, const double* __restrict__ bvalues,
                                                  const int32_t* __restrict__ bn, int cap, int n_frames, int k_max,
                                                  uint32_t* __restrict__ out, uint32_t* __restrict__ flags)
{
    const int t = blockIdx.x;
    const int n = bn[t];
    uint2* dst = reinterpret_cast<uint2*>(out) + (size_t)t * k_max;
    for(int i = threadIdx.x; i < k_max; i += 256)
    {
        uint2 e = make_uint2(0u, 0u);
        if(i < n)
            e = make_uint2(bwords[(size_t)t * cap + i], __float_as_uint((float)bvalues[(size_t)t * cap + i]));
        dst[i] = e;
    }
    if(threadIdx.x == 0)
    {
        out[(size_t)n_frames * k_max * 2 + t] = (uint32_t)min(n, k_max);
        if(n > k_max)
            atomicOr(flags, kFlagBowPackOverflow);
    }
}

__global__ __launch_bounds__(256) void k_bow_cross_packed(const uint32_t* __restrict__ sets, int n_sets, int self,
                                                          int n_frames, int k_max, double* __restrict__ scores)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm_cross[]; // [k_max] words, [k_max] f32 values
    uint32_t* lw = sm_cross;
    float* lv = reinterpret_cast<float*>(sm_cross + k_max);
    const int t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t stride = (size_t)n_frames * (2 * (size_t)k_max + 1);
    const uint32_t* mine = sets + stride * self;
    const int n1 = min((int)mine[(size_t)n_frames * k_max * 2 + t], k_max);
    for(int i = tid; i < n1; i += 256)
    {
        const uint2 e = reinterpret_cast<const uint2*>(mine)[(size_t)t * k_max + i];
        lw[i] = e.x;
        lv[i] = __uint_as_float(e.y);
    }
    __syncthreads();
    for(int r = wave; r < n_sets; r += 4)
    {
        const uint32_t* other = sets + stride * r;
        const int n2 = min((int)other[(size_t)n_frames * k_max * 2 + t], k_max);
        const uint2* ov = reinterpret_cast<const uint2*>(other) + (size_t)t * k_max;
        double s = 0.0;
        for(int base = 0; base < n2; base += 64)
        {
            const int i = base + lane;
            double c = 0.0;
            bool found = false;
            if(i < n2)
            {
                const uint2 e = ov[i];
                int lo = 0, hi = n1; // first local word >= e.x
                while(lo < hi)
                {
                    const int mid = (lo + hi) >> 1;
                    if(lw[mid] < e.x)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                if(lo < n1 && lw[lo] == e.x)
                {
                    const double vi = (double)lv[lo], wi = (double)__uint_as_float(e.y);
                    c = fabs(vi - wi) - fabs(vi) - fabs(wi);
                    found = true;
                }
            }
            unsigned long long m = __ballot(found);
            while(m) // ascending word order = ascending lane order within the chunk
            {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)__double_as_longlong(c), l);
                const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(__double_as_longlong(c) >> 32), l);
                s += __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo32));
            }
        }
        if(lane == 0)
            scores[(size_t)t * n_sets + r] = -s / 2.0;
    }
}

// ---- host side ------------------------------------------------------------------------------------------
#define BHIPCHK(c, call)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if(e_ != hipSuccess)                                                                                           \
        {                                                                                                              \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                              \
            return MSLAM_HIP_E_RUNTIME;                                                                                \
        }                                                                                                              \
    } while(0)

static int bfail(mslam_hip_ctx* c, int code, const char* msg)
{
    c->err = msg;
    return code;
}

template <typename T>
static hipError_t bmalloc(T*& p, size_t n)
{
    return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T));
}

static int pow2_at_least(int n)
{
    int p = 64;
    while(p < n)
        p <<= 1;
    return p;
}

struct Reader
{
    const uint8_t* p;
    size_t size, pos = 0;
    bool ok = true;
    template <typename T>
    T get()
    {
        T v{};
        if(pos + sizeof(T) > size)
        {
            ok = false;
            return v;
        }
        std::memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
};

static int bow_load_impl(mslam_hip_ctx* c, const void* blob, size_t size)
{
    // Vocabulary::fromStream (dbow3.patch:2544-2651)
    Reader r{static_cast<const uint8_t*>(blob), size};
    if(r.get<uint64_t>() != 88877711233ull)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: bad magic (not a DBoW3 binary vocabulary)");
    const uint8_t compressed = r.get<uint8_t>();
    const uint32_t n_nodes = r.get<uint32_t>();
    if(!r.ok || n_nodes == 0)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: empty vocabulary");
    std::vector<uint8_t> body; // the decompressed stream when the file was saved with compressed = true
    if(compressed)
    {
        // dbow3.patch:2594-2611: u32 nChunks, then one QuickLZ packet per 10 000 bytes of the stream below
        const uint32_t n_chunks = r.get<uint32_t>();
        if(!r.ok || qlz_decode_stream(r.p + r.pos, size - r.pos, n_chunks, body) != MSLAM_HIP_OK)
            return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: cannot decode the QuickLZ packets of a compressed vocabulary "
                                                "(levels 1 and 3 of QuickLZ 1.5 are handled)");
        r = Reader{body.data(), body.size()};
        size = body.size();
    }
    const int k = r.get<int32_t>(), L = r.get<int32_t>(), scoring = r.get<int32_t>(), weighting = r.get<int32_t>();
    if(!r.ok || k < 1 || weighting < 0 || weighting > 3 || scoring < 0 || scoring > 5)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: bad header");
    if(scoring != MSLAM_BOW_L1_NORM)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: only L1_NORM scoring vocabularies are supported");
    std::vector<uint32_t> parent(n_nodes, 0), order(n_nodes, 0), nchild(n_nodes, 0), word(n_nodes, 0);
    std::vector<double> weight(n_nodes, 0.0);
    std::vector<uint8_t> desc((size_t)n_nodes * 32, 0);
    for(uint32_t i = 1; i < n_nodes; ++i)
    {
        const uint32_t nid = r.get<uint32_t>(), par = r.get<uint32_t>();
        const double w = r.get<double>();
        const int32_t cols = r.get<int32_t>(), rows = r.get<int32_t>(), type = r.get<int32_t>(); // DescManip::fromStream
        if(!r.ok || nid >= n_nodes || par >= n_nodes || cols != 32 || rows != 1 || type != 0 || r.pos + 32 > size)
            return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: bad node record (only 32-byte CV_8U descriptors are supported)");
        parent[nid] = par;
        weight[nid] = w;
        std::memcpy(&desc[(size_t)nid * 32], r.p + r.pos, 32);
        r.pos += 32;
        order[i] = nid;
        nchild[par]++;
    }
    const uint32_t n_words = r.get<uint32_t>();
    if(!r.ok)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: truncated stream");
    for(uint32_t i = 0; i < n_words; ++i)
    {
        const uint32_t wid = r.get<uint32_t>(), nid = r.get<uint32_t>();
        if(!r.ok || wid >= n_words || nid >= n_nodes)
            return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: bad word record");
        word[nid] = wid;
    }
    // children in stream order (m_nodes[parent].children.push_back, :2626)
    std::vector<uint32_t> coff(n_nodes + 1, 0), fill(n_nodes, 0), child(n_nodes, 0);
    for(uint32_t i = 0; i < n_nodes; ++i)
        coff[i + 1] = coff[i] + nchild[i];
    for(uint32_t i = 1; i < n_nodes; ++i)
    {
        const uint32_t nid = order[i], par = parent[nid];
        child[coff[par] + fill[par]++] = nid;
    }
    // slots: BFS so that siblings are contiguous
    std::vector<uint32_t> slot_node(n_nodes, 0), node_slot(n_nodes, 0);
    uint32_t next = 1, head = 0;
    int max_children = 0;
    if(nchild[0] == 0)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: root has no children");
    while(head < next)
    {
        const uint32_t nid = slot_node[head++];
        max_children = std::max(max_children, (int)nchild[nid]);
        for(uint32_t j = 0; j < nchild[nid]; ++j)
        {
            const uint32_t ch = child[coff[nid] + j];
            if(next >= n_nodes)
                return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: tree is not connected");
            node_slot[ch] = next;
            slot_node[next++] = ch;
        }
    }
    if(next != n_nodes)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: unreachable nodes");
    if(max_children > 65535)
        return bfail(c, MSLAM_HIP_E_FORMAT, "bow_load: branching factor too large");
    std::vector<uint8_t> sdesc((size_t)n_nodes * 32);
    std::vector<uint32_t> sfirst(n_nodes), snchild(n_nodes), sword(n_nodes);
    std::vector<double> sweight(n_nodes);
    for(uint32_t s = 0; s < n_nodes; ++s)
    {
        const uint32_t nid = slot_node[s];
        std::memcpy(&sdesc[(size_t)s * 32], &desc[(size_t)nid * 32], 32);
        snchild[s] = nchild[nid];
        sfirst[s] = nchild[nid] ? node_slot[child[coff[nid]]] : 0;
        sword[s] = word[nid];
        sweight[s] = weight[nid];
    }

    {
        // k_bow_score keeps the query vector and its hash table in LDS: cap * 8 + (slots + cap) * 4 bytes with
        // slots = the power of two >= 2 * cap; check it against the device limit here, not at the first query
        int slots = 1024, lds_max = 0;
        while(slots < 2 * c->p.max_keypoints)
            slots <<= 1;
        const size_t lds = (size_t)c->p.max_keypoints * 8 + ((size_t)slots + c->p.max_keypoints) * 4;
        if(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->p.device) != hipSuccess)
            return bfail(c, MSLAM_HIP_E_RUNTIME, "bow_load: cannot query the LDS size");
        if(lds > (size_t)lds_max || c->p.max_keypoints > 8192)
        {
            c->err = "bow_load: max_keypoints = " + std::to_string(c->p.max_keypoints) + " needs " + std::to_string(lds) +
                     " bytes of LDS for BoW scoring (device limit " + std::to_string(lds_max) +
                     "); create the context with max_keypoints <= 8192";
            return MSLAM_HIP_E_INVALID;
        }
    }
    BowState* b = new BowState();
    b->k = k, b->L = L, b->scoring = scoring, b->weighting = weighting;
    b->n_nodes = n_nodes, b->n_words = n_words, b->max_children = max_children;
    b->cap = c->p.max_keypoints;
    b->B = c->p.max_batch;
    b->RP = b->R + b->B;
    const size_t cap = (size_t)b->cap, B = (size_t)b->B, RP = (size_t)b->RP;
    auto fail_free = [&](int rc) {
        bow_destroy(b);
        return rc;
    };
#define BALLOC(ptr, n)                                                                                                 \
    if(bmalloc(ptr, n) != hipSuccess)                                                                                  \
    {                                                                                                                  \
        c->err = "bow_load: device allocation failed";                                                                 \
        return fail_free(MSLAM_HIP_E_RUNTIME);                                                                         \
    }
    BALLOC(b->d_desc, (size_t)n_nodes * 2);
    BALLOC(b->d_first, n_nodes);
    BALLOC(b->d_nchild, n_nodes);
    BALLOC(b->d_word, n_nodes);
    BALLOC(b->d_weight, n_nodes);
    BALLOC(b->d_fword, (B + 1) * cap);
    BALLOC(b->d_fweight, (B + 1) * cap);
    BALLOC(b->d_bwords, (B + 1) * cap);
    BALLOC(b->d_bvalues, (B + 1) * cap);
    BALLOC(b->d_bn, B + 1);
    BALLOC(b->d_rwords, RP * cap);
    BALLOC(b->d_rvalues, RP * cap);
    BALLOC(b->d_rn, RP);
    BALLOC(b->d_scores, (B + 1) * (size_t)b->R);
    BALLOC(b->d_contrib, (B + 1) * cap * (size_t)b->R);
    BALLOC(b->d_match_cnt, (B + 1) * (size_t)b->R);
    BALLOC(b->d_best_entry, B + 1);
    BALLOC(b->d_best_score, B + 1);
    BALLOC(b->d_hdesc, cap * 32);
    BALLOC(b->d_leaf_desc, (size_t)n_words * 8);
    BALLOC(b->d_leaf_weight, n_words);
    BALLOC(b->d_fbest, (B + 1) * cap);
    BALLOC(b->ix_head, n_words);
    BALLOC(b->ix_size, 1);
    BALLOC(b->ix_total, 1);
#undef BALLOC
    if(hipMemset(b->ix_head, 0, (size_t)n_words * 4) != hipSuccess || hipMemset(b->ix_size, 0, 4) != hipSuccess)
    {
        c->err = "bow_load: upload failed";
        return fail_free(MSLAM_HIP_E_RUNTIME);
    }
    bool ok = hipMemcpy(b->d_desc, sdesc.data(), sdesc.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_first, sfirst.data(), n_nodes * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_nchild, snchild.data(), n_nodes * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_word, sword.data(), n_nodes * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(b->d_weight, sweight.data(), n_nodes * 8, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemset(b->d_rn, 0, RP * 4) == hipSuccess && hipMemset(b->d_bn, 0, (B + 1) * 4) == hipSuccess;
    {
        // leaves in word-id order for the flat search (a word id is 20 bits in its key)
        std::vector<uint8_t> ldesc((size_t)n_words * 32, 0);
        std::vector<double> lweight(n_words, 0.0);
        std::vector<uint8_t> seen(n_words, 0);
        bool leaves_ok = n_words <= (1u << 20);
        for(uint32_t nid = 0; nid < n_nodes && leaves_ok; ++nid)
            if(nid != 0 && nchild[nid] == 0)
            {
                const uint32_t w = word[nid];
                if(w >= n_words || seen[w])
                    leaves_ok = false;
                else
                {
                    seen[w] = 1;
                    std::memcpy(&ldesc[(size_t)w * 32], &desc[(size_t)nid * 32], 32);
                    lweight[w] = weight[nid];
                }
            }
        b->flat = leaves_ok ? 0 : -1; // -1: flat mode unavailable for this vocabulary (words are not 1:1 with leaves)
        ok = ok && hipMemcpy(b->d_leaf_desc, ldesc.data(), ldesc.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(b->d_leaf_weight, lweight.data(), lweight.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
    if(!ok)
    {
        c->err = "bow_load: upload failed";
        return fail_free(MSLAM_HIP_E_RUNTIME);
    }
    if(c->bow)
        bow_destroy(c->bow);
    c->bow = b;
    return MSLAM_HIP_OK;
}

// transform `n_frames` descriptor sets (device) into BoW vectors at batch slots slot0..; counts from device or fixed
static int bow_transform_dev(mslam_hip_ctx* c, const uint8_t* d_desc, long long stride, const int32_t* d_counts,
                             int n_fixed, int n_frames, int slot0)
{
    BowState* b = c->bow;
    hipStream_t s = c->stream;
    const int cap = b->cap;
    if(b->flat == 1)
    {
        StageScope t(c, "bow_flat");
        uint32_t* best = b->d_fbest + (size_t)slot0 * cap;
        BHIPCHK(c, hipMemsetAsync(best, 0xFF, (size_t)n_frames * cap * 4, s));
        // slices: enough workgroups to fill 256 CUs x 8 even for a single frame, at least 2048 words each
        const int qblocks = (cap + 63) / 64;
        int n_slices = std::max(1, std::min((int)((b->n_words + 2047) / 2048), std::max(1, 4096 / std::max(1, qblocks * n_frames))));
        const int slice = (int)((b->n_words + n_slices - 1) / n_slices);
        n_slices = (int)((b->n_words + slice - 1) / slice);
        hipLaunchKernelGGL(k_bow_flat, dim3(qblocks, n_slices, n_frames), dim3(64 * kFlatWaves), 0, s, d_desc, stride, d_counts,
                           n_fixed, cap, b->d_leaf_desc, (int)b->n_words, slice, best);
        hipLaunchKernelGGL(k_bow_flat_finish, dim3((cap + 255) / 256, n_frames), dim3(256), 0, s, best, d_counts, n_fixed, cap,
                           b->d_leaf_weight, b->d_fword + (size_t)slot0 * cap, b->d_fweight + (size_t)slot0 * cap);
    }
    else
    {
        StageScope t(c, "bow_descend");
        const int per_block = (256 / kBowGroup) * kBowDepth; // descriptors per workgroup
        dim3 grid((cap + per_block - 1) / per_block, n_frames);
        hipLaunchKernelGGL(k_bow_descend, grid, dim3(256), 0, s, d_desc, stride, d_counts, n_fixed, cap, b->d_desc,
                           b->d_first, b->d_nchild, b->d_word, b->d_weight, b->d_fword + (size_t)slot0 * cap,
                           b->d_fweight + (size_t)slot0 * cap);
    }
    {
        StageScope t(c, "bow_vector");
        const int npow2 = pow2_at_least(cap);
        if((size_t)npow2 * 8 > 48 * 1024)
            BHIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_vector),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, npow2 * 8));
        hipLaunchKernelGGL(k_bow_vector, dim3(n_frames), dim3(VT), (size_t)npow2 * 8, s, b->d_fword + (size_t)slot0 * cap,
                           b->d_fweight + (size_t)slot0 * cap, d_counts, n_fixed, cap, npow2, b->weighting, b->scoring,
                           b->d_bwords, b->d_bvalues, b->d_bn, slot0);
    }
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

static int bow_score_dev(mslam_hip_ctx* c, int qslot, int n_frames, long long base_id, int per_frame_id)
{
    BowState* b = c->bow;
    StageScope t3(c, "bow_score");
    int slots = 1024;
    while(slots < 2 * b->cap)
        slots <<= 1;
    const size_t lds = (size_t)b->cap * 8 + (size_t)(slots + b->cap) * 4;
    if(lds > 48 * 1024)
        BHIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_score),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bow_score, dim3(kScoreBlocksPerFrame, n_frames), dim3(64 * kScoreWaves), lds, c->stream, b->d_bwords, b->d_bvalues, b->d_bn, qslot,
                       b->cap, b->d_rwords, b->d_rvalues, b->d_rn, base_id, per_frame_id, b->R, b->RP, slots, b->d_contrib,
                       b->d_match_cnt);
    hipLaunchKernelGGL(k_bow_sum, dim3((b->R + 63) / 64, n_frames), dim3(64), 0, c->stream, b->d_contrib, b->d_match_cnt,
                       qslot, b->cap, b->R, b->d_scores);
    hipLaunchKernelGGL(k_bow_best, dim3(n_frames), dim3(64), 0, c->stream, b->d_scores, qslot, base_id, per_frame_id, b->R,
                       b->d_best_entry, b->d_best_score);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

// ---- inverted file: host side -----------------------------------------------------------------------------------
template <typename T>
static int ix_grow(mslam_hip_ctx* c, T*& ptr, size_t old_n, size_t new_n, bool zero)
{
    T* np = nullptr;
    BHIPCHK(c, bmalloc(np, new_n));
    if(zero)
        BHIPCHK(c, hipMemsetAsync(np, 0, new_n * sizeof(T), c->stream));
    if(ptr && old_n)
        BHIPCHK(c, hipMemcpyAsync(np, ptr, old_n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    if(ptr)
        (void)hipFree(ptr);
    ptr = np;
    return MSLAM_HIP_OK;
}

// make room for `max_entries` database entries (postings: max_entries x max_keypoints in the worst case)
static int ix_reserve(mslam_hip_ctx* c, long long max_entries)
{
    BowState* b = c->bow;
    if(max_entries <= b->ix_max_entries)
        return MSLAM_HIP_OK;
    if(max_entries > (1ll << 31) / std::max(b->cap, 1))
        return bfail(c, MSLAM_HIP_E_CAPACITY, "bow database: more postings than a 32-bit index can address");
    const size_t old_e = (size_t)b->ix_max_entries, new_e = (size_t)max_entries;
    const size_t old_p = b->ix_cap_postings ? b->ix_cap_postings + 1 : 0, new_p = new_e * (size_t)b->cap + 1;
    int rc = 0;
    rc = rc ? rc : ix_grow(c, b->ix_entry, old_p, new_p, false);
    rc = rc ? rc : ix_grow(c, b->ix_prev, old_p, new_p, false);
    rc = rc ? rc : ix_grow(c, b->ix_value, old_p, new_p, false);
    rc = rc ? rc : ix_grow(c, b->ix_removed, old_e, new_e, true);
    rc = rc ? rc : ix_grow(c, b->ix_cnt, 0, new_e, true);
    rc = rc ? rc : ix_grow(c, b->ix_ofs, 0, new_e + 1, true);
    rc = rc ? rc : ix_grow(c, b->ix_fill, 0, new_e, true);
    rc = rc ? rc : ix_grow(c, b->ix_scores, 0, new_e, false);
    rc = rc ? rc : ix_grow(c, b->ix_keys, 0, new_p, false);
    rc = rc ? rc : ix_grow(c, b->ix_vals, 0, new_p, false);
    if(rc)
        return rc;
    b->ix_max_entries = max_entries;
    b->ix_cap_postings = new_p - 1;
    b->ix_cap_terms = new_p;
    return MSLAM_HIP_OK;
}

// append the BoW vectors in batch slots [slot0, slot0 + n) as entries next_id .. next_id + n - 1
static int ix_add(mslam_hip_ctx* c, int slot0, int n)
{
    BowState* b = c->bow;
    if(b->next_id + n > b->ix_max_entries)
    {
        const int rc = ix_reserve(c, std::max<long long>(2 * b->ix_max_entries, std::max<long long>(1024, b->next_id + n)));
        if(rc)
            return rc;
    }
    hipLaunchKernelGGL(k_ix_alloc, dim3(1), dim3(1024), 0, c->stream, b->d_bn + slot0, n, b->ix_size, b->ix_ofs,
                       (uint32_t)b->ix_cap_postings, c->d_flags); // ix_ofs doubles as the per-entry start scratch (n <= B + 1 <= capacity)
    hipLaunchKernelGGL(k_ix_append, dim3(n), dim3(256), 0, c->stream, b->d_bwords + (size_t)slot0 * b->cap,
                       b->d_bvalues + (size_t)slot0 * b->cap, b->d_bn + slot0, b->cap, b->next_id, b->ix_ofs, b->ix_size,
                       (uint32_t)b->ix_cap_postings, b->ix_head, b->ix_entry, b->ix_prev, b->ix_value);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

// score the query vector in batch slot `qslot` against every entry with id < id_limit: ix_scores[e] (-1: no common word)
static int ix_query(mslam_hip_ctx* c, int qslot, long long id_limit)
{
    BowState* b = c->bow;
    hipStream_t s = c->stream;
    const long long N = id_limit;
    BHIPCHK(c, hipMemsetAsync(b->ix_cnt, 0, (size_t)N * 4, s));
    BHIPCHK(c, hipMemsetAsync(b->ix_fill, 0, (size_t)N * 4, s));
    const uint32_t* qw = b->d_bwords + (size_t)qslot * b->cap;
    const double* qv = b->d_bvalues + (size_t)qslot * b->cap;
    const int32_t* qn = b->d_bn + qslot;
    const dim3 grid((b->cap + 255) / 256);
    hipLaunchKernelGGL(k_ix_walk<false>, grid, dim3(256), 0, s, qw, qv, qn, b->ix_head, b->ix_entry, b->ix_prev, b->ix_value,
                       b->ix_removed, id_limit, b->ix_cnt, b->ix_ofs, b->ix_fill, b->ix_keys, b->ix_vals,
                       (uint32_t)b->ix_cap_terms);
    hipLaunchKernelGGL(k_ix_scan, dim3(1), dim3(1024), 0, s, b->ix_cnt, N, b->ix_ofs, b->ix_total);
    hipLaunchKernelGGL(k_ix_walk<true>, grid, dim3(256), 0, s, qw, qv, qn, b->ix_head, b->ix_entry, b->ix_prev, b->ix_value,
                       b->ix_removed, id_limit, b->ix_cnt, b->ix_ofs, b->ix_fill, b->ix_keys, b->ix_vals,
                       (uint32_t)b->ix_cap_terms);
    const unsigned blocks = (unsigned)std::min<long long>((N + 3) / 4, 4096);
    hipLaunchKernelGGL(k_ix_sum, dim3(blocks), dim3(256), 0, s, b->ix_cnt, b->ix_ofs, b->ix_keys, b->ix_vals, N, b->ix_scores);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

int bow_batch(mslam_hip_ctx* c, int add_to_db)
{
    BowState* b = c->bow;
    if(c->n_last < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_batch_dev: no detect batch");
    const size_t K = (size_t)c->p.max_keypoints;
    const int n = c->n_last;
    int rc = bow_transform_dev(c, c->d_desc + K * 32, (long long)K * 32, c->d_count + 1, 0, n, 0);
    if(rc)
        return rc;
    if(add_to_db)
    {
        hipLaunchKernelGGL(k_bow_commit, dim3(n), dim3(256), 0, c->stream, b->d_bwords, b->d_bvalues, b->d_bn, b->cap,
                           b->next_id, b->RP, b->d_rwords, b->d_rvalues, b->d_rn);
        rc = ix_add(c, 0, n); // the inverted file sees the batch's entries too (host-side queries are unbounded)
        if(rc)
            return rc;
    }
    rc = bow_score_dev(c, 0, n, b->next_id, add_to_db ? 1 : 0);
    if(rc)
        return rc;
    if(add_to_db)
        b->next_id += n;
    return MSLAM_HIP_OK;
}

} // namespace mslam

using namespace mslam;

static int need_voc(mslam_hip_ctx* c)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!c->bow)
    {
        c->err = "no vocabulary loaded (call mslam_hip_bow_load first)";
        return MSLAM_HIP_E_NO_VOCABULARY;
    }
    BHIPCHK(c, hipSetDevice(c->p.device));
    return MSLAM_HIP_OK;
}

// upload n host descriptors and transform them into batch slot B (the host-query slot)
static int host_transform(mslam_hip_ctx* c, const uint8_t* desc, int n)
{
    BowState* b = c->bow;
    if(n < 0 || n > b->cap || (n > 0 && !desc))
        return bfail(c, MSLAM_HIP_E_INVALID, "bow: descriptor count outside [0, max_keypoints]");
    if(n > 0)
        BHIPCHK(c, hipMemcpyAsync(b->d_hdesc, desc, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    return bow_transform_dev(c, b->d_hdesc, 0, nullptr, n, 1, b->B);
}

extern "C" {

int mslam_hip_bow_load(mslam_hip_ctx* c, const void* blob, size_t size)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!blob)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_load: null blob");
    BHIPCHK(c, hipSetDevice(c->p.device));
    if(hipStreamSynchronize(c->stream) != hipSuccess)
        return bfail(c, MSLAM_HIP_E_RUNTIME, "bow_load: stream sync failed");
    try
    {
        return bow_load_impl(c, blob, size);
    }
    catch(const std::bad_alloc&) // an untrusted header can ask for more host memory than there is: no exception crosses the C ABI
    {
        return bfail(c, MSLAM_HIP_E_RUNTIME, "bow_load: out of host memory while decoding the vocabulary");
    }
}

int mslam_hip_bow_info(mslam_hip_ctx* c, int* k, int* L, int* n_nodes, int* n_words, int* scoring, int* weighting)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    const BowState* b = c->bow;
    if(k) *k = b->k;
    if(L) *L = b->L;
    if(n_nodes) *n_nodes = (int)b->n_nodes;
    if(n_words) *n_words = (int)b->n_words;
    if(scoring) *scoring = b->scoring;
    if(weighting) *weighting = b->weighting;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_words(mslam_hip_ctx* c, const uint8_t* desc, int n, uint32_t* word, double* weight)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    if(n == 0)
        return MSLAM_HIP_OK;
    if(!word || !weight)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_words: null output");
    rc = host_transform(c, desc, n);
    if(rc)
        return rc;
    const size_t o = (size_t)b->B * b->cap;
    BHIPCHK(c, hipMemcpyAsync(word, b->d_fword + o, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    BHIPCHK(c, hipMemcpyAsync(weight, b->d_fweight + o, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_transform(mslam_hip_ctx* c, const uint8_t* desc, int n, uint32_t* words, double* values, int* n_words)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    if(!n_words)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_transform: null output");
    *n_words = 0;
    rc = host_transform(c, desc, n);
    if(rc)
        return rc;
    int32_t m = 0;
    BHIPCHK(c, hipMemcpyAsync(&m, b->d_bn + b->B, 4, hipMemcpyDeviceToHost, c->stream));
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    if(m > 0)
    {
        if(!words || !values)
            return bfail(c, MSLAM_HIP_E_INVALID, "bow_transform: null output");
        const size_t o = (size_t)b->B * b->cap;
        BHIPCHK(c, hipMemcpy(words, b->d_bwords + o, (size_t)m * 4, hipMemcpyDeviceToHost));
        BHIPCHK(c, hipMemcpy(values, b->d_bvalues + o, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    *n_words = m;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_score(mslam_hip_ctx* c, const uint32_t* w1, const double* v1, int n1, const uint32_t* w2,
                        const double* v2, int n2, double* score)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    if(!score || n1 < 0 || n2 < 0 || n1 > b->cap || n2 > b->cap || (n1 > 0 && (!w1 || !v1)) || (n2 > 0 && (!w2 || !v2)))
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_score: bad argument");
    // vector 1 -> the host-query batch slot, vector 2 -> the per-feature scratch of that slot
    const size_t o = (size_t)b->B * b->cap;
    uint32_t* tw = b->d_fword + o;
    double* tv = b->d_fweight + o;
    hipStream_t s = c->stream;
    if(n1 > 0)
    {
        BHIPCHK(c, hipMemcpyAsync(b->d_bwords + o, w1, (size_t)n1 * 4, hipMemcpyHostToDevice, s));
        BHIPCHK(c, hipMemcpyAsync(b->d_bvalues + o, v1, (size_t)n1 * 8, hipMemcpyHostToDevice, s));
    }
    if(n2 > 0)
    {
        BHIPCHK(c, hipMemcpyAsync(tw, w2, (size_t)n2 * 4, hipMemcpyHostToDevice, s));
        BHIPCHK(c, hipMemcpyAsync(tv, v2, (size_t)n2 * 8, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_bow_score_pair, dim3(1), dim3(1), 0, s, b->d_bwords + o, b->d_bvalues + o, n1, tw, tv, n2,
                       b->d_best_score + b->B);
    BHIPCHK(c, hipGetLastError());
    BHIPCHK(c, hipMemcpyAsync(score, b->d_best_score + b->B, 8, hipMemcpyDeviceToHost, s));
    BHIPCHK(c, hipStreamSynchronize(s));
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_db_add(mslam_hip_ctx* c, const uint8_t* desc, int n, int* entry_id)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    rc = host_transform(c, desc, n);
    if(rc)
        return rc;
    const size_t o = (size_t)b->B * b->cap;
    hipLaunchKernelGGL(k_bow_commit, dim3(1), dim3(256), 0, c->stream, b->d_bwords + o, b->d_bvalues + o, b->d_bn + b->B,
                       b->cap, b->next_id, b->RP, b->d_rwords, b->d_rvalues, b->d_rn);
    BHIPCHK(c, hipGetLastError());
    rc = ix_add(c, b->B, 1);
    if(rc)
        return rc;
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    if(entry_id)
        *entry_id = (int)b->next_id;
    b->next_id += 1;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_db_query(mslam_hip_ctx* c, const uint8_t* desc, int n, int max_results, int32_t* entry_ids,
                           double* scores, int* n_results)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    if(!n_results || max_results < 0 || (max_results > 0 && (!entry_ids || !scores)))
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_db_query: bad argument");
    *n_results = 0;
    rc = host_transform(c, desc, n);
    if(rc)
        return rc;
    // Database::queryL1 over the inverted file: every entry ever added (and not removed) that shares a word
    const long long N = b->next_id;
    if(N == 0)
        return MSLAM_HIP_OK;
    rc = ix_query(c, b->B, N);
    if(rc)
        return rc;
    std::vector<double> sc((size_t)N);
    BHIPCHK(c, hipMemcpyAsync(sc.data(), b->ix_scores, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<std::pair<double, long long>> res;
    for(long long id = 0; id < N; ++id)
        if(sc[(size_t)id] >= 0)
            res.emplace_back(sc[(size_t)id], id);
    std::sort(res.begin(), res.end(), [](const auto& x, const auto& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
    const int m = std::min<int>(max_results, (int)res.size());
    for(int i = 0; i < m; ++i)
    {
        entry_ids[i] = (int32_t)res[i].second;
        scores[i] = res[i].first;
    }
    *n_results = m;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_set_assignment(mslam_hip_ctx* c, int mode)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    if(mode != MSLAM_BOW_ASSIGN_TREE && mode != MSLAM_BOW_ASSIGN_FLAT)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_set_assignment: unknown mode");
    if(mode == MSLAM_BOW_ASSIGN_FLAT && c->bow->flat < 0)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_set_assignment: this vocabulary's words are not one per leaf (or more than 2^20)");
    c->bow->flat = mode;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_db_remove(mslam_hip_ctx* c, int entry_id)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BowState* b = c->bow;
    if(entry_id < 0 || entry_id >= b->next_id)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_db_remove: no such entry");
    BHIPCHK(c, hipMemsetAsync(b->ix_removed + entry_id, 1, 1, c->stream)); // its postings are skipped from now on
    if(entry_id >= b->next_id - b->R)
        // the batched path's window: an entry without words shares no word with any query
        BHIPCHK(c, hipMemsetAsync(b->d_rn + (entry_id % b->RP), 0, 4, c->stream));
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_db_clear(mslam_hip_ctx* c)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    BHIPCHK(c, hipStreamSynchronize(c->stream));
    BHIPCHK(c, hipMemset(c->bow->d_rn, 0, (size_t)c->bow->RP * 4));
    BHIPCHK(c, hipMemset(c->bow->ix_head, 0, (size_t)c->bow->n_words * 4));
    BHIPCHK(c, hipMemset(c->bow->ix_size, 0, 4));
    if(c->bow->ix_removed)
        BHIPCHK(c, hipMemset(c->bow->ix_removed, 0, (size_t)c->bow->ix_max_entries));
    c->bow->next_id = 0;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_db_reserve(mslam_hip_ctx* c, int max_entries)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    if(max_entries < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_db_reserve: bad argument");
    return ix_reserve(c, max_entries);
}

int mslam_hip_bow_db_size(mslam_hip_ctx* c, int* n_entries)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    if(n_entries)
        *n_entries = (int)c->bow->next_id;
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_batch_dev(mslam_hip_ctx* c, int add_to_db)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    return bow_batch(c, add_to_db);
}

int mslam_hip_bow_cross_score_dev(mslam_hip_ctx* c, const uint32_t* d_words, const double* d_values,
                                  const int32_t* d_n, int n_sets, int capacity, double* d_scores)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    const BowState* b = c->bow;
    if(!d_words || !d_values || !d_n || !d_scores || n_sets < 1 || capacity < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_cross_score_dev: bad argument");
    if(c->n_last < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_cross_score_dev: no BoW batch");
    const int total = c->n_last * n_sets;
    hipLaunchKernelGGL(k_bow_cross_score, dim3((total + 63) / 64), dim3(64), 0, c->stream, b->d_bwords, b->d_bvalues,
                       b->d_bn, b->cap, c->n_last, d_words, d_values, d_n, n_sets, capacity, b->B, d_scores);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_pack_dev(mslam_hip_ctx* c, int k_max, uint32_t* d_out)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    const BowState* b = c->bow;
    if(!d_out || k_max < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_pack_dev: bad argument");
    if(c->n_last < 1)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_pack_dev: no BoW batch");
    hipLaunchKernelGGL(k_bow_pack, dim3(c->n_last), dim3(256), 0, c->stream, b->d_bwords, b->d_bvalues, b->d_bn, b->cap,
                       c->n_last, k_max, d_out, c->d_flags);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

int mslam_hip_bow_cross_score_packed_dev(mslam_hip_ctx* c, const uint32_t* d_sets, int n_sets, int self_set, int n_frames,
                                         int k_max, double* d_scores, void* stream)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    BHIPCHK(c, hipSetDevice(c->p.device));
    if(!d_sets || !d_scores || n_sets < 1 || self_set < 0 || self_set >= n_sets || n_frames < 1 || k_max < 1 ||
       k_max > 16384)
        return bfail(c, MSLAM_HIP_E_INVALID, "bow_cross_score_packed_dev: bad argument");
    const size_t lds = (size_t)k_max * 8;
    if(lds > 48 * 1024)
        BHIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_cross_packed),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bow_cross_packed, dim3(n_frames), dim3(256), lds, stream ? (hipStream_t)stream : c->stream, d_sets,
                       n_sets, self_set, n_frames, k_max, d_scores);
    BHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

int mslam_hip_get_bow_view(mslam_hip_ctx* c, mslam_hip_bow_view* v)
{
    int rc = need_voc(c);
    if(rc)
        return rc;
    if(!v)
        return MSLAM_HIP_E_INVALID;
    const BowState* b = c->bow;
    v->capacity = b->cap;
    v->words = b->d_bwords;
    v->values = b->d_bvalues;
    v->n_words = b->d_bn;
    v->best_entry = b->d_best_entry;
    v->best_score = b->d_best_score;
    return MSLAM_HIP_OK;
}

} // extern "C"
