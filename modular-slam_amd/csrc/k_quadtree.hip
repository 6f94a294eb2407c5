// k_quadtree.hip — quadtree keypoint selection, one workgroup per (level, frame).
//
// Replaces distribute_keypoints_via_tree + initialize_nodes + orb_extractor_node::divide_node +
// assign_child_nodes + find_keypoints_with_max_response (reference distributed_cv_feature.cpp:981-1155,
// :306-355).  The reference walks a std::list sequentially; what it computes is order-sensitive
// (children are push_front'ed, the loop stops when a pass leaves the list size unchanged, and the
// surviving keypoints come out in list order).  This kernel reproduces that order in parallel:
//
//   list order after one pass = [ children of the LAST divided node (reversed) ... children of the
//   FIRST divided node (reversed) ] ++ [ undivided nodes in their old order ]
//
// so with D_i = number of non-empty children created by divided nodes before node i, U_i = number of
// undivided nodes before i and T = total children, child rank r of node i lands at T-1-(D_i+r) and an
// undivided node at T+U_i: two workgroup-wide exclusive scans per pass.  Keypoints never move; each
// carries the list position of its node.  The per-leaf winner ("first maximum", :1142-1149) is an
// atomicMax over (score, -original index).
#include "common.hpp"
#include <cstdlib>

namespace mslam
{

constexpr int QT = 512;           // threads per workgroup (k_quadtree's small class: kSmallQT)
// k_quadtree's classes: a (level, frame) pair runs in the smallest instance that holds its candidates — the passes are
// latency-bound, so what sets the throughput is how many workgroups share a CU (LDS: 37 bytes per candidate)
constexpr int kSmallKp = 512, kSmallQT = 256; // 19 KB, 256 threads: 8 workgroups per CU
constexpr int kMidKp = 1024, kMidQT = 512;    // 38 KB, 512 threads: 4 workgroups per CU (the large class: 74 KB, 2 per CU)
constexpr int kLdsKp = 2048;      // levels with at most this many candidates keep ALL working arrays in LDS
constexpr int kMaxCellsPerLevel = 2048;
constexpr int kBigNodes = 4096;   // k_quadtree_big: node arrays in LDS up to this many list nodes,
constexpr int kBigKp = 12288;     //                 keypoint -> node links in LDS up to this many candidates
constexpr int kBigMinPixels = 400000; // levels with more pixels than this get the k_quadtree_big launch
constexpr int kMaxInitNodes = 64;
constexpr int kMaxPasses = 40;

struct Scan
{
    uint32_t wsum[QT / 64];
};

// inclusive prefix sum over the 64 lanes of a wave with DPP adds (VALU only; __shfl_up would take six trips through
// the LDS crossbar)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xE, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xC, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); // row_bcast:31
    return (uint32_t)x;
}

// exclusive scan of one value per thread across the workgroup; returns the exclusive prefix, `total`
// receives the workgroup sum.  Two barriers; `s` may be reused right after return.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, Scan& s, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if(lane == 63)
        s.wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for(int w = 0; w < NT / 64; ++w)
    {
        const uint32_t x = s.wsum[w];
        if(w < wave)
            base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// the same for two values at once (one pair of barriers)
template <int NT>
__device__ __forceinline__ void block_excl_scan2(uint32_t a, uint32_t b, Scan& s, Scan& s2, uint32_t& exa, uint32_t& exb,
                                                 uint32_t& tota, uint32_t& totb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t ia = wave_incl_scan(a), ib = wave_incl_scan(b);
    if(lane == 63)
    {
        s.wsum[wave] = ia;
        s2.wsum[wave] = ib;
    }
    __syncthreads();
    uint32_t ba = 0, bb = 0, ta = 0, tb = 0;
#pragma unroll
    for(int w = 0; w < NT / 64; ++w)
    {
        const uint32_t x = s.wsum[w], y = s2.wsum[w];
        if(w < wave)
        {
            ba += x;
            bb += y;
        }
        ta += x;
        tb += y;
    }
    __syncthreads();
    tota = ta;
    totb = tb;
    exa = ba + ia - a;
    exb = bb + ib - b;
}

__device__ __forceinline__ uint32_t ld_atomic(const uint32_t* p)
{
    // values produced by atomics must not be served from a stale L1 line
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void unpack_node(uint2 b, int& bx, int& by, int& ex, int& ey)
{
    bx = (int)(b.x & 0xFFFF), by = (int)(b.x >> 16), ex = (int)(b.y & 0xFFFF), ey = (int)(b.y >> 16);
}
__device__ __forceinline__ uint2 pack_node(int bx, int by, int ex, int ey)
{
    return make_uint2((uint32_t)bx | ((uint32_t)by << 16), (uint32_t)ex | ((uint32_t)ey << 16));
}

// Storage policy of the two instantiations.  The LDS one packs indices into 16 bits and the four child
// counters of a node into two dwords so that two workgroups fit one CU (the passes are latency-bound,
// so resident workgroups per CU is what sets the throughput); the global one is sized for any level.
struct LdsStore
{
    using idx_t = uint16_t;
    using info_t = uint8_t;
    static constexpr bool kLds = true;
    static constexpr bool kCandCopy = true;  // candidates are gathered into LDS; the global copy is for debug_read
    static constexpr uint32_t kNodeCap = 0;  // 0: the node arrays hold as many nodes as there are candidates
    static __device__ __forceinline__ void cc_zero(uint32_t* cc, uint32_t pos) { cc[2 * pos] = 0, cc[2 * pos + 1] = 0; }
    static __device__ __forceinline__ void cc_add(uint32_t* cc, uint32_t pos, int c) { atomicAdd(&cc[2 * pos + (c >> 1)], 1u << (16 * (c & 1))); }
    static __device__ __forceinline__ uint32_t cc_get(const uint32_t* cc, uint32_t pos, int c) { return (cc[2 * pos + (c >> 1)] >> (16 * (c & 1))) & 0xFFFFu; }
};
// Large levels (1280x720 and up: 2 000 - 12 000 candidates): the candidates themselves stay in global memory (read
// only after the gather), the per-node arrays and the keypoint -> node links are in LDS (152 KB, one workgroup per CU).
struct BigStore : LdsStore
{
    static constexpr bool kCandCopy = false;
    static constexpr uint32_t kNodeCap = kBigNodes; // more nodes than this: give up, the caller runs GlobalStore
};
struct GlobalStore
{
    using idx_t = uint32_t;
    using info_t = uint32_t;
    static constexpr bool kLds = false;
    static constexpr bool kCandCopy = false;
    static constexpr uint32_t kNodeCap = 0;
    static __device__ __forceinline__ void cc_zero(uint32_t* cc, uint32_t pos) { cc[4 * pos] = 0, cc[4 * pos + 1] = 0, cc[4 * pos + 2] = 0, cc[4 * pos + 3] = 0; }
    static __device__ __forceinline__ void cc_add(uint32_t* cc, uint32_t pos, int c) { atomicAdd(&cc[4 * pos + c], 1u); }
    static __device__ __forceinline__ uint32_t cc_get(const uint32_t* cc, uint32_t pos, int c) { return ld_atomic(&cc[4 * pos + c]); }
};

// Everything after the candidate count is known.  Instantiated twice and force-inlined so that, in the
// LDS instance, every working array is a known LDS object (ds_* instructions instead of flat_*).
template <class S, int NT>
__device__ __forceinline__ bool quad_run(const Geometry& g, const QuadArgs& a, const LevelGeom& lv, size_t frame,
                                         size_t slot, int tid, uint32_t N, int n_cells, const uint32_t* cell_off,
                                         uint32_t* cand, uint32_t* g_cand, typename S::idx_t* kp_node, uint2* nodes,
                                         uint2* nodes2, typename S::idx_t* ncnt, typename S::idx_t* ncnt2,
                                         typename S::info_t* ninfo, typename S::idx_t* nbase, uint32_t* cc,
                                         uint32_t* best, uint32_t* sel, Scan& scan, Scan& scan2, uint32_t* init_cnt,
                                         uint32_t* init_pos, uint32_t& sh_n)
{
    using idx_t = typename S::idx_t;
    constexpr uint32_t kNoNode = (uint32_t)(idx_t)~(idx_t)0;

    const uint32_t* ckp = a.cell_kp + (frame * g.n_cells + lv.cell_base) * (size_t)kCellCap;
    for(uint32_t j = tid; j < N; j += NT)
    {
        int lo = 0, hi = n_cells - 1; // last cell with cell_off <= j
        while(lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if(cell_off[mid] <= j)
                lo = mid;
            else
                hi = mid - 1;
        }
        const uint32_t v = ckp[(size_t)lo * kCellCap + (j - cell_off[lo])];
        cand[j] = v;
        if(S::kCandCopy)
            g_cand[j] = v; // the global copy is kept for mslam_hip_debug_read
    }
    __syncthreads(); // cell_off may share storage with arrays written below

    // ---- 1. initial nodes (:1025-1105)
    const int n_init = lv.nxg * lv.nyg;
    if(tid < kMaxInitNodes)
        init_cnt[tid] = 0;
    __syncthreads();
    for(uint32_t k = tid; k < N; k += NT)
    {
        const uint32_t p = cand[k];
        const unsigned ix = (unsigned)((double)(float)kp_x(p) / lv.delta_x);
        const unsigned iy = (unsigned)((double)(float)kp_y(p) / lv.delta_y);
        const unsigned idx = ix + iy * (unsigned)lv.nxg;
        if(idx < (unsigned)n_init)
        {
            atomicAdd(&init_cnt[idx], 1u);
            kp_node[k] = (idx_t)idx;
        }
        else
            kp_node[k] = (idx_t)kNoNode;
    }
    __syncthreads();
    if(tid == 0)
    {
        uint32_t n0 = 0;
        for(int i = 0; i < n_init; ++i)
        {
            init_pos[i] = n0;
            if(init_cnt[i] != 0)
            {
                const int ix = i % lv.nxg, iy = i / lv.nxg;
                nodes[n0] = pack_node((int)(lv.delta_x * ix), (int)(lv.delta_y * iy), (int)(lv.delta_x * (ix + 1)),
                                      (int)(lv.delta_y * (iy + 1)));
                ncnt[n0] = (idx_t)init_cnt[i];
                ++n0;
            }
        }
        sh_n = n0;
    }
    __syncthreads();
    for(uint32_t k = tid; k < N; k += NT)
    {
        const uint32_t idx = kp_node[k];
        if(idx != kNoNode)
            kp_node[k] = (idx_t)init_pos[idx];
    }
    uint32_t n = sh_n;
    __syncthreads();

    // ---- 2. subdivision passes (:992-1020)
    const float sf = lv.scale;
    const float min_size_f = (float)a.min_size;
    bool converged = false;
    for(int pass = 0; pass < kMaxPasses && n > 0; ++pass)
    {
        // a. which nodes divide (:1002)
        for(uint32_t pos = tid; pos < n; pos += NT)
        {
            int bx, by, ex, ey;
            unpack_node(nodes[pos], bx, by, ex, ey);
            const unsigned area = (unsigned)((ex - bx) * (ey - by));
            const bool keep = (uint32_t)ncnt[pos] == 1u || __fmul_rn(__fmul_rn((float)area, sf), sf) <= min_size_f;
            ninfo[pos] = keep ? 0u : 1u;
            if(!keep)
                S::cc_zero(cc, pos);
        }
        __syncthreads();
        // b. count keypoints per child (:340-352)
        for(uint32_t k = tid; k < N; k += NT)
        {
            const uint32_t pos = kp_node[k];
            if(pos == kNoNode || !(ninfo[pos] & 1u))
                continue;
            int bx, by, ex, ey;
            unpack_node(nodes[pos], bx, by, ex, ey);
            const int cx = bx + ((ex - bx + 1) >> 1), cy = by + ((ey - by + 1) >> 1); // cvCeil(d/2.0)
            const uint32_t p = cand[k];
            S::cc_add(cc, pos, (cx <= kp_x(p) ? 1 : 0) + (cy <= kp_y(p) ? 2 : 0));
        }
        __syncthreads();
        // c. list positions after this pass
        uint32_t D = 0, U = 0;
        for(uint32_t base = 0; base < n; base += NT)
        {
            const uint32_t pos = base + tid;
            uint32_t nchild = 0, und = 0, mask = 0, div = 0;
            if(pos < n)
            {
                div = ninfo[pos] & 1u;
                if(div)
                {
                    mask = (S::cc_get(cc, pos, 0) ? 1u : 0u) | (S::cc_get(cc, pos, 1) ? 2u : 0u) |
                           (S::cc_get(cc, pos, 2) ? 4u : 0u) | (S::cc_get(cc, pos, 3) ? 8u : 0u);
                    nchild = (uint32_t)__popc(mask);
                }
                else
                    und = 1;
            }
            uint32_t totD, totU, exD, exU;
            block_excl_scan2<NT>(nchild, und, scan, scan2, exD, exU, totD, totU);
            if(pos < n)
            {
                ninfo[pos] = (typename S::info_t)(div | (mask << 1));
                nbase[pos] = (idx_t)(div ? D + exD : U + exU);
            }
            D += totD;
            U += totU;
        }
        const uint32_t T = D;
        __syncthreads();
        if(S::kNodeCap != 0 && T + U > S::kNodeCap)
            return false; // (workgroup-uniform) the next list does not fit the LDS node arrays
        // d. materialise the new list; e. re-point the keypoints.  Both only read what step c wrote
        //    (nbase = scan prefix, ninfo = divide flag + child mask) and T, so they share one phase.
        for(uint32_t pos = tid; pos < n; pos += NT)
        {
            const uint32_t info = ninfo[pos];
            if(info & 1u)
            {
                int bx, by, ex, ey;
                unpack_node(nodes[pos], bx, by, ex, ey);
                const int cx = bx + ((ex - bx + 1) >> 1), cy = by + ((ey - by + 1) >> 1);
                const uint32_t first = T - 1 - (uint32_t)nbase[pos];
                const uint32_t mask = info >> 1;
                uint32_t r = 0;
                if(mask & 1u) { nodes2[first - r] = pack_node(bx, by, cx, cy); ncnt2[first - r] = (idx_t)S::cc_get(cc, pos, 0); ++r; }
                if(mask & 2u) { nodes2[first - r] = pack_node(cx, by, ex, cy); ncnt2[first - r] = (idx_t)S::cc_get(cc, pos, 1); ++r; }
                if(mask & 4u) { nodes2[first - r] = pack_node(bx, cy, cx, ey); ncnt2[first - r] = (idx_t)S::cc_get(cc, pos, 2); ++r; }
                if(mask & 8u) { nodes2[first - r] = pack_node(cx, cy, ex, ey); ncnt2[first - r] = (idx_t)S::cc_get(cc, pos, 3); ++r; }
            }
            else
            {
                const uint32_t np = T + (uint32_t)nbase[pos];
                nodes2[np] = nodes[pos];
                ncnt2[np] = ncnt[pos];
            }
        }
        for(uint32_t k = tid; k < N; k += NT)
        {
            const uint32_t pos = kp_node[k];
            if(pos == kNoNode)
                continue;
            const uint32_t info = ninfo[pos];
            uint32_t np;
            if(info & 1u)
            {
                int bx, by, ex, ey;
                unpack_node(nodes[pos], bx, by, ex, ey);
                const int cx = bx + ((ex - bx + 1) >> 1), cy = by + ((ey - by + 1) >> 1);
                const uint32_t p = cand[k];
                const int c = (cx <= kp_x(p) ? 1 : 0) + (cy <= kp_y(p) ? 2 : 0);
                np = T - 1 - (uint32_t)nbase[pos] - (uint32_t)__popc((info >> 1) & ((1u << c) - 1u));
            }
            else
                np = T + (uint32_t)nbase[pos];
            kp_node[k] = (idx_t)np;
        }
        __syncthreads();
        {
            uint2* t = nodes; nodes = nodes2; nodes2 = t;
            idx_t* u = ncnt; ncnt = ncnt2; ncnt2 = u;
        }
        const uint32_t n2 = T + U;
        const bool same = n2 == n; // :1016-1019 — the pass's effects stay even when it is the last
        n = n2;
        if(same)
        {
            converged = true;
            break;
        }
    }
    if(!converged && n > 0 && tid == 0)
        atomicOr(a.flags, kFlagQuadNoConverge);

    // ---- 3. winner per node, emitted in list order (:1128-1155)
    for(uint32_t pos = tid; pos < n; pos += NT)
        best[pos] = 0;
    __syncthreads();
    for(uint32_t k = tid; k < N; k += NT)
    {
        const uint32_t pos = kp_node[k];
        if(pos == kNoNode)
            continue;
        atomicMax(&best[pos], ((uint32_t)kp_score(cand[k]) << 24) | (0xFFFFFFu - k));
    }
    __syncthreads();
    for(uint32_t pos = tid; pos < n; pos += NT)
    {
        const uint32_t k = 0xFFFFFFu - ((S::kLds ? best[pos] : ld_atomic(&best[pos])) & 0xFFFFFFu);
        sel[pos] = cand[k];
    }
    if(tid == 0)
        a.sel_cnt[slot] = n;
    return true;
}

// gather step shared by both kernels: candidate offsets per cell (-> cell_off[0 .. n_cells]), returns N
template <int NT>
__device__ __forceinline__ uint32_t quad_cell_offsets(const uint32_t* ccnt, int n_cells, uint32_t* cell_off, Scan& scan)
{
    const int tid = threadIdx.x;
    uint32_t running = 0;
    for(int base = 0; base < n_cells; base += NT)
    {
        const int i = base + tid;
        const uint32_t v = i < n_cells ? ccnt[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<NT>(v, scan, tot);
        if(i < n_cells)
            cell_off[i] = running + ex;
        running += tot;
    }
    if(tid == 0)
        cell_off[n_cells] = running;
    __syncthreads();
    return running;
}

// KP / NT: LDS capacity (candidates) and threads of the instance, CLS its class (0 small, 1 mid, 2 large).  The three
// instances are launched on the same grid, one after the other; each takes the (level, frame) pairs of its class and leaves
// the others alone (a few microseconds for the candidate count).  On a 640x480 pyramid (about 990 ... 240 candidates on
// levels 0 ... 7) four levels run in the small class and four in the mid one.
__device__ __forceinline__ int quad_class(uint32_t N, int n_cells, unsigned classes)
{
    if((classes & 1u) && N <= (uint32_t)kSmallKp && n_cells <= 2 * kSmallKp - 1)
        return 0;
    if((classes & 2u) && N <= (uint32_t)kMidKp && n_cells <= 2 * kMidKp - 1)
        return 1;
    return 2;
}
// (waves per SIMD: the small and mid instances must fit 8 — 8 x 4 resp. 4 x 8 waves per CU —, i.e. 64 registers)
template <int KP, int NT, int CLS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(CLS == 2 ? 4 : 8))) void k_quadtree(Geometry g, QuadArgs a, unsigned big_levels)
{
    __shared__ Scan scan, scan2;
    __shared__ uint32_t l_cand[KP];
    __shared__ uint2 l_nodes_a[KP], l_nodes_b[KP]; // l_nodes_b doubles as the cell offset table of step 0
    __shared__ uint32_t l_cc[KP * 2];               // packed child counters; reused for the winners
    __shared__ uint16_t l_kp_node[KP], l_ncnt_a[KP], l_ncnt_b[KP], l_nbase[KP];
    __shared__ uint8_t l_ninfo[KP];
    __shared__ uint32_t init_cnt[kMaxInitNodes];
    __shared__ uint32_t init_pos[kMaxInitNodes];
    __shared__ uint32_t sh_n;
    constexpr int kCellsFit = 2 * KP - 1; // cells whose offsets (n_cells + 1 dwords) fit l_nodes_b
    static_assert(CLS != 2 || kCellsFit >= kMaxCellsPerLevel, "cell offsets must fit");
    const unsigned classes = big_levels >> 30; // which smaller instances were launched (bits 30, 31)
    uint32_t* cell_off = reinterpret_cast<uint32_t*>(l_nodes_b);

    // Workgroups go to the 8 XCDs round-robin by linear id = frame * n_levels + blockIdx.x.  With the usual 8
    // levels a plain level = blockIdx.x would give XCD 0 every level-0 workgroup (the heaviest) and XCD 7 every
    // level-7 one; rotating the level by the frame index gives every XCD the same mix.
    const int level = (int)((blockIdx.x + blockIdx.y) % (unsigned)g.n_levels);
    const size_t frame = blockIdx.y + g.frame0;
    const LevelGeom& lv = g.lv[level];
    const int tid = threadIdx.x;
    const size_t slot = frame * g.n_levels + level;
    const size_t cap = (size_t)a.cand_cap;
    uint32_t* cand = a.cand + slot * cap;
    uint32_t* sel = a.sel + slot * cap;

    // ---- 0. gather this level's candidates in the reference's order: cells row-major, then the
    //         row-major order inside each cell (:878-951)
    const int n_cells = lv.n_cells;
    if(n_cells > kCellsFit)
        return; // (a larger class takes the level)
    const uint32_t* ccnt = a.cell_cnt + frame * g.n_cells + lv.cell_base;
    const uint32_t N = quad_cell_offsets<NT>(ccnt, n_cells, cell_off, scan);
    if(quad_class(N, n_cells, classes) != CLS)
        return; // another instance's pair
    if(N > (uint32_t)a.cand_cap)
    {
        if(tid == 0)
        {
            atomicOr(a.flags, kFlagCandOverflow);
            a.cand_cnt[slot] = 0;
            a.sel_cnt[slot] = 0;
        }
        return;
    }
    if(tid == 0)
        a.cand_cnt[slot] = N;
    if(N == 0)
    {
        if(tid == 0)
            a.sel_cnt[slot] = 0;
        return;
    }
    if(N <= (uint32_t)KP)
        quad_run<LdsStore, NT>(g, a, lv, frame, slot, tid, N, n_cells, cell_off, l_cand, cand, l_kp_node, l_nodes_a, l_nodes_b,
                               l_ncnt_a, l_ncnt_b, l_ninfo, l_nbase, l_cc, l_cc, sel, scan, scan2, init_cnt, init_pos, sh_n);
    else if((big_levels >> level) & 1u)
        return; // k_quadtree_big, launched right behind this kernel, takes this (level, frame)
    else
        quad_run<GlobalStore, NT>(g, a, lv, frame, slot, tid, N, n_cells, cell_off, cand, cand, a.kp_node + slot * cap,
                                  a.nodes_a + slot * cap, a.nodes_b + slot * cap, a.ncnt_a + slot * cap,
                                  a.ncnt_b + slot * cap, a.ninfo + slot * cap, a.best + slot * cap,
                                  a.child_cnt + slot * cap * 4, a.best + slot * cap, sel, scan, scan2, init_cnt, init_pos,
                                  sh_n);
}

// Second kernel for the levels of large images: the (level, frame) pairs k_quadtree left alone because they have more
// than kLdsKp candidates.  Same algorithm (quad_run), node arrays in 152 KB of dynamic LDS; levels that outgrow even
// that (more than kBigKp candidates, or a list of more than kBigNodes nodes) run with every array in global memory.
__global__ __launch_bounds__(QT) void k_quadtree_big(Geometry g, QuadArgs a, unsigned big_levels)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t big_lds[];
    __shared__ Scan scan, scan2;
    __shared__ uint32_t init_cnt[kMaxInitNodes];
    __shared__ uint32_t init_pos[kMaxInitNodes];
    __shared__ uint32_t sh_n;
    uint2* l_nodes_a = reinterpret_cast<uint2*>(big_lds);
    uint2* l_nodes_b = l_nodes_a + kBigNodes; // doubles as the cell offset table of step 0
    uint32_t* l_cc = reinterpret_cast<uint32_t*>(l_nodes_b + kBigNodes);
    uint16_t* l_kp_node = reinterpret_cast<uint16_t*>(l_cc + 2 * kBigNodes);
    uint16_t* l_ncnt_a = l_kp_node + kBigKp;
    uint16_t* l_ncnt_b = l_ncnt_a + kBigNodes;
    uint16_t* l_nbase = l_ncnt_b + kBigNodes;
    uint8_t* l_ninfo = reinterpret_cast<uint8_t*>(l_nbase + kBigNodes);
    static_assert(kBigNodes * sizeof(uint2) >= (kMaxCellsPerLevel + 1) * sizeof(uint32_t), "cell offsets must fit");
    uint32_t* cell_off = reinterpret_cast<uint32_t*>(l_nodes_b);

    // blockIdx.x counts the set bits of big_levels, rotated by the frame (as k_quadtree: with 2 or 4 big levels a plain
    // count would tie each level to a fixed subset of the XCDs)
    int level = 0;
    {
        unsigned m = big_levels;
        const unsigned which = (blockIdx.x + blockIdx.y) % gridDim.x;
        for(unsigned i = 0; i < which; ++i)
            m &= m - 1;
        level = __ffs((int)m) - 1;
    }
    const size_t frame = blockIdx.y + g.frame0;
    const LevelGeom& lv = g.lv[level];
    const int tid = threadIdx.x;
    const size_t slot = frame * g.n_levels + level;
    const size_t cap = (size_t)a.cand_cap;
    uint32_t* cand = a.cand + slot * cap;
    uint32_t* sel = a.sel + slot * cap;
    const int n_cells = lv.n_cells;
    const uint32_t* ccnt = a.cell_cnt + frame * g.n_cells + lv.cell_base;
    const uint32_t N = quad_cell_offsets<QT>(ccnt, n_cells, cell_off, scan);
    if(N <= (uint32_t)kLdsKp || N > (uint32_t)a.cand_cap)
        return; // done (or flagged) by k_quadtree
    bool done = false;
    if(N <= (uint32_t)kBigKp)
        done = quad_run<BigStore, QT>(g, a, lv, frame, slot, tid, N, n_cells, cell_off, cand, cand, l_kp_node, l_nodes_a, l_nodes_b,
                                  l_ncnt_a, l_ncnt_b, l_ninfo, l_nbase, l_cc, l_cc, sel, scan, scan2, init_cnt, init_pos, sh_n);
    if(!done)
    {
        __syncthreads();
        // the cell offsets were overwritten (they share storage with the node arrays): rebuild them
        quad_cell_offsets<QT>(ccnt, n_cells, cell_off, scan);
        quad_run<GlobalStore, QT>(g, a, lv, frame, slot, tid, N, n_cells, cell_off, cand, cand, a.kp_node + slot * cap,
                              a.nodes_a + slot * cap, a.nodes_b + slot * cap, a.ncnt_a + slot * cap,
                              a.ncnt_b + slot * cap, a.ninfo + slot * cap, a.best + slot * cap,
                              a.child_cnt + slot * cap * 4, a.best + slot * cap, sel, scan, scan2, init_cnt, init_pos,
                              sh_n);
    }
}

void launch_quadtree(const Geometry& g, const QuadArgs& a, int frame0, int n_frames, hipStream_t s)
{
    dim3 grid(g.n_levels, n_frames);
    Geometry gg = g;
    gg.frame0 = frame0;
    unsigned big_levels = 0;
    for(int l = 0; l < g.n_levels; ++l)
        if(g.lv[l].w * g.lv[l].h > kBigMinPixels)
            big_levels |= 1u << l;
    // MSLAM_HIP_QUAD_CLASSES: bit 0 = small instance, bit 1 = mid instance (default both; 0 = the large instance alone)
    static const unsigned classes_env = [] { const char* e = getenv("MSLAM_HIP_QUAD_CLASSES"); return e ? (unsigned)atoi(e) & 3u : 3u; }();
    // a handful of frames (the synchronous single-frame call): one launch — there is no occupancy to gain, and every launch
    // is a few microseconds of the call's latency
    const unsigned classes = n_frames < 8 ? 0u : classes_env;
    const unsigned bl = big_levels | (classes << 30);
    hipLaunchKernelGGL((k_quadtree<kLdsKp, QT, 2>), grid, dim3(QT), 0, s, gg, a, bl);
    if(classes & 2u)
        hipLaunchKernelGGL((k_quadtree<kMidKp, kMidQT, 1>), grid, dim3(kMidQT), 0, s, gg, a, bl);
    if(classes & 1u)
        hipLaunchKernelGGL((k_quadtree<kSmallKp, kSmallQT, 0>), grid, dim3(kSmallQT), 0, s, gg, a, bl);
    if(big_levels != 0)
    {
        constexpr size_t lds = (size_t)kBigNodes * (8 + 8 + 8 + 2 + 2 + 2 + 1) + (size_t)kBigKp * 2;
        static const bool attr = [] {
            return hipFuncSetAttribute((const void*)k_quadtree_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        }();
        (void)attr;
        hipLaunchKernelGGL(k_quadtree_big, dim3((unsigned)__builtin_popcount(big_levels), n_frames), dim3(QT), lds, s, gg, a,
                           big_levels);
    }
}

} // namespace mslam
