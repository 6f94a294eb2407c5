// k_fast.hip — per-cell FAST-9/16 corner score + 3x3 non-max suppression + ordered emission.
//
// Replaces the cell loop of compute_fast_keypoints (reference distributed_cv_feature.cpp:858-952):
// for every 64-px cell (70x70 sub-image with the 6-px overlap) cv::FAST(sub, thr=ini, nms=true) and,
// only if that cell yields nothing, cv::FAST(sub, thr=min, nms=true) (:918-926).
//
// One 256-thread workgroup per (cell, frame).  The kernel is integer-VALU bound, so it is organised
// to spend instructions only where corners can be:
//   A. every tested pixel takes the 4-point compass test (a 9-arc always contains two adjacent
//      compass points of one polarity), four pixels per lane with packed 16-bit compares on dword LDS
//      reads; survivors are compacted into an LDS list with ballots;
//   B. the list is processed densely (all 64 lanes busy), TWO pixels per lane: 16 circle reads each and the arc score
//        S = max(max_arc min(d), max_arc min(-d)) - 1,   d = centre - circle pixel,
//      over the 16 arcs of 9 pixels, with packed 16-bit 3-input min/max (arc_score.hpp).  For a pixel that passes the 9-contiguous
//      test at threshold t this is OpenCV's cornerScore<16>(.., t), and it passes iff S >= t, so the
//      test itself never has to be evaluated separately;
//   C. listed pixels with S >= t are checked against their 8 neighbours in the LDS score map (strict >)
//      and set one bit in a 64x64 bitmap; row popcounts of the bitmap give the reference's row-major
//      output order with no dependence on atomic ordering.
// If the cell is empty at the first threshold the same three steps run again with the fallback one.
#include <type_traits>
#include "common.hpp"
#include "arc_score.hpp"

namespace mslam
{

constexpr int kTileP = 80; // tile row pitch in bytes: five 16-byte LDS-DMA chunks from column x0 - 1 on,
constexpr int kTileX = 1;  // i.e. tile byte = column + 1: the tested columns 3+4i .. 6+4i are bytes 4+4i .. 7+4i, dword i + 1
constexpr int kScP = 68;   // score-map row pitch

// A cell descriptor through the scalar unit (the compiler takes a vector load + v_readfirstlane per field for it).
__device__ __forceinline__ CellDesc load_cell(const CellDesc* __restrict__ cells, int cell)
{
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t r;
    const uint32_t off = (uint32_t)cell * (uint32_t)sizeof(CellDesc);
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(cells), "s"(off));
    CellDesc c;
    c.level = (int16_t)(r.x & 0xFFFFu), c.cw = (int16_t)(r.x >> 16);
    c.ch = (int16_t)(r.y & 0xFFFFu), c.pad = 0;
    c.x0 = (int16_t)(r.z & 0xFFFFu), c.y0 = (int16_t)(r.z >> 16);
    c.ox = (int16_t)(r.w & 0xFFFFu), c.oy = (int16_t)(r.w >> 16);
    return c;
}
static_assert(sizeof(CellDesc) == 16, "load_cell reads a descriptor as four dwords");

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// waves_per_eu(8, 8): the kernel is VALU-bound and the latency of its LDS phases is hidden by the OTHER workgroups of the
// CU; left alone the allocator takes 75 VGPRs (6 waves per SIMD), held to 64 it spills two dwords and runs 8 waves per
// SIMD: 0.52 -> 0.45 ms per 500 frames.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast_cells(const uint8_t* __restrict__ pyr, Geometry g,
                                                    const CellDesc* __restrict__ cells, uint32_t* __restrict__ cell_cnt,
                                                    uint32_t* __restrict__ cell_kp, int ini_thr, int min_thr, int n_frames)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[70 * kTileP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[(66 * kScP + 15) / 16 * 16];
    __shared__ __attribute__((aligned(4))) uint16_t cand[64 * 64];
    __shared__ uint32_t bitmap[64 * 2];
    constexpr uint32_t kPassCap = 512;
    __shared__ uint16_t plist[kPassCap]; // candidates whose score reaches the threshold (what phase C looks at)
    __shared__ uint32_t n_cand, n_pass;

    // XCD-aware mapping: workgroups are handed to the 8 XCDs round-robin by linear id, and neighbouring cells share
    // 128-byte lines (a 70-byte cell row straddles two of them, its neighbours use the rest).  With cells of one
    // frame spread over all XCDs every L2 fetched those lines again — 2.7x the level's bytes at the memory side
    // (profiles/r02_a_pmc_fetch_write_per_launch.json); all cells of a frame now get ids with the same (id & 7).
    // (grid: x = cell * 8 + XCD, y = group of 8 frames — the XCD of a workgroup follows its linear id, and gridDim.x is a
    // multiple of 8; no division, and the cell's descriptor comes through the scalar unit)
    // A handful of frames (the synchronous single-frame calls, n_frames < 8): grid x = cell, y = frame — the cells of the one
    // frame spread over all XCDs; that launch is latency-bound, not memory-bound.
    const bool spread = n_frames < 8;
    const int xcd = blockIdx.x & 7, cell_id = spread ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    const int f_local = spread ? (int)blockIdx.y : (int)blockIdx.y * 8 + xcd;
    if(f_local >= n_frames)
        return;
    const size_t frame = (size_t)f_local + g.frame0;
    const CellDesc c = load_cell(cells, cell_id);
    const LevelGeom& lv = g.lv[c.level];
    const int cw = c.cw, ch = c.ch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // stage the sub-image by LDS-DMA: a row is five 16-byte chunks that go from global memory straight into the tile — no
    // registers, no vector instructions for the copy.  The chunks start at column x0 - 1 = 18 + 64 j, which puts the
    // tested columns on dword boundaries of the tile (phase A reads centre / upper / lower pixels without a byte shift) and
    // is NOT an aligned address: LDS-DMA takes any byte address (tools/experiments/dma_align_probe.hip: every shift 0..19
    // exact, dwordx4 and dword).  (Round 3 started at x0 - 3, a 16-byte boundary, and paid three v_alignbyte per step.)
    // Chunk t = (row t / 5, chunk t % 5) lands at tile + 16 t; the last chunk of a row ends 9 bytes beyond the cell (the
    // next cell's pixels, or the slab's pad behind the last row).
    {
        const uint8_t* src = pyr + frame * g.slab + lv.offset + (size_t)c.y0 * lv.pitch + (c.x0 - kTileX);
        const int n_chunks = ch * 5;
#pragma unroll
        for(int p = 0; p < 2; ++p)
        {
            const int t = p * 256 + tid;
            if(t < n_chunks)
            {
                const int r = (t * 13108) >> 16, j = t - r * 5; // t / 5 for t < 16384
                const uint32_t off = (uint32_t)(r * lv.pitch + 16 * j); // (32-bit lane offset on a scalar base: no 64-bit vector adds)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + off),
                    (__attribute__((address_space(3))) void*)&tile[(p * 256 + wave * 64) * 16], 16, 0, 0);
            }
        }
    }
    // meanwhile: zero the score map (untested pixels must read 0, like FAST_t's zeroed row buffers) and the bitmap
    // (its first 4 KB serve phase A as record segments and are cleared there)
    if(tid < (66 * kScP + 15) / 16 - 256)
        reinterpret_cast<uint4*>(sc)[256 + tid] = make_uint4(0u, 0u, 0u, 0u);
    if(tid < 128)
        bitmap[tid] = 0;
    if(tid == 0)
        n_cand = 0, n_pass = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the wave's DMA chunks have landed
    __syncthreads();

    // LDS byte offset of the list's fill counter (for the hand-written reservation in phase A)
    const uint32_t n_cand_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(&n_cand);
    const uint32_t n_pass_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(&n_pass);
    uint32_t total = 0;
    for(int pass = 0; pass < 2; ++pass)
    {
        const int thr = pass == 0 ? ini_thr : min_thr;

        // ---- A. compass test, four pixels per lane as ONE dword of bytes: a wave covers 4 rows x 64 columns per step.
        //      v_lerp_u8 is a per-byte (a + b + r) >> 1 with no carries between the bytes; with b = ~centre it is the halved,
        //      biased difference q = (p - c + 255) >> 1 of four pixels at once, and bit 7 of a second lerp against a
        //      constant is a threshold test on q.  Halving costs one bit: q >= (t + 256) >> 1 is implied by p > c + t and
        //      q <= (254 - t) >> 1 by p < c - t, so the test admits a few pixels the exact compass test would reject —
        //      the list is a work queue for the exact arc score of phase B, any superset of the corners is correct.
        //      Surviving GROUPS (dwords with a flag) are compacted by ballot into the wave's own segment of a record list
        //      (flags in bits 7 / 15 / 23 / 31, (y << 8 | x) of the group's first pixel in the bits between); the sparse
        //      records are then expanded to pixels by the same wave.  The segments live in the score map's storage (which
        //      phase B wants zeroed anyway: every wave clears its own segment once it has expanded it).
        {
            typedef __attribute__((address_space(3))) uint32_t* lds32_t;
            typedef __attribute__((address_space(3))) uint16_t* lds16_t;
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) u32x4_t* lds128_t;
            // (everything lane-dependent of this phase is derived from an opaque copy of the thread id inside the pass loop:
            // hoisted out of it, these values would hold registers through phases B and C, where the kernel sits at its 64)
            uint32_t t_op = threadIdx.x;
            asm volatile("" : "+v"(t_op));
            const uint32_t ln = t_op & 63u;
            const int wv = __builtin_amdgcn_readfirstlane((int)(t_op >> 6));
            // lane = (row sub, column group i4) of a step.  A full cell has 16 column groups (64 tested columns) and a wave
            // step covers 4 rows; the narrow cells that end a level's cell rows (and the short ones that end its cell columns)
            // use fewer groups per row and more rows per step — lg = log2(groups per row) — and only as many steps as they
            // have rows: 8 % of the steps of a 640x480 pyramid are saved.  (The order of the list is irrelevant: it is a
            // work queue, the output order comes from the bitmap.)
            const int ngrp = (cw - 6 + 3) >> 2;                                             // column groups that hold a tested column
            const int lg = ngrp > 8 ? 4 : ngrp > 4 ? 3 : ngrp > 2 ? 2 : ngrp > 1 ? 1 : 0; // (wave-uniform)
            const int i4 = (int)(ln & ((1u << lg) - 1u)), sub = (int)(ln >> lg);
            const int rps = 64 >> lg;                                                     // rows per wave step
            const int n_steps = (ch - 6 + rps - 1) >> (6 - lg);                            // row steps of the cell, dealt to the waves round-robin (<= 16)
            const int tq = min(max(thr, 0), 254);
            const uint32_t kb = (uint32_t)(256 - ((tq + 256) >> 1)) * 0x01010101u; // bit 7 of lerp(q, kb): q >= (t + 256) >> 1
            const uint32_t kd = (uint32_t)(255 - ((254 - tq) >> 1)) * 0x01010101u; // bit 7 of lerp(q, kd): q >  (254 - t) >> 1
            // which of this lane's four columns 3+4i .. 6+4i are tested (x < cw - 3), as bit 7 of the four bytes
            const int xl = 3 + 4 * i4;
            const int n_col = min(max(cw - 3 - xl, 0), 4);
            const uint32_t colmask = (uint32_t)(0x80808080ull >> (8 * (4 - n_col)));
            const uint32_t tile_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)tile;
            const uint32_t cand_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)cand;
            const uint32_t seg_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)sc + (uint32_t)wv * 1024u; // <= 4 steps x 64 groups per wave
            int y = 3 + wv * rps + sub;
            uint32_t row = tile_lds + (uint32_t)((y - 3) * kTileP + 4 * i4); // byte offset of dword i4 of tile row y - 3 (the topmost row a step reads)
            uint32_t yx = (uint32_t)((y << 8) | xl);
            uint32_t n_grp = 0; // (wave-uniform)
            // CHECKED = false: a full cell (every lane has four tested columns, every step four full rows — the interior cells,
            // most of them): no per-row / per-lane validity test, no row counter
            auto steps = [&](auto checked) {
                constexpr bool CHECKED = decltype(checked)::value;
#pragma unroll 1
                for(int st = wv; st < n_steps; st += 4, y += CHECKED ? 4 * rps : 0, row += (uint32_t)(4 * rps * kTileP), yx += (uint32_t)(4 * rps) << 8)
                {
                    uint32_t keep = 0;
                    if(!CHECKED || (colmask != 0 && y < ch - 3))
                    {
                        // the four tested pixels are dword i + 1 of the row
                        const lds32_t r = lds_ptr<lds32_t>(row);
                        const uint32_t d0 = r[60], d1 = r[61], d2 = r[62];
                        const uint32_t nC = ~d1, U = r[1], D = r[121];
                        const uint32_t Lf = __builtin_amdgcn_alignbyte(d1, d0, 1); // columns x-3 of the four pixels
                        const uint32_t Rt = __builtin_amdgcn_alignbyte(d2, d1, 3); // columns x+3
                        const uint32_t q0 = __builtin_amdgcn_lerp(D, nC, 0u), q8 = __builtin_amdgcn_lerp(U, nC, 0u);
                        const uint32_t q4 = __builtin_amdgcn_lerp(Rt, nC, 0u), q12 = __builtin_amdgcn_lerp(Lf, nC, 0u);
                        // brighter: (p0 or p8) and (p4 or p12);  not darker: (n0 and n8) or (n4 and n12) with n = "not darker"
                        const uint32_t br = (__builtin_amdgcn_lerp(q0, kb, 0u) | __builtin_amdgcn_lerp(q8, kb, 0u)) &
                                            (__builtin_amdgcn_lerp(q4, kb, 0u) | __builtin_amdgcn_lerp(q12, kb, 0u));
                        const uint32_t nd = (__builtin_amdgcn_lerp(q0, kd, 0u) & __builtin_amdgcn_lerp(q8, kd, 0u)) |
                                            (__builtin_amdgcn_lerp(q4, kd, 0u) & __builtin_amdgcn_lerp(q12, kd, 0u));
                        keep = (br | ~nd) & (CHECKED ? colmask : 0x80808080u);
                    }
                    const bool any = keep != 0;
                    const unsigned long long vote = __ballot(any);
                    if(any)
                        lds_ptr<lds32_t>(seg_lds)[n_grp + __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u))] =
                            keep | yx;
                    n_grp += (uint32_t)__popcll(vote);
                }
            };
            if(cw == 70 && ch == 70)
                steps(std::false_type{});
            else
                steps(std::true_type{});
            // expansion of the wave's records: pixel k of a record goes to the list slot given by the ballot of flag k
            for(uint32_t i0 = 0; i0 < n_grp; i0 += 64)
            {
                const uint32_t rec = i0 + ln < n_grp ? lds_ptr<lds32_t>(seg_lds)[i0 + ln] : 0u;
                const bool f0 = (rec & 0x80u) != 0, f1 = (rec & 0x8000u) != 0, f2 = (rec & 0x800000u) != 0, f3 = (int)rec < 0;
                const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
                const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2),
                               c3 = (uint32_t)__popcll(m3);
                // one lane reserves the wave's range of the list.  Written out by hand: around `if(lane == 0) atomicAdd(..)`
                // the compiler's atomic optimiser builds a wave reduction for a value that is wave-uniform already.
                uint32_t base;
                {
                    uint32_t got;
                    unsigned long long save;
                    const uint32_t add = c0 + c1 + c2 + c3; // (a vector register copy of the wave-uniform count)
                    asm volatile("s_mov_b64 %1, exec\n\t"
                                 "s_mov_b64 exec, 1\n\t"
                                 "ds_add_rtn_u32 %0, %2, %3\n\t"
                                 "s_waitcnt lgkmcnt(0)\n\t"
                                 "s_mov_b64 exec, %1"
                                 : "=&v"(got), "=&s"(save)
                                 : "v"(n_cand_lds), "v"(add)
                                 : "memory");
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                }
                const uint32_t v = rec & 0x7F7Fu;
                // (byte addresses, the wave-uniform part of each in a scalar: one v_lshl_add per store address)
                uint32_t b2 = cand_lds + 2u * base;
                if(f0)
                    *lds_ptr<lds16_t>(b2 + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))) = (uint16_t)v;
                b2 += 2u * c0;
                if(f1)
                    *lds_ptr<lds16_t>(b2 + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))) = (uint16_t)(v + 1);
                b2 += 2u * c1;
                if(f2)
                    *lds_ptr<lds16_t>(b2 + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u))) = (uint16_t)(v + 2);
                b2 += 2u * c2;
                if(f3)
                    *lds_ptr<lds16_t>(b2 + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, 0u))) = (uint16_t)(v + 3);
            }
            // the segment's storage becomes score map again (bytes beyond the four segments were cleared at the start)
            {
                uint32_t z = 0;
                asm volatile("" : "+v"(z)); // (a zero made here: hoisted, a uint4 of zeros holds four registers through the kernel)
                lds_ptr<lds128_t>(seg_lds)[ln] = u32x4_t{z, z, z, z};
            }
        }
        __syncthreads();
        const uint32_t n = n_cand;

        // ---- B. arc score of the survivors, two per lane (packed 16-bit 3-input min / max)
        for(uint32_t j = tid; 2 * j < n; j += 256)
        {
            const uint32_t two = *reinterpret_cast<const uint32_t*>(&cand[2 * j]); // entries 2j, 2j+1
            const uint32_t ca = two & 0xFFFFu, cb = 2 * j + 1 < n ? two >> 16 : ca;  // an odd tail scores its last pixel twice
            const int ay = (int)(ca >> 8), ax = (int)(ca & 0xFF), by = (int)(cb >> 8), bx = (int)(cb & 0xFF);
            const uint8_t* pa = &tile[ay * kTileP + ax + kTileX];
            const uint8_t* pb = &tile[by * kTileP + bx + kTileX];
            // circle pixel k of both candidates side by side in one register (16-bit halves): the byte loads write the halves
            // directly (ds_read_u8_d16 / _d16_hi), no vector instruction packs them
            typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
            uint32_t e[16];
#define MSLAM_E(k, off)                                                                    \
    {                                                                                      \
        u16x2_t v2;                                                                        \
        v2.x = (unsigned short)pa[off];                                                    \
        v2.y = (unsigned short)pb[off];                                                    \
        e[k] = __builtin_bit_cast(uint32_t, v2);                                           \
    }
            MSLAM_E(0, 3 * kTileP);
            MSLAM_E(1, 3 * kTileP + 1);
            MSLAM_E(2, 2 * kTileP + 2);
            MSLAM_E(3, kTileP + 3);
            MSLAM_E(4, 3);
            MSLAM_E(5, -kTileP + 3);
            MSLAM_E(6, -2 * kTileP + 2);
            MSLAM_E(7, -3 * kTileP + 1);
            MSLAM_E(8, -3 * kTileP);
            MSLAM_E(9, -3 * kTileP - 1);
            MSLAM_E(10, -2 * kTileP - 2);
            MSLAM_E(11, -kTileP - 3);
            MSLAM_E(12, -3);
            MSLAM_E(13, kTileP - 3);
            MSLAM_E(14, 2 * kTileP - 2);
            MSLAM_E(15, 3 * kTileP - 1);
#undef MSLAM_E
            int sa, sb;
            arc_score2_raw(e, (int)pa[0], (int)pb[0], sa, sb);
            const bool pass_a = sa >= thr && sa > 0, pass_b = sb >= thr && sb > 0 && 2 * j + 1 < n;
            if(pass_a)
                sc[(ay - 2) * kScP + (ax - 2)] = (uint8_t)sa;
            if(pass_b)
                sc[(by - 2) * kScP + (bx - 2)] = (uint8_t)sb;
            // the passing pixels (about a third of the list) go to a second, short list: phase C then runs one or two dense
            // wave-iterations instead of one per 64 candidates
            const unsigned long long ma = __ballot(pass_a), mb = __ballot(pass_b);
            const uint32_t na = (uint32_t)__popcll(ma), nb = (uint32_t)__popcll(mb);
            if(na + nb != 0)
            {
                uint32_t got;
                unsigned long long save;
                const uint32_t add = na + nb;
                asm volatile("s_mov_b64 %1, exec\n\t"
                             "s_mov_b64 exec, 1\n\t"
                             "ds_add_rtn_u32 %0, %2, %3\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(got), "=&s"(save)
                             : "v"(n_pass_lds), "v"(add)
                             : "memory");
                const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                const uint32_t ia = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
                const uint32_t ib = base + na + __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
                if(pass_a && ia < kPassCap) // (a list that overflows is not used: n_pass > kPassCap sends phase C over all candidates)
                    plist[ia] = (uint16_t)ca;
                if(pass_b && ib < kPassCap)
                    plist[ib] = (uint16_t)cb;
            }
        }
        __syncthreads();

        // ---- C. 3x3 strict non-max suppression over the passing pixels -> bitmap
        const uint32_t np = n_pass;
        const bool short_list = np <= kPassCap;
        for(uint32_t i = tid; i < (short_list ? np : n); i += 256)
        {
            const uint32_t cxy = short_list ? plist[i] : cand[i];
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
            // inclusive scan of the row counts with DPP adds (VALU only: __shfl_up is a trip through the LDS crossbar per
            // step, six of them in a row while the other three waves wait at the barrier)
            uint32_t inc = cnt;
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true); // row_shr:1
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true); // row_shr:2
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xE, true); // row_shr:4
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xC, true); // row_shr:8
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xA, 0xF, true); // row_bcast:15
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xC, 0xF, true); // row_bcast:31
            total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
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
            {
                n_cand = total != 0 ? 0xFFFFFFFFu : 0u; // tells the other waves whether to run the fallback pass
                n_pass = 0;
            }
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

// One-off self-check at context creation (api.hip): arc_score2_raw and the FAST kernels of both detector modes order the f16
// DENORMAL bit patterns 0x0000 .. 0x00FF (raw pixels) with v_pk_minimum3_f16 / v_pk_maximum3_f16, which holds only while the
// kernels run with f16 denormals preserved (.amdhsa_float_denorm_mode_16_64 3, the compiler's default).  A build flag or a
// function attribute that flushes them would collapse every score without a compile-time signal: this kernel — compiled in
// this translation unit, with the same flags — evaluates the two instructions on such patterns, and creation fails loudly if
// the result is not the integer min / max.  out[0] = 1 when every check holds.
__global__ void k_denorm_selfcheck(uint32_t* out)
{
    const uint32_t t = threadIdx.x; // 64 lanes: patterns (t, 255 - t) in the two halves
    const uint32_t a = t | ((255u - t) << 16), b = ((t * 7u + 3u) & 255u) | (((t * 13u + 1u) & 255u) << 16), c = 1u | (254u << 16);
    const uint32_t mx = pk_max3(a, b, c), mn = pk_min3(a, b, c);
    auto lo = [](uint32_t x) { return x & 0xFFFFu; };
    auto hi = [](uint32_t x) { return x >> 16; };
    const bool ok = lo(mx) == max(max(lo(a), lo(b)), lo(c)) && hi(mx) == max(max(hi(a), hi(b)), hi(c)) &&
                    lo(mn) == min(min(lo(a), lo(b)), lo(c)) && hi(mn) == min(min(hi(a), hi(b)), hi(c));
    const unsigned long long all = __ballot(ok);
    if(t == 0)
        out[0] = all == ~0ull ? 1u : 0u;
}

void launch_denorm_selfcheck(uint32_t* d_out, hipStream_t s) { hipLaunchKernelGGL(k_denorm_selfcheck, dim3(1), dim3(64), 0, s, d_out); }

void launch_fast(const uint8_t* d_pyr, const Geometry& g, const CellDesc* d_cells, uint32_t* d_cell_cnt,
                 uint32_t* d_cell_kp, int ini_thr, int min_thr, int frame0, int n_frames, hipStream_t s)
{
    const dim3 grid = n_frames < 8 ? dim3((unsigned)g.n_cells, (unsigned)n_frames) : dim3(8u * (unsigned)g.n_cells, (unsigned)((n_frames + 7) / 8));
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d_pyr, gg, d_cells, d_cell_cnt, d_cell_kp, ini_thr, min_thr,
                       n_frames);
}

} // namespace mslam
