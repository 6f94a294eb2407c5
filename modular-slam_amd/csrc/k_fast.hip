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
#include "common.hpp"
#include "arc_score.hpp"

namespace mslam
{

constexpr int kTileP = 80; // tile row pitch in bytes: five 16-byte LDS-DMA chunks from column x0 - 3 on (a 16-byte boundary),
constexpr int kTileX = 3;  // i.e. tile byte = column + 3: the tested columns 3+4i .. 6+4i are bytes 6+4i .. 9+4i
constexpr int kScP = 68;   // score-map row pitch

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
    __shared__ __attribute__((aligned(4))) uint16_t cand[64 * 64 + 2];
    constexpr uint32_t kDump = 64 * 64; // write-only slot for rejected pixels
    __shared__ uint32_t bitmap[64 * 2];
    __shared__ uint32_t n_cand;

    // XCD-aware mapping: workgroups are handed to the 8 XCDs round-robin by linear id, and neighbouring cells share
    // 128-byte lines (a 70-byte cell row straddles two of them, its neighbours use the rest).  With cells of one
    // frame spread over all XCDs every L2 fetched those lines again — 2.7x the level's bytes at the memory side
    // (profiles/r02_a_pmc_fetch_write_per_launch.json); all cells of a frame now get ids with the same (id & 7).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int f_local = (slot / g.n_cells) * 8 + xcd;
    if(f_local >= n_frames)
        return;
    const int cell_id = slot % g.n_cells;
    const size_t frame = (size_t)f_local + g.frame0;
    const CellDesc c = cells[cell_id];
    const LevelGeom& lv = g.lv[c.level];
    const int cw = c.cw, ch = c.ch;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // stage the sub-image by LDS-DMA: the cell's rows start at column x0 - 3 = 16 + 64 j, a 16-byte boundary of the level
    // plane (pitch and level offset are multiples of 16), so a row is five 16-byte chunks that go from global memory
    // straight into the tile — no registers, no vector instructions for the copy (round 2 staged through registers
    // with a 2-byte shift to make the tested columns dword-aligned: ~45 instructions per wave; the unshifted image
    // costs phase A three more v_alignbyte per step).  Chunk t = (row t / 5, chunk t % 5) lands at tile + 16 t.
    {
        const uint8_t* src = pyr + frame * g.slab + lv.offset + (size_t)c.y0 * lv.pitch + (c.x0 - 3);
        const int n_chunks = ch * 5;
#pragma unroll
        for(int p = 0; p < 2; ++p)
        {
            const int t = p * 256 + tid;
            if(t < n_chunks)
            {
                const int r = (t * 13108) >> 16, j = t - r * 5; // t / 5 for t < 16384
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(src + (size_t)r * lv.pitch + 16 * j),
                    (__attribute__((address_space(3))) void*)&tile[(p * 256 + wave * 64) * 16], 16, 0, 0);
            }
        }
    }
    // meanwhile: zero the score map (untested pixels must read 0, like FAST_t's zeroed row buffers) and the bitmap
    for(int i = tid; i < (66 * kScP + 15) / 16; i += 256)
        reinterpret_cast<uint4*>(sc)[i] = make_uint4(0u, 0u, 0u, 0u);
    if(tid < 128)
        bitmap[tid] = 0;
    if(tid == 0)
        n_cand = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the wave's DMA chunks have landed
    __syncthreads();

    // LDS byte offset of the list's fill counter (for the hand-written reservation in phase A)
    const uint32_t n_cand_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(&n_cand);
    uint32_t total = 0;
    for(int pass = 0; pass < 2; ++pass)
    {
        const int thr = pass == 0 ? ini_thr : min_thr;

        // ---- A. compass test + compaction, four pixels per lane: a wave covers 4 rows x 64 columns per step.
        //      Pixels are widened to u16 pairs and compared with packed 16-bit subtracts (sign bit = result).
        {
            const uint32_t* T = reinterpret_cast<const uint32_t*>(tile);
            // lane = (row sub, column group i4) of a step.  A full cell has 16 column groups (64 tested columns) and a wave
            // step covers 4 rows; the narrow cells that end a level's cell rows (and the short ones that end its cell columns)
            // use fewer groups per row and more rows per step — lg = log2(groups per row) — and only as many steps as they
            // have rows: 8 % of the steps of a 640x480 pyramid are saved.  (The order of the list is irrelevant: it is a
            // work queue, the output order comes from the bitmap.)
            const int ngrp = (cw - 6 + 3) >> 2;                                             // column groups that hold a tested column
            const int lg = ngrp > 8 ? 4 : ngrp > 4 ? 3 : ngrp > 2 ? 2 : ngrp > 1 ? 1 : 0; // (wave-uniform)
            const int i4 = lane & ((1 << lg) - 1), sub = lane >> lg;
            const int rps = 64 >> lg;                                                     // rows per wave step
            const int n_steps = (ch - 6 + rps - 1) >> (6 - lg);                            // row steps of the cell, dealt to the waves round-robin
            const uint32_t thr2 = (uint32_t)thr * 0x00010001u;
            // which of this lane's four columns 3+4i .. 6+4i are tested (x < cw - 3)
            const int xl = 3 + 4 * i4;
            // (as sign bits of the four bytes: the layout the packed compares deliver, see `keep` below)
            const uint32_t colmask = (xl < cw - 3 ? 0x80u : 0u) | (xl + 1 < cw - 3 ? 0x8000u : 0u) |
                                     (xl + 2 < cw - 3 ? 0x800000u : 0u) | (xl + 3 < cw - 3 ? 0x80000000u : 0u);
#pragma unroll 1
            for(int st = __builtin_amdgcn_readfirstlane(wave); st < n_steps; st += 4)
            {
                const int y = 3 + st * rps + sub;
                uint32_t keep = 0;
                if(colmask != 0 && y < ch - 3)
                {
                    // the four tested pixels are bytes 6+4i .. 9+4i of the row: dwords (i+1, i+2) shifted by two bytes
                    const uint32_t* row = T + y * 20 + i4;
                    const uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3];
                    const uint32_t C = __builtin_amdgcn_alignbyte(d2, d1, 2);
                    const uint32_t U = __builtin_amdgcn_alignbyte(row[2 - 3 * 20], row[1 - 3 * 20], 2);
                    const uint32_t D = __builtin_amdgcn_alignbyte(row[2 + 3 * 20], row[1 + 3 * 20], 2);
                    const uint32_t Lf = __builtin_amdgcn_alignbyte(d1, d0, 3); // columns x-3 of the four pixels
                    const uint32_t Rt = __builtin_amdgcn_alignbyte(d3, d2, 1); // columns x+3
                    uint32_t k[2];
#pragma unroll
                    for(int h = 0; h < 2; ++h)
                    {
                        const uint32_t sel = h == 0 ? 0x0c010c00u : 0x0c030c02u; // bytes (0,1) or (2,3) as u16 lanes
                        typedef short s16x2 __attribute__((ext_vector_type(2)));
                        // (the selector as a scalar operand: as a vector constant it holds a register the kernel, capped at 64, spills for)
                        auto w = [&](uint32_t v) {
                            uint32_t d;
                            asm("v_perm_b32 %0, 0, %1, %2" : "=v"(d) : "v"(v), "s"(sel));
                            return __builtin_bit_cast(s16x2, d);
                        };
                        const s16x2 t2 = __builtin_bit_cast(s16x2, thr2);
                        const s16x2 cc = w(C), hi = cc + t2, lo = cc - t2;
                        const s16x2 p0 = w(D), p4 = w(Rt), p8 = w(U), p12 = w(Lf);
                        // brighter: (p0 or p8 > c + t) and (p4 or p12 > c + t)  <=>  min(max(p0, p8), max(p4, p12)) > c + t
                        // darker : (p0 or p8 < c - t) and (p4 or p12 < c - t)  <=>  max(min(p0, p8), min(p4, p12)) < c - t
                        // (packed 16-bit min / max; the comparisons are the sign bits of packed subtractions)
                        const s16x2 mb = __builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12));
                        const s16x2 md = __builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12));
                        k[h] = __builtin_bit_cast(uint32_t, (s16x2)(hi - mb)) | __builtin_bit_cast(uint32_t, (s16x2)(md - lo));
                    }
                    // the four results are the sign bits of the 16-bit lanes of k[0], k[1]: one v_perm puts their high bytes
                    // side by side, pixel j's flag is then bit 8j + 7
                    keep = __builtin_amdgcn_perm(k[1], k[0], 0x07050301u) & colmask;
                }
                // compaction: wave-wide inclusive scan of the per-lane counts with DPP adds (no LDS, no ballots);
                // every lane then issues its four stores unconditionally, rejected pixels into a dump slot
                const uint32_t cnt = (uint32_t)__popc(keep);
                uint32_t inc = cnt;
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true); // row_shr:1
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true); // row_shr:2
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xE, true); // row_shr:4
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xC, true); // row_shr:8
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xA, 0xF, true); // row_bcast:15
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xC, 0xF, true); // row_bcast:31
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if(tot != 0)
                {
                    // one lane reserves the wave's range of the list.  Written out by hand: around `if(lane == 0) atomicAdd(..)`
                    // the compiler's atomic optimiser builds a wave reduction (mbcnt, bcnt, a multiply, two exec-mask
                    // regions: ~8 vector instructions per step) for a value that is wave-uniform already.
                    uint32_t base;
                    {
                        uint32_t got;
                        unsigned long long save;
                        const uint32_t add = tot; // (a vector register copy of the wave-uniform count)
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
                    uint32_t pos = base + inc - cnt;
                    const uint32_t yx = (uint32_t)((y << 8) | xl);
                    // slot = kept ? pos : dump as ONE v_bfi_b32 on an all-ones / all-zeros mask (v_bfe_i32), and the
                    // running slot advanced with a 24-bit multiply-add on the same mask: 3 VALU ops per store instead of
                    // the compare + select + shift chains the compiler builds (which also cost wait states here)
                    uint32_t pos2 = 2u * pos; // byte offset into cand[]
                    const uint32_t dump2 = 2u * kDump;
#pragma unroll
                    for(int k = 0; k < 4; ++k)
                    {
                        const int m = __builtin_amdgcn_sbfe((int)keep, 8 * k + 7, 1); // -1 when kept
                        uint32_t slot2;
                        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(slot2) : "v"(m), "v"(pos2), "s"(dump2));
                        *reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(cand) + slot2) = (uint16_t)(yx + k);
                        pos2 = (uint32_t)__mul24(m, -2) + pos2;
                    }
                }
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
            // (centre + bias) of both pixels; the subtraction below never borrows from the upper half
            const uint32_t vv = ((uint32_t)pa[0] + kArcBias) | (((uint32_t)pb[0] + kArcBias) << 16);
            uint32_t e[16];
#define MSLAM_E(k, off) e[k] = vv - ((uint32_t)pa[off] | ((uint32_t)pb[off] << 16))
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
            arc_score2(e, sa, sb);
            if(sa >= thr && sa > 0)
                sc[(ay - 2) * kScP + (ax - 2)] = (uint8_t)sa;
            if(sb >= thr && sb > 0)
                sc[(by - 2) * kScP + (bx - 2)] = (uint8_t)sb;
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
    const unsigned grid = (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)g.n_cells;
    Geometry gg = g;
    gg.frame0 = frame0;
    hipLaunchKernelGGL(k_fast_cells, dim3(grid), dim3(256), 0, s, d_pyr, gg, d_cells, d_cell_cnt, d_cell_kp, ini_thr, min_thr,
                       n_frames);
}

} // namespace mslam
