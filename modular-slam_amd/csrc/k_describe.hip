// k_describe.hip — intensity-centroid orientation + rotated-BRIEF descriptor + output packing.
//
// Replaces, per keypoint: ic_angle (reference distributed_cv_feature.cpp:543-570, on the UNBLURRED
// level, :977), compute_orb_descriptor scalar branch (:572-629, on the blurred level, :797-801),
// correct_keypoint_scale (:1166-1179) and the output packing of detect() (:1200-1214).
//
// One wave per keypoint (see k_describe).  Descriptor: lane i evaluates pairs i, i+64, i+128, i+192, so
// the four __ballot masks ARE descriptor bytes 0-7, 8-15, 16-23, 24-31 (bit i of byte j = pair 8j+i,
// LSB first, :614-625).  All float arithmetic is individually rounded (no FMA) and follows
// the reference expression order; cvRound = round-half-even, cvFloor = floor.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>
#include "../../include/mslam_orb_pattern.h"
#include "../../include/mslam_sincos.h"

namespace mslam
{

// the sampling pattern as floats (x0, y0, x1, y1 per pair): the reference multiplies float table entries
__constant__ __attribute__((aligned(16))) float c_pattern_f[1024] = MSLAM_ORB_PATTERN_INIT;

// cv::fastAtan2 (OpenCV core, atan_f32), degrees in [0, 360)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    constexpr float scale = (float)(180 / 3.1415926535897932384626433832795);
    constexpr float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    constexpr float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float)2.2204460492503131e-16; // (float)DBL_EPSILON
    float a, c, c2;
    if(ax >= ay)
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    else
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fadd_rn(__fmul_rn(p7, c2), p5);
    a = __fadd_rn(__fmul_rn(a, c2), p3);
    a = __fadd_rn(__fmul_rn(a, c2), p1);
    a = __fmul_rn(a, c);
    if(!(ax >= ay))
        a = __fsub_rn(90.f, a);
    if(x < 0)
        a = __fsub_rn(180.f, a);
    if(y < 0)
        a = __fsub_rn(360.f, a);
    return a;
}

// util::cos / util::sin (:456-503): degree-4 polynomial with quadrant folding
__device__ __forceinline__ float poly_cos(float v)
{
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(c1, __fmul_rn(v2, __fadd_rn(c2, __fmul_rn(c3, v2))));
}
__device__ __forceinline__ float util_cos(float v)
{
    constexpr float PI = 3.14159265358979f;
    constexpr float PI_2 = PI / 2.0f, TWO_PI = 2.0f * PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    v = __fsub_rn(v, __fmul_rn((float)(int)floorf(__fmul_rn(v, INV_TWO_PI)), TWO_PI));
    v = (0.0f < v) ? v : -v;
    if(v < PI_2)
        return poly_cos(v);
    else if(v < PI)
        return -poly_cos(__fsub_rn(PI, v));
    else if(v < THREE_PI_2)
        return -poly_cos(__fsub_rn(v, PI));
    else
        return poly_cos(__fsub_rn(TWO_PI, v));
}
__device__ __forceinline__ float util_sin(float v)
{
    constexpr float PI_2 = 3.14159265358979f / 2.0f;
    return util_cos(__fsub_rn(PI_2, v));
}

constexpr int kPatchR = 18;               // the rotated pattern reaches |18| (its radius is 18.38; the reference keeps a 19-px border)
constexpr int kPatchRows = 2 * kPatchR + 1; // 37
constexpr int kPatchRowsT = kPatchRows + 1;  // tiled plane: the fetched window starts at an even row (whole tile rows): 38
constexpr int kPatchDw = 12;              // 48-byte rows: 37 needed bytes from a 16-byte aligned start, or from 12 bytes past one
#ifndef MSLAM_DISC_DW
#define MSLAM_DISC_DW 16
#endif
constexpr int kPatchDwT = 16;             // tiled blurred plane (below): up to four aligned 16-byte chunks per row
constexpr int kPatchBufsT = 4;            // ... and a 4-slot ring (37 x 64 bytes per slot: 37.9 KB per workgroup, 4 workgroups per CU)
constexpr int kPatchBufs = 5;            // LDS patch ring per wave: one being sampled, four in flight (3 / 4 / 5 slots of 1776 bytes: 0.505 / 0.493 / 0.486 ms per 500 frames; 35.5 KB per workgroup = 4 workgroups per CU: with 5 (4 slots) the frames in flight per XCD outgrow its L2 and the memory-side reads rise from 1.9 to 2.2 MB per frame)
constexpr int kBlocksPerFrame = 32;

// dst[lane sel] = val (both wave-uniform): v_writelane_b32.  Inline asm (this compiler has no builtin for it): the lane select
// goes through M0 (a VOP3 instruction takes one SGPR besides it) — saved and restored around the instruction, because M0 is
// a register the compiler manages on its own (the LDS-DMA base) and does not track through asm clobbers — and the s_nop
// covers the wait states gfx950 wants between a vector instruction that wrote `val`'s scalar register (a ballot, a
// v_readlane) and a vector instruction that reads it: the compiler inserts them for its own instructions, not in front of
// inline asm.
__device__ __forceinline__ uint32_t write_lane(uint32_t dst, uint32_t val, int sel)
{
    uint32_t keep_m0;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(dst), "=&s"(keep_m0)
                 : "s"(val), "s"(sel));
    return dst;
}

// 64-lane integer sum with DPP adds (VALU only, no LDS crossbar); the total lands in lane 63
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xE, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xC, true); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true); // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true); // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}


// two independent sums at once: the DPP adds of one fill the wait states of the other
__device__ __forceinline__ void wave_sum_dpp2(int a, int b, int& ra, int& rb)
{
    a += __builtin_amdgcn_update_dpp(0, a, 0x111, 0xF, 0xF, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x111, 0xF, 0xF, true);
    a += __builtin_amdgcn_update_dpp(0, a, 0x112, 0xF, 0xF, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x112, 0xF, 0xF, true);
    a += __builtin_amdgcn_update_dpp(0, a, 0x114, 0xF, 0xE, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x114, 0xF, 0xE, true);
    a += __builtin_amdgcn_update_dpp(0, a, 0x118, 0xF, 0xC, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x118, 0xF, 0xC, true);
    a += __builtin_amdgcn_update_dpp(0, a, 0x142, 0xA, 0xF, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x142, 0xA, 0xF, true);
    a += __builtin_amdgcn_update_dpp(0, a, 0x143, 0xC, 0xF, true);
    b += __builtin_amdgcn_update_dpp(0, b, 0x143, 0xC, 0xF, true);
    ra = __builtin_amdgcn_readlane(a, 63);
    rb = __builtin_amdgcn_readlane(b, 63);
}

// Every WAVE is an independent worker (no workgroup barriers): it takes kBatch keypoints at a time through
//  0. one LANE per keypoint: which level, which candidate word, the byte offsets of its two windows (kept in that
//     lane's registers and broadcast later with v_readlane);
//  A. moments, the whole wave on one keypoint: the radius-15 disc of the unblurred level comes in by LDS-DMA (31 rows
//     x 48 bytes, two global_load_lds_dwordx4 per keypoint, three windows in flight in a per-wave LDS ring); lane t
//     (+64k) reads dword (row t/8, column group t%8) and folds it into m10 / m01 with two signed v_dot4 against
//     per-lane weight bytes (u resp. v inside the disc, 0 outside; host-built table).  Pixels are biased by -128
//     (xor 0x80) to fit i8; the bias cancels exactly because the disc is symmetric (sum of u = sum of v = 0).
//     Integer sums: order-free.
//  B. one LANE per keypoint: fastAtan2, the f64 degree->radian product, util::cos/sin (or mslam_sincos_f32 in the
//     cv::ORB mode) and the scalar outputs (coordinates, octave, angle, response) — the wave-uniform float work of
//     phase C is thereby done once per keypoint instead of once per lane.
//  C. descriptors, the whole wave on one keypoint: the 37 x 37 blurred patch comes in by LDS-DMA as well (37 rows x
//     48 bytes, two instructions, same ring), the 512 rotated sample points are LDS byte gathers (both points of a
//     pair rotated with packed-f32 arithmetic), and four ballots are the 32 descriptor bytes.
// Everything a wave needs about "its" keypoint is wave-uniform and kept in SGPRs (readlane).  What bounds the kernel
// is the L2 -> LDS window traffic, not arithmetic: DESIGN.md §4.6.
constexpr int kBatch = 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// TILED: the blurred plane is stored in kTileW x kTileH pixel tiles (one 128-byte line each, common.hpp: tiled_off) — a
// 37-row patch then touches fewer lines than its 37 rows x 1.4 (row-major: every patch row is a line of its own, and what
// this kernel pays for is lines).  A tiled row can only be fetched in aligned 16-byte chunks: three when the patch starts at
// byte <= 11 of its first chunk, else four (LDS row pitch 48 / 64 bytes, wave-uniform per keypoint).
template <bool TILED>
__global__ __launch_bounds__(256) void k_describe(Geometry g, DescArgs a, int bpf, int kb)
{
    constexpr int kBufs = TILED ? kPatchBufsT : kPatchBufs;
    constexpr int kSlotDw = TILED ? kPatchRowsT * kPatchDwT : kPatchRows * kPatchDw;
    constexpr int kDiscDw = TILED ? MSLAM_DISC_DW : 12; // LDS row pitch of the raw disc window in dwords
    __shared__ __attribute__((aligned(16))) uint32_t patch[4][kBufs][kSlotDw];

    // XCD-aware mapping: workgroups go to the 8 XCDs round-robin by linear id, so all kBlocksPerFrame
    // workgroups of a frame are given ids with the same (id & 7): the two level slabs of a frame (1.9 MB)
    // are then gathered through ONE 4 MB L2 instead of being pulled into all eight.
    // (kb < kBatch: a handful of frames — the synchronous single-frame calls: fewer keypoints per wave, more workgroups, and
    // the workgroups of the one frame on all XCDs: that launch runs at the latency of one wave's batch)
    const bool spread = kb < kBatch;
    const int xcd = blockIdx.x & 7, slot = spread ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    const int f_local = spread ? slot / bpf : (slot / bpf) * 8 + xcd;
    if(f_local >= a.n_frames)
        return;
    const int bx = slot % bpf;
    const size_t frame = (size_t)f_local + g.frame0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // levels are concatenated in order (:787-808)
    const uint32_t* sel_cnt = a.sel_cnt + frame * g.n_levels;
    int total = 0;
    for(int l = 0; l < g.n_levels; ++l)
        total += (int)sel_cnt[l];
    if(bx == 0 && threadIdx.x == 0)
    {
        a.count[frame] = min(total, a.max_kp);
        if(total > a.max_kp)
            atomicOr(a.flags, kFlagKpOverflow);
        // Mirror mode (the synchronous single-frame call): the host reads count and flags ONLY from the mapped block, so this
        // snapshot must see every flag of the call.  It does because k_describe is the LAST kernel of the single-frame sequence
        // (api.hip: enqueue_detect, both detector modes) and this thread is its only flag writer: the flags of the earlier
        // kernels are final when it starts.  A kernel appended behind k_describe, or a second flag writer inside it, must send
        // the flags by a copy of its own instead (tests/test_gpu_parity.py::test_capacity_is_loud runs both result paths).
        if(a.h_mirror)
        {
            reinterpret_cast<int32_t*>(a.h_mirror)[0] = min(total, a.max_kp);
            reinterpret_cast<uint32_t*>(a.h_mirror)[1] = __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | (total > a.max_kp ? kFlagKpOverflow : 0u);
        }
    }
    const int n_kp = min(total, a.max_kp);

    // per-lane constants: disc weights and this lane's four sampling pairs
    uint32_t wu[4], wv[4];
    f32x2 patX[4], patY[4]; // (x0, x1) and (y0, y1) of the lane's four pairs: operands of the packed-f32 rotation
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        wu[k] = a.orient_w[lane + 64 * k];
        wv[k] = a.orient_w[256 + lane + 64 * k];
        const float4 q = reinterpret_cast<const float4*>(c_pattern_f)[lane + 64 * k];
        patX[k] = f32x2{q.x, q.z};
        patY[k] = f32x2{q.y, q.w};
    }
    const uint8_t* pyr = a.pyr + frame * g.slab;
    const uint8_t* blur = a.blur + frame * g.slab;
    auto bc = [&](uint32_t v, int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, k); };
    // lane l (< n_levels <= 16) keeps level l's cumulative keypoint count, row pitch, slab offset and scale
    int lv_cum = 0;
    uint32_t lv_pitch = 16, lv_offset = 0, lv_scale = 0;
    {
        const int ll = min(lane, g.n_levels - 1);
        int c = lane < g.n_levels ? (int)sel_cnt[ll] : 0;
#pragma unroll
        for(int o = 1; o < kMaxLevels; o <<= 1)
        {
            const int t = __shfl_up(c, o);
            c += lane >= o ? t : 0;
        }
        lv_cum = c;
        const LevelGeom* lvp = &g.lv[0];
        lv_pitch = (uint32_t)lvp[ll].pitch;
        lv_offset = (uint32_t)lvp[ll].offset;
        lv_scale = __float_as_uint(lvp[ll].scale);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // the table is in registers: no waits for it inside the batch loop

    // A workgroup takes 64 consecutive list positions at a time and deals them to its four waves round-robin (wave w
    // works on positions w, w+4, ...): neighbours in the quadtree's list order are neighbours in the image, and the
    // four waves run in step, so at any moment they are fetching windows that share 128-byte lines.  The window
    // fetches are what this kernel's time goes to; giving each wave 16 consecutive positions instead
    // measured 0.529 vs 0.487 ms per 500 frames.  (Ranking the 64 by (level, column band, row) first bought another
    // 0.01 ms and cost 0.02 ms in the serial head of every batch; 8- and 16-wave workgroups: 0.51 / 0.66 ms.)
    constexpr int kStr = 4;
    for(int base = bx * 4 * kb + wave; base < n_kp; base += bpf * 4 * kb)
    {
        const int n_here = min(kb, (n_kp - base + kStr - 1) / kStr); // wave-uniform

        // ---- 0. one lane per keypoint: which level, which candidate word, and everything phases A and C need to
        //         address its two windows (byte offsets inside the frame slab, row pitch, sub-16 shifts).  The level
        //         loop runs on a wave-uniform counter, so the level geometry comes through scalar loads ONCE per
        //         batch; the per-keypoint phases then only broadcast lane k's registers (v_readlane).
        uint32_t my_kp = 0;
        int my_level = 0;
        uint32_t my_doff = 0, my_poff = 0, my_pitch = 16, my_sh = 0, my_lut = 0;
        float my_scale = 1.f, my_resp = 0.f;
        {
            const int idx = base + lane * kStr; // position in the frame's concatenated keypoint list (:787-808)
            // the level of list position idx = the number of levels whose cumulative count it has reached (one broadcast, one
            // compare-and-add per level); the level's pitch / offset / scale and the list position of its first keypoint then
            // come from the lanes that keep them (ds_bpermute: the LDS crossbar, not the vector ALU) — the select chain this
            // replaces was ten vector instructions per level
            uint32_t lofs = 0;
            {
                int lvl = 0;
#pragma unroll 1
                for(int l = 0; l + 1 < g.n_levels; ++l)
                    lvl += idx >= __builtin_amdgcn_readlane(lv_cum, l) ? 1 : 0;
                my_level = lvl;
                // (every lane takes part in the exchange — a ds_bpermute returns 0 for a source lane that is switched off, so it
                // must not sit in the arm of a per-lane condition)
                const int prev_cum = __shfl(lv_cum, lvl > 0 ? lvl - 1 : 0);
                const int first = lvl > 0 ? prev_cum : 0; // list position of the level's first keypoint
                my_pitch = (uint32_t)__shfl((int)lv_pitch, lvl);
                lofs = (uint32_t)__shfl((int)lv_offset, lvl);
                my_scale = __uint_as_float((uint32_t)__shfl((int)lv_scale, lvl));
                my_kp = (uint32_t)(idx - first); // index inside the level, replaced by the word below
            }
            if(lane < n_here)
            {
                const size_t si = (frame * g.n_levels + my_level) * (size_t)a.cand_cap + my_kp;
                my_kp = a.sel[si];
                my_resp = a.sel_resp ? a.sel_resp[si] : (float)kp_score(my_kp);
                const int px = kp_x(my_kp) + kBorder, py = kp_y(my_kp) + kBorder; // :966-967
                my_doff = lofs + (uint32_t)(py - 15) * my_pitch + (uint32_t)((px - 15) & ~15);
                my_sh = (uint32_t)((px - 15) & 15) | ((uint32_t)((px - kPatchR) & 15) << 4);
                if(TILED)
                {
                    // Tiled plane (common.hpp: tiled_off): byte (x, y) = (y & ~1) pitch + (y & 1) 64 + 2 (x & ~63) + (x & 63).  The
                    // window is fetched from the EVEN row ya & ~1 on (38 rows) in 16-byte chunks starting at xa = 64 A + 16 a:
                    // everything that depends on the keypoint only is folded into ONE scalar offset (my_poff) and the four
                    // column terms of chunk c = 0 .. 3, ((a + c) + ((a + c) & 4)) in units of 16 bytes, into the bytes of my_lut
                    // — what is left per lane is a multiply-add on lane constants (dma_patch)
                    const uint32_t xa = (uint32_t)((px - kPatchR) & ~15), ya = (uint32_t)(py - kPatchR);
                    my_poff = lofs + (ya & ~1u) * my_pitch + 2u * (xa & ~63u);
                    const uint32_t sum = 0x03020100u + ((xa >> 4) & 3u) * 0x01010101u;       // a + c per byte
                    my_lut = sum + (((sum + 0x7C7C7C7Cu) & 0x80808080u) >> 5);              // + 4 where a + c >= 4
                    my_sh |= (ya & 1u) << 8;                                                 // the patch's first row inside the window
                }
                else
                    my_poff = lofs + (uint32_t)(py - kPatchR) * my_pitch + (uint32_t)((px - kPatchR) & ~15);
            }
        }
        // The candidate words are in registers from here on: without this explicit wait the compiler re-inserts
        // "s_waitcnt vmcnt(0)" in front of every later use in a conditional block — i.e. in front of each LDS-DMA
        // issue of the prologues below, which would drain the DMA just issued and serialise them.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        // ---- A. moments
        int my_m10 = 0, my_m01 = 0;
        {
            // the 31 x 31 window of the unblurred level comes in by LDS-DMA as 31 rows x 48 bytes from a 16-byte
            // aligned start (two global_load_lds_dwordx4 per keypoint into the ring phase C uses later); lane t (+64q)
            // then reads dword (row t/8, column group t%8) of the window as two aligned LDS dwords + v_alignbyte
            auto dma_disc = [&](int k, int buf) {
                const uint8_t* src = pyr + bc(my_doff, k);
                const uint32_t pitch = bc(my_pitch, k);
#pragma unroll
                for(int q = 0; q < 2; ++q)
                {
                    const uint32_t t = (uint32_t)lane + 64u * q;
                    if(kDiscDw == 16)
                    {
                        // four lanes per row (the fourth idle): every quad of lanes asks for ONE row, i.e. one cache line.  Row
                        // (lane >> 2) + 16 q: the lane's part of the address is one multiply-add on lane constants, the
                        // instruction's part (16 q rows) is scalar
                        const uint8_t* sq = src + (size_t)(16u * q * pitch);
                        asm volatile("" : "+s"(sq));
                        if(t < 124u && (t & 3u) != 3u)
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void*)(sq + (__umul24((uint32_t)lane >> 2, pitch) + 16u * ((uint32_t)lane & 3u))),
                                (__attribute__((address_space(3))) void*)&patch[wave][buf][q * 256], 16, 0, 0);
                    }
                    else if(t < 93u)
                    {
                        const uint32_t row = (t * 21846u) >> 16; // t / 3
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(src + __umul24(row, pitch) + 16u * (t - 3u * row)),
                            (__attribute__((address_space(3))) void*)&patch[wave][buf][q * 256], 16, 0, 0);
                    }
                }
            };
            auto reduce = [&](int k, int buf) {
                const int sh = (int)(bc(my_sh, k) & 15u);
                const uint32_t* d = patch[wave][buf] + (sh >> 2);
                int m10 = 0, m01 = 0;
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
                    const int t = lane + 64 * q;
                    uint32_t x = 0x80808080u;
                    if(t < 31 * 8)
                    {
                        const uint32_t* e = d + (t >> 3) * kDiscDw + (t & 7);
                        x = __builtin_amdgcn_alignbyte(e[1], e[0], sh & 3);
                    }
                    x ^= 0x80808080u;
                    m10 = __builtin_amdgcn_sdot4((int)x, (int)wu[q], m10, false);
                    m01 = __builtin_amdgcn_sdot4((int)x, (int)wv[q], m01, false);
                }
                int t10, t01;
                wave_sum_dpp2(m10, m01, t10, t01);
                my_m10 = (int)write_lane((uint32_t)my_m10, (uint32_t)t10, k);
                my_m01 = (int)write_lane((uint32_t)my_m01, (uint32_t)t01, k);
            };
            constexpr int kDepthA = kBufs - 1;
#pragma unroll
            for(int i = 0; i < kDepthA; ++i)
                if(i < n_here)
                    dma_disc(i, i);
            for(int k = 0; k < n_here; ++k)
            {
                const int younger = min(kDepthA, n_here - 1 - k); // windows issued after window k: 2 DMAs each
                if(k + kDepthA < n_here)
                    dma_disc(k + kDepthA, (k + kDepthA) % kBufs);
                if(younger >= 4)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
                else if(younger == 3)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
                else if(younger == 2)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
                else if(younger == 1)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
                else
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
                __builtin_amdgcn_wave_barrier();
                reduce(k, k % kBufs);
                __builtin_amdgcn_wave_barrier(); // the reads are done before the slot is refilled
            }
        }

        // The blurred patches of the batch's first keypoints are requested BEFORE phase B: they only need addresses,
        // and phase B (one lane per keypoint, scalar-like float code) then runs under their latency.  (All ring
        // slots are free here: phase A has reduced its last window.)
        auto dma_patch = [&](int k, int buf) {
            const uint8_t* bsrc = blur + bc(my_poff, k);
            const uint32_t pitch = bc(my_pitch, k);
            // the 37 needed bytes of a row start `sh` bytes into its first 16-byte chunk;
            // three 16-byte chunks per row: 111 lanes = TWO DMA instructions, rows packed at a 48-byte pitch.  When sh > 11
            // the chunks start 12 bytes past the aligned one (bytes [12, 60) of the row's 64-byte span hold the 37 needed
            // ones): LDS-DMA b128 only needs a 4-byte aligned global address.  (A 12-byte DMA would not do: b96 writes its
            // 12 bytes at a 16-byte lane stride in LDS.)
            const uint32_t shp = (bc(my_sh, k) >> 4) & 15u;
            if(TILED)
            {
                // (see phase 0) lane part of a chunk's address: (row & ~1) pitch + (row & 1) 64 + 16 lut[c]; the rows of the
                // instructions after the first differ by an even number (16 resp. 20), i.e. by a scalar
                const uint32_t lut = bc(my_lut, k);
                if(shp > 11u)
                {
                    // four chunks per row: chunk t = lane + 64 q is (row (lane >> 2) + 16 q, c = lane & 3); 38 rows = 152 chunks
                    const uint32_t lo = (__builtin_amdgcn_ubfe(lut, 8u * ((uint32_t)lane & 3u), 8u) << 4) +
                                        (__umul24(((uint32_t)lane >> 3) << 1, pitch) + (((uint32_t)lane & 4u) << 4));
#pragma unroll
                    for(int q = 0; q < 3; ++q)
                    {
                        const uint8_t* bq = bsrc + (size_t)(16u * q * pitch);
                        asm volatile("" : "+s"(bq)); // (a scalar base per instruction: else the compiler adds the rows to the lane offsets in 64-bit vector adds)
                        if((uint32_t)lane + 64u * q < (uint32_t)(kPatchRowsT * 4))
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bq + lo),
                                                             (__attribute__((address_space(3))) void*)&patch[wave][buf][q * 256], 16, 0, 0);
                    }
                }
                else
                {
                    // three chunks per row, 60 lanes per instruction: chunk t = lane + 60 q is (row lane / 3 + 20 q, c = lane % 3)
                    const uint32_t r3 = ((uint32_t)lane * 21846u) >> 16, c3 = (uint32_t)lane - 3u * r3;
                    const uint32_t lo = (__builtin_amdgcn_ubfe(lut, 8u * c3, 8u) << 4) + (__umul24(r3 & ~1u, pitch) + ((r3 & 1u) << 6));
#pragma unroll
                    for(int q = 0; q < 2; ++q)
                    {
                        const uint8_t* bq = bsrc + (size_t)(20u * q * pitch);
                        asm volatile("" : "+s"(bq));
                        if((uint32_t)lane < (q == 0 ? 60u : (uint32_t)(kPatchRowsT * 3 - 60)))
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bq + lo),
                                                             (__attribute__((address_space(3))) void*)&patch[wave][buf][q * 240], 16, 0, 0);
                    }
                }
                return;
            }
            bsrc += shp > 11u ? 12 : 0;
#pragma unroll
            for(int q = 0; q < 2; ++q)
            {
                const uint32_t t = (uint32_t)lane + 64u * q;
                const uint32_t row = (t * 21846u) >> 16; // t / 3
                if(t < (uint32_t)(kPatchRows * 3))
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(bsrc + __umul24(row, pitch) + 16u * (t - 3u * row)),
                        (__attribute__((address_space(3))) void*)&patch[wave][buf][q * 256], 16, 0, 0);
            }
        };
        constexpr int kDepth = kBufs - 1; // patches in flight beside the one being sampled
#pragma unroll
        for(int i = 0; i < kDepth; ++i)
            if(i < n_here)
                dma_patch(i, i);

        // ---- B. one lane per keypoint: angle, cos/sin, scalar outputs (kept in registers and stored after phase C:
        //         stores issued here would sit between the patch DMAs in the vmcnt queue and make phase C's counted
        //         waits stricter than they need to be)
        float my_ca = 0.f, my_sa = 0.f, my_angle = 0.f, my_ox = 0.f, my_oy = 0.f;
        if(lane < n_here)
        {
            my_angle = fast_atan2_deg((float)my_m01, (float)my_m10);
            if(a.cv_mode)
            {
                // orb.cpp computeOrbDescriptors: angle *= (float)(CV_PI/180.f); a = (float)cos(angle), b = (float)sin(angle)
                const float rad = __fmul_rn(my_angle, (float)(3.1415926535897932384626433832795 / 180.f));
                mslam_sincos_f32(rad, &my_sa, &my_ca);
            }
            else
            {
                const float rad = (float)((double)my_angle * 3.14159265358979323846 / 180.0); // :574
                my_ca = util_cos(rad);
                my_sa = util_sin(rad);
            }
            const float fx = (float)(kp_x(my_kp) + kBorder), fy = (float)(kp_y(my_kp) + kBorder);
            // correct_keypoint_scale (:1166-1179): float multiply, skipped for level 0
            my_ox = my_level == 0 ? fx : __fmul_rn(fx, my_scale); // (cv mode: level 0 has scale 1.0f, same result)
            my_oy = my_level == 0 ? fy : __fmul_rn(fy, my_scale);
        }

        // ---- C. descriptors
        uint32_t desc_lo = 0, desc_hi = 0;
        {
            // (the patches were requested by dma_patch above: 37 rows x 48 bytes each, lane-linear in LDS)
            auto describe = [&](int k, int buf) {
                const float ca = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_ca), k));
                const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sa), k));
                // Both points of a pair are rotated at once with packed f32 multiplies / adds (v_pk_mul_f32, v_pk_add_f32:
                // every product and sum is rounded on its own, as in the reference; the file is built with
                // -ffp-contract=off).  cvRound (round-half-even) is the magic-number add: the f32 sum v + 1.5 * 2^23 has the
                // bit pattern 0x4B400000 + rint(v) for |v| < 2^22, so row * 64 + col comes out of one shift-add on the bit
                // patterns and the constant 65 * 0x4B400000 is folded into the (wave-uniform) centre offset, mod 2^32.
                f32x2 ca2 = f32x2{ca, ca}, sa2 = f32x2{sa, sa};
                asm volatile("" : "+v"(ca2), "+v"(sa2)); // keep them vector register pairs (not re-associated onto the scalars)
                const f32x2 magic = f32x2{12582912.f, 12582912.f};
                const uint32_t lp_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)patch[wave][buf];
                const uint32_t shk = bc(my_sh, k);
                const uint32_t shp = (shk >> 4) & 15u, odd_row = (shk >> 8) & 1u;
                unsigned long long bits[4];
                // row-major plane: 48-byte row pitch; the row starts at the aligned chunk (shp <= 11) or 12 bytes past it (see
                // dma_patch).  Tiled plane: the row always starts at the aligned chunk; 48-byte pitch when shp <= 11, else 64.
                auto sample = [&](auto wide) {
                    constexpr bool WIDE = decltype(wide)::value;
                    constexpr uint32_t P = WIDE ? 64u : 48u;
                    // LDS byte address of a sample = patch base + row * P + column.  The magic-number sums carry rint(v) in their
                    // low bits on top of 0x4B400000: row * P as a 24-bit multiply-add (which only sees 0x400000 + row) or a shift-add
                    // with the column's bit pattern as the addend — one full-rate instruction per sample (the compiler made a
                    // 32-bit v_mul_lo, a quarter-rate instruction, out of (r << 5) + (r << 4)) — and everything constant in `ctr`
                    const uint32_t ctr = lp_lds + (uint32_t)(kPatchR * P + kPatchR) + (TILED ? shp + odd_row * P : shp <= 11u ? shp : shp - 12u) -
                                         (WIDE ? P * 0x4B400000u : P * 0x00400000u) - 0x4B400000u;
#pragma unroll
                    for(int t = 0; t < 4; ++t)
                    {
                        // GET_VALUE (:603-605): row = cvRound(x*sin + y*cos), col = cvRound(x*cos - y*sin)
                        const f32x2 rr = (patX[t] * sa2 + patY[t] * ca2) + magic;
                        const f32x2 cc = (patX[t] * ca2 - patY[t] * sa2) + magic;
                        const uint32_t r0 = __float_as_uint(rr.x), r1 = __float_as_uint(rr.y);
                        const uint32_t i0 = (WIDE ? (r0 << 6) + __float_as_uint(cc.x) : __umul24(r0, P) + __float_as_uint(cc.x)) + ctr;
                        const uint32_t i1 = (WIDE ? (r1 << 6) + __float_as_uint(cc.y) : __umul24(r1, P) + __float_as_uint(cc.y)) + ctr;
                        const int v0 = *lds_ptr<const __attribute__((address_space(3))) uint8_t*>(i0);
                        const int v1 = *lds_ptr<const __attribute__((address_space(3))) uint8_t*>(i1);
                        bits[t] = __ballot(v0 < v1);
                    }
                };
                if(TILED && shp > 11u)
                {
                    asm volatile("");
                    sample(std::true_type{});
                }
                else
                {
                    asm volatile("");
                    sample(std::false_type{});
                }
                // the batch's descriptors are collected in lanes (4 k + t holds dword pair t of keypoint k) and stored once per batch — one store instruction instead of sixteen on the
                // memory pipeline the window fetches are queued on
#pragma unroll
                for(int t = 0; t < 4; ++t) // (v_writelane: one instruction per dword, the ballots are scalars already)
                {
                    desc_lo = write_lane(desc_lo, (uint32_t)bits[t], 4 * k + t);
                    desc_hi = write_lane(desc_hi, (uint32_t)(bits[t] >> 32), 4 * k + t);
                }
            };
            static_assert(kBufs >= 2 && kBufs <= 5, "the vmcnt immediates of phases A and C cover up to four younger windows");
            static_assert(kDepth >= 1 && kDepth <= 5, "the vmcnt immediates below cover up to 5 patches in flight");
            for(int k = 0; k < n_here; ++k)
            {
                // patch k must have landed: LDS-DMA is counted by vmcnt, in issue order, so "at most the DMAs issued
                // after it outstanding" is the condition (3 instructions per patch; a younger store only makes the
                // wait a little longer than necessary)
                const int younger = min(kDepth, n_here - 1 - k);
                if(k + kDepth < n_here)
                    dma_patch(k + kDepth, (k + kDepth) % kBufs); // its ring slot was sampled in iteration k - 1
                // (a patch is 2 or 3 instructions: counting 2 for every younger one is the safe side)
                if(younger >= 5)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 10);
                else if(younger == 4)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
                else if(younger == 3)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 6);
                else if(younger == 2)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
                else if(younger == 1)
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 2);
                else
                    __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
                __builtin_amdgcn_wave_barrier();
                describe(k, k % kBufs);
                __builtin_amdgcn_wave_barrier(); // the gathers are done before the slot is refilled
            }
        }
        if(lane < 4 * n_here)
            reinterpret_cast<uint2*>(a.desc + (frame * (size_t)a.max_kp + base + (lane >> 2) * kStr) * 32)[lane & 3] = make_uint2(desc_lo, desc_hi);
        if(lane < n_here)
        {
            const size_t o = frame * (size_t)a.max_kp + base + lane * kStr;
            reinterpret_cast<float2*>(a.xy)[o] = make_float2(my_ox, my_oy);
            a.octave[o] = my_level;
            a.angle[o] = my_angle;
            a.response[o] = my_resp;
        }
        if(a.h_mirror) // wave-uniform: the same records into the mapped host block (layout of mslam_hip_detect's staging block)
        {
            const size_t K = (size_t)a.max_kp;
            uint8_t* h = a.h_mirror + 16;
            if(lane < 4 * n_here)
                reinterpret_cast<uint2*>(h + K * 8 + (size_t)(base + (lane >> 2) * kStr) * 32)[lane & 3] = make_uint2(desc_lo, desc_hi);
            if(lane < n_here)
            {
                const size_t o = (size_t)(base + lane * kStr);
                reinterpret_cast<float2*>(h)[o] = make_float2(my_ox, my_oy);
                reinterpret_cast<int32_t*>(h + K * 40)[o] = my_level;
                reinterpret_cast<float*>(h + K * 44)[o] = my_angle;
                reinterpret_cast<float*>(h + K * 48)[o] = my_resp;
            }
        }
    }
}

void launch_describe(const Geometry& g, const DescArgs& a, int frame0, int n_frames, hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    DescArgs aa = a;
    aa.n_frames = n_frames;
    static const int bpf_env = [] { const char* e = getenv("MSLAM_DESC_BPF"); return e ? atoi(e) : 0; }();
    // a handful of frames (the synchronous single-frame call): twice the workgroups per frame, half the keypoints per wave
    // a handful of frames (the synchronous single-frame call): 4 keypoints per wave and batch instead of 16, 128 workgroups per frame
    static const int kb_env = [] { const char* e = getenv("MSLAM_DESC_KB_SMALL"); return e ? atoi(e) : 0; }();
    const bool small = n_frames < 8;
    const int kb = small ? (kb_env >= 1 && kb_env <= kBatch ? kb_env : 4) : kBatch;
    const int bpf = bpf_env ? bpf_env : small ? 4 * kBlocksPerFrame : kBlocksPerFrame;
    const unsigned grid = kb < kBatch ? (unsigned)n_frames * (unsigned)bpf : (unsigned)((n_frames + 7) / 8) * 8u * (unsigned)bpf;
    if(g.blur_tiled)
        hipLaunchKernelGGL(k_describe<true>, dim3(grid), dim3(256), 0, s, gg, aa, bpf, kb);
    else
        hipLaunchKernelGGL(k_describe<false>, dim3(grid), dim3(256), 0, s, gg, aa, bpf, kb);
}

} // namespace mslam
