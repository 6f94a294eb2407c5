// k_level.hip — one pyramid level per launch, produced AND blurred in the same pass (round 3).
//
//   level 0 : BGR -> gray (frame.cpp:6-27)                                        + 7x7 sigma-2 blur (DCF:797-798)
//   level l : cv::resize(level l-1, INTER_LINEAR [_EXACT in the cv::ORB mode]) (DCF:839) + the same blur
//
// Why fused: the blur of a level was a kernel of its own that read the level plane back (1.17 MB of the 8.2 MB per frame
// the pipeline moved in round 2) and was bound by that read, not by its arithmetic.  Here the freshly produced row never
// leaves the registers before it is filtered: the raw plane is written once (FAST, the orientation and the next level
// read it), the blurred plane is written once, nothing is read back.
//
// Shape (both kernels): lanes are flattened over (frame, 4-pixel column) of the level, as in k_resize_col, and walk DOWN
// a block of R = 6k + 2 rows (+ 6 halo rows, REFLECT_101 at the top and bottom: the row index is wave-uniform, so the
// reflection is scalar).  Per row a lane owns ONE dword of raw pixels; the 7-tap needs the dword left and right of it:
// these come from the neighbouring LANES with two DPP wave shifts (no LDS, no re-load).  Lanes 0 and 63 of every wave
// are halo lanes (they produce their dword but no output): waves overlap by one lane on each side, 62 of 64 lanes are
// productive.  At the first / last column of a frame REFLECT_101 replaces the neighbour (v_perm selectors computed once
// per lane, identity for interior lanes): a frame boundary inside a wave needs no special case.
//   horizontal 7-tap: 10 v_dot4_u32_u8 on the three dwords against taps shifted to the pixel's position (k_blur.hip)
//   vertical 7-tap  : v_dot2_u32_u16 on (row, row+1) pairs kept in a 6-deep register ring; the row loop is unrolled by
//                     six so that the ring is addressed statically
// The arithmetic is the one of k_gray4 / k_resize_col / k_blur2, whose stand-alone forms stay as the generic fallbacks
// (widths that are not a multiple of 4 at level 0, scale factors beyond the 12-byte window, batches beyond 32-bit offsets).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace mslam
{

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) BufRsrc;

__device__ __forceinline__ uint32_t lv_dot2u(uint32_t pair, uint32_t taps, uint32_t acc)
{
    u16x2_t a, b;
    a.x = (unsigned short)(pair & 0xFFFF);
    a.y = (unsigned short)(pair >> 16);
    b.x = (unsigned short)(taps & 0xFFFF);
    b.y = (unsigned short)(taps >> 16);
    return __builtin_amdgcn_udot2(a, b, acc, false);
}

constexpr uint32_t kDropLane = 0xFFFFFFF0u; // vector offset of a lane whose buffer stores are to be dropped (host: batch slabs < 0xFFFF0000 bytes)

// per-lane REFLECT_101 selectors for the window [x0-4, x0+8) held as (L, B, R) = (left neighbour, own, right neighbour)
struct EdgeSel
{
    uint32_t selA, selB, selT, selU, maskT;
};
__device__ __forceinline__ EdgeSel edge_selectors(int x0, int w)
{
    EdgeSel e{0, 0, 0, 0, 0};
#pragma unroll
    for(int i = 0; i < 12; ++i)
    {
        int col = x0 - 4 + i;
        if(col < 0)
            col = -col;
        else if(col >= w)
            col = 2 * (w - 1) - col;
        int s = col - (x0 - 4); // source byte index inside the unreflected window
        if(s < 0 || s > 11)
            s = i; // only feeds outputs that are discarded
        const int b = i & 3;
        if(i < 4)
            e.selA |= (uint32_t)(s <= 7 ? s : i) << (8 * b);
        else if(i < 8)
            e.selB |= (uint32_t)(s <= 7 ? s : i) << (8 * b);
        else if(s < 4)
        {
            e.selT |= (uint32_t)s << (8 * b);
            e.maskT |= 0xFFu << (8 * b);
        }
        else
            e.selU |= (uint32_t)(s - 4) << (8 * b);
    }
    return e;
}

// the blur's register state of one lane: six (row, row+1) pair rows and the previous row's horizontal sums
struct BlurRing
{
    uint32_t pr[6][4]; // pr[p][j]: horizontal sums of rows (i - 1, i) of pixel j as a u16 pair, written when row i with (i + 5) % 6 == p arrives
};

// Feeds raw row i (i % 6 == PH) of the lane's column into the filter.  Returns true and the blurred dword of row i - 3
// (the block's output row i - 6) when EMIT.
template <int PH, bool EMIT>
__device__ __forceinline__ uint32_t blur_feed(BlurRing& st, uint32_t B, const EdgeSel& e, const BlurK& k)
{
    // neighbour dwords: lane l-1's and lane l+1's raw dword of this row (DPP wave shifts; lanes 0 / 63 are halo lanes)
    const uint32_t L = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)B, 0x138, 0xF, 0xF, true); // wave_shr:1
    const uint32_t R = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)B, 0x130, 0xF, 0xF, true); // wave_shl:1
    const uint32_t A2 = __builtin_amdgcn_perm(B, L, e.selA);
    const uint32_t B2 = __builtin_amdgcn_perm(B, L, e.selB);
    const uint32_t T = __builtin_amdgcn_perm(B, L, e.selT);
    const uint32_t U = __builtin_amdgcn_perm(R, B, e.selU);
    uint32_t C2; // (T & maskT) | (U & ~maskT) as one instruction (the compiler builds it from two)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(C2) : "v"(e.maskT), "v"(T), "v"(U));
    uint32_t hv[4];
    hv[0] = __builtin_amdgcn_udot4(A2, k.ta[0], __builtin_amdgcn_udot4(B2, k.tb[0], 0u, false), false);
    hv[1] = __builtin_amdgcn_udot4(A2, k.ta[1], __builtin_amdgcn_udot4(B2, k.tb[1], __builtin_amdgcn_udot4(C2, k.tc[1], 0u, false), false), false);
    hv[2] = __builtin_amdgcn_udot4(A2, k.ta[2], __builtin_amdgcn_udot4(B2, k.tb[2], __builtin_amdgcn_udot4(C2, k.tc[2], 0u, false), false), false);
    hv[3] = __builtin_amdgcn_udot4(B2, k.tb[3], __builtin_amdgcn_udot4(C2, k.tc[3], 0u, false), false);
    // pair (row i-1, row i): row i-1's sum is the high half of the pair written one row ago — v_alignbit takes it from there
    // (no separate copy of the previous row's sums: four registers)
#pragma unroll
    for(int j = 0; j < 4; ++j)
        st.pr[(PH + 5) % 6][j] = __builtin_amdgcn_alignbit(hv[j], st.pr[(PH + 4) % 6][j], 16);
    uint32_t out = 0;
    if(EMIT)
    {
        uint32_t acc[4];
#pragma unroll
        for(int j = 0; j < 4; ++j)
        {
            acc[j] = lv_dot2u(st.pr[PH][j], k.t01, 32768u);           // rows i-6, i-5
            acc[j] = lv_dot2u(st.pr[(PH + 2) % 6][j], k.t23, acc[j]); // rows i-4, i-3
            acc[j] = lv_dot2u(st.pr[(PH + 4) % 6][j], k.t45, acc[j]); // rows i-2, i-1
            acc[j] += __umul24(hv[j], k.t6);                          // row i
        }
        out = __builtin_amdgcn_perm(acc[1], acc[0], 0x0C0C0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020C0Cu);
    }
    return out;
}

// REFLECT_101 of a wave-uniform row index (|y|, then min(y, 2 (h - 1) - y))
__device__ __forceinline__ int reflect_row(int y, int h)
{
    const int ya = y < 0 ? -y : y;
    const int yb = 2 * (h - 1) - ya;
    return ya < yb ? ya : yb;
}

__device__ __forceinline__ uint32_t lv_gray_px(uint32_t c0, uint32_t c1, uint32_t c2)
{
    // (0.299f*c0 + 0.587f*c1) + 0.114f*c2 with every operation individually rounded (no contraction), frame.cpp:13-19
    float v = __fadd_rn(__fmul_rn(0.299f, (float)c0), __fmul_rn(0.587f, (float)c1));
    v = __fadd_rn(v, __fmul_rn(0.114f, (float)c2));
    v = fminf(255.0f, v);
    return (uint32_t)(int)v;
}
__device__ __forceinline__ uint32_t lv_gray4(uint32_t a, uint32_t b, uint32_t c)
{
    // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3   (little endian)
    const uint32_t p0 = lv_gray_px(a & 0xFF, (a >> 8) & 0xFF, (a >> 16) & 0xFF);
    const uint32_t p1 = lv_gray_px(a >> 24, b & 0xFF, (b >> 8) & 0xFF);
    const uint32_t p2 = lv_gray_px((b >> 16) & 0xFF, b >> 24, c & 0xFF);
    const uint32_t p3 = lv_gray_px((c >> 8) & 0xFF, (c >> 16) & 0xFF, c >> 24);
    return p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}

struct Bgr3
{
    uint32_t a, b, c;
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// ---- level 0: gray + blur --------------------------------------------------------------------------------------------
// TILED: the blurred plane is written in kTileW x kTileH pixel tiles (common.hpp: tiled_off; k_describe's patches then
// touch fewer lines).  A lane's dword keeps its place inside the tile row (its vector offset); the row's offset inside the
// plane — tiled_off(pitch, 0, y) — is wave-uniform and rides in the store's scalar offset.
// DEEP = N > 0: the launch is a handful of waves (the synchronous single-frame call) and lasts as long as ONE wave's walk, so
// all N = R + 6 rows of the block are requested before the first is used (N x 3 registers) instead of two rows ahead: one
// memory round trip per block instead of one per two rows.
// the walk of ONE wave: wave gw of the launch's (frame, quad) items, row block `by`
template <bool TILED, int DEEP>
__device__ __forceinline__ void gray_blur_wave(const GrayBlurArgs& a, const int gw, const int by)
{
    const int lane = threadIdx.x & 63;
    const int n_items = a.n_frames * a.quads;
    const int item_raw = gw * 62 - 1 + lane;
    if(gw * 62 >= n_items)
        return; // whole wave beyond the last item (wave-uniform)
    const bool productive = lane >= 1 && lane <= 62 && item_raw < n_items;
    const int item = max(0, min(item_raw, n_items - 1));
    const int f = (int)(((float)item + 0.5f) * a.inv_quads); // item / quads, exact for items < 2^22 (host check)
    const int q = item - f * a.quads;
    const int R = 6 * a.k6 + 2;
    const int y0 = min(by * R, a.H - R); // the last block ends at the last row (host: R <= H)
    const EdgeSel e = edge_selectors(4 * q, a.W);
    const uint32_t src_v = (uint32_t)(f + a.frame0) * (uint32_t)(a.W * a.H * 3) + 12u * (uint32_t)q;
    // Every lane stores every row, unconditionally: the compiler can then count the stores in its vmcnt waits and the
    // loads stay two rows ahead (a store that may or may not issue makes every counted wait stricter).  The stores are
    // buffer stores: the lane's column offset rides in the vector operand, the wave-uniform row offset in the scalar one (no
    // vector address arithmetic per row), and what must not be written is dropped by the buffer's range check — a halo lane
    // carries an offset beyond the buffer (kDropLane), a halo row is stored against an empty buffer (0 records).  Neither
    // reaches the cache.  (Round 3 first wrote those to dump words behind the slab: one more write request per store.)
    const uint32_t col_v = (uint32_t)(f + a.frame0) * a.slab + 4u * (uint32_t)q;
    const uint32_t raw_v = productive ? col_v : kDropLane;
    const uint32_t blur_v = !productive ? kDropLane : TILED ? (uint32_t)(f + a.frame0) * a.slab + tiled_off((unsigned)a.pitch, 4 * q, 0) : col_v;
    const uint32_t n_rec = (uint32_t)(a.frame0 + a.n_frames) * a.slab;
    const BufRsrc blur_rs = __builtin_amdgcn_make_buffer_rsrc(a.blur, 0, (int)n_rec, 0x00020000);
    const uint32_t row_bytes = (uint32_t)a.W * 3u;

    // buffer loads: the wave-uniform row offset rides in the scalar offset operand, the lane's offset in the vector one —
    // no 64-bit vector address arithmetic per load (the compiler built a v_mad_u64_u32 per row for `row pointer + lane offset`)
    const BufRsrc bgr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.bgr), 0, -1, 0x00020000);
    auto load = [&](int i) {
        const int y = reflect_row(y0 - 3 + i, a.H);
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(bgr_rs, (int)src_v, (int)((uint32_t)y * row_bytes), 0);
        return Bgr3{v.x, v.y, v.z};
    };
    BlurRing st;
#pragma unroll
    for(int j = 0; j < 4; ++j)
        st.pr[4][j] = 0; // (row 0 pairs with "row -1": never looked at, but defined)
    // one row: i = i0 + PH, PH = i % 6 (static); RAW: the row belongs to the block (rows 3 .. R+2), BLUR: i >= 6
    auto compute = [&](auto ph, auto emit, auto sraw, int i, bool raw, const Bgr3& w) {
        constexpr int PH = decltype(ph)::value;
        constexpr bool EMIT = decltype(emit)::value;
        constexpr bool SRAW = decltype(sraw)::value; // false for block rows 0 .. 2: no raw store is emitted at all
        const uint32_t g = lv_gray4(w.a, w.b, w.c);
        if(SRAW) // rows 3 .. R+2 need no reflection; a halo row is stored against an empty buffer
            __builtin_amdgcn_raw_buffer_store_b32(g, __builtin_amdgcn_make_buffer_rsrc(a.pyr, 0, __builtin_amdgcn_readfirstlane(raw ? (int)n_rec : 0), 0x00020000),
                                                  (int)raw_v, (y0 - 3 + i) * a.pitch, 0);
        const uint32_t o = blur_feed<PH, EMIT>(st, g, e, a.bk);
        if(EMIT)
            __builtin_amdgcn_raw_buffer_store_b32(o, blur_rs, (int)blur_v,
                                                  (int)(TILED ? tiled_off((unsigned)a.pitch, 0, y0 - 6 + i) : (unsigned)((y0 - 6 + i) * a.pitch)), 0);
    };
    using std::integral_constant;
    if constexpr(DEEP > 0)
    {
        // (host: R + 6 == DEEP) every row of the block in flight at once, then the rows in order
        Bgr3 all[DEEP];
#pragma unroll
        for(int i = 0; i < DEEP; ++i)
            all[i] = load(i);
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would otherwise sink every load to its use)
        static_for([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr bool RAWROW = I >= 3 && I <= DEEP - 4; // rows 3 .. R + 2
            compute(integral_constant<int, I % 6>{}, integral_constant<bool, (I >= 6)>{}, integral_constant<bool, RAWROW>{}, I, RAWROW, all[I]);
        }, std::make_integer_sequence<int, DEEP>{});
        return;
    }
    Bgr3 ring[3];
    ring[0] = load(0);
    ring[1] = load(1);
    auto row = [&](auto ph, auto emit, auto sraw, int i, bool raw) {
        constexpr int PH = decltype(ph)::value;
        ring[(PH + 2) % 3] = load(min(i + 2, R + 5));
        compute(ph, emit, sraw, i, raw, ring[PH % 3]);
    };
#define MSLAM_ROW(PH, EMIT, SRAW, I, RAW) row(integral_constant<int, PH>{}, integral_constant<bool, EMIT>{}, integral_constant<bool, SRAW>{}, I, RAW)
    // rows 0 .. 5: fill the ring (rows 3, 4, 5 are the block's first rows)
    MSLAM_ROW(0, false, false, 0, false);
    MSLAM_ROW(1, false, false, 1, false);
    MSLAM_ROW(2, false, false, 2, false);
    MSLAM_ROW(3, false, true, 3, true);
    MSLAM_ROW(4, false, true, 4, true);
    MSLAM_ROW(5, false, true, 5, R > 2); // (a 2-row block — k6 = 0, the single-frame launches — ends at row 4: row 5 is a halo row)
#pragma unroll 1
    for(int i0 = 6; i0 < R + 4; i0 += 6)
    {
        MSLAM_ROW(0, true, true, i0, true);
        MSLAM_ROW(1, true, true, i0 + 1, true);
        MSLAM_ROW(2, true, true, i0 + 2, true);
        MSLAM_ROW(3, true, true, i0 + 3, true);
        MSLAM_ROW(4, true, true, i0 + 4, true);
        MSLAM_ROW(5, true, true, i0 + 5, i0 + 5 < R + 3); // the block's last raw row is R + 2: in the last trip this one is a halo row
    }
    // rows R+4, R+5 (R + 4 = 6 (k6 + 1)): halo rows below the block
    MSLAM_ROW(0, true, false, R + 4, false);
    MSLAM_ROW(1, true, false, R + 5, false);
#undef MSLAM_ROW
}

template <bool TILED, int DEEP>
__global__ __launch_bounds__(256) void k_gray_blur(GrayBlurArgs a)
{
    // XCD-aware wave numbering: workgroups go to the 8 XCDs round-robin by linear id (gridDim.x is a multiple of 8, so the
    // XCD of a workgroup is blockIdx.x & 7 for every row block).  Each XCD gets a contiguous eighth of the waves: waves that
    // are neighbours in the image (shared 128-byte lines at their edges, shared halo rows between row blocks) meet in one L2.
    const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x & 7) * a.waves_per_xcd + (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)));
    gray_blur_wave<TILED, DEEP>(a, gw, (int)blockIdx.y);
}

// ---- level l > 0: resize + blur --------------------------------------------------------------------------------------
// The interpolation is k_resize_col's (host tables: per destination quad a 12-byte source window, v_perm selectors and
// weight pairs; per destination row the source row and the weight pair).  Rows are walked in DESTINATION order here (the
// blur ring wants a static row phase), so the two horizontally interpolated source rows a destination row blends live in
// (hA, hB) with their wave-uniform row numbers: a row re-uses them, advances by one source row, or reloads both (scale
// factors up to 2, and the reflected rows at the top / bottom of the level, which walk backwards).  The source rows of
// destination row i+1 are requested while row i is computed (PA / PB).
struct Raw3
{
    uint32_t d0, d1, d2;
};
struct HRow
{
    uint32_t h[4];
};

// NEED: bit k set = pixel k of a quad may take its (S[x], S[x+1]) pair from window dwords (1,2) instead of (0,1) — a
// compile-time superset of the level's mask (at scale 1.2 only the fourth pixel ever does), so that the other pixels carry
// no per-lane selects
// DEEP = N > 0: as in k_gray_blur — both source-row windows of all N = R + 6 rows of the block are requested up front (the
// single-frame launches, whose duration is one wave's walk: 6 N registers)
// the walk of ONE wave: wave gw of the launch's (frame, quad) items, row block `by`.  COH: the source level was written by
// this very kernel (k_level_chain): its loads go past the CU's L1 (sc0).
template <bool EXACT, int NEED, bool TILED, int DEEP, bool COH>
__device__ __forceinline__ void resize_blur_wave(const ResizeBlurArgs& a, const int gw, const int by)
{
    const int lane = threadIdx.x & 63;
    const int n_items = a.n_frames * a.quads;
    const int R = 6 * a.k6 + 2; // host: R + 6 <= 64 (the block's row table lives in lane registers), R <= dh
    const int y0 = min(by * R, a.dh - R);
    // lane i keeps the table entry of the block's row i (destination row reflect(y0 - 3 + i)); loaded by ALL lanes before
    // any leaves (v_readlane reads lanes that have exited, too)
    const int my_row = reflect_row(y0 - 3 + min(lane, R + 5), a.dh);
    const int my_y0 = a.yofs[my_row];
    const uint32_t my_yc = a.ycoef[my_row];
    if(gw * 62 >= n_items)
        return; // whole wave beyond the last item (wave-uniform)
    const int item_raw = gw * 62 - 1 + lane;
    const bool productive = lane >= 1 && lane <= 62 && item_raw < n_items;
    const int item = max(0, min(item_raw, n_items - 1));
    const int f = (int)(((float)item + 0.5f) * a.inv_quads); // item / quads, exact for items < 2^22 (host check)
    const int qx = item - f * a.quads;
    const EdgeSel e = edge_selectors(4 * qx, a.dw);
    const uint4 t0 = a.qt[3 * qx], t1 = a.qt[3 * qx + 1], t2 = a.qt[3 * qx + 2];
    const uint32_t sel[4] = {t0.z, t0.w, t1.x, t1.y}, coef[4] = {t1.z, t1.w, t2.x, t2.y};
    const uint32_t src_v = (uint32_t)(f + a.frame0) * a.slab + t0.x;
    // unconditional buffer stores, halo lanes / halo rows dropped by the range check: see k_gray_blur
    const uint32_t col_v = (uint32_t)(f + a.frame0) * a.slab + (uint32_t)a.dst_off + 4u * (uint32_t)qx;
    const uint32_t raw_v = productive ? col_v : kDropLane;
    const uint32_t blur_v = !productive ? kDropLane
                            : TILED   ? (uint32_t)(f + a.frame0) * a.slab + (uint32_t)a.dst_off + tiled_off((unsigned)a.dpitch, 4 * qx, 0)
                                      : col_v;
    const uint32_t n_rec = (uint32_t)(a.frame0 + a.n_frames) * a.slab;
    const BufRsrc blur_rs = __builtin_amdgcn_make_buffer_rsrc(a.blur, 0, (int)n_rec, 0x00020000);
    const uint8_t* src_lv = a.pyr + a.src_off;

    const BufRsrc src_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_lv), 0, -1, 0x00020000);
    auto load = [&](int sy) { // (buffer load: scalar row offset + vector lane offset, see k_gray_blur)
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(src_rs, (int)src_v, (int)((uint32_t)sy * (uint32_t)a.spitch), COH ? 1 : 0);
        return Raw3{v.x, v.y, v.z};
    };
    auto load_if = [&](int sy, bool need) {
        // (readfirstlane: the record count must be a scalar register for the compiler, else it wraps the load in a waterfall loop)
        const BufRsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src_lv), 0, __builtin_amdgcn_readfirstlane(need ? -1 : 0), 0x00020000);
        const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)src_v, (int)((uint32_t)sy * (uint32_t)a.spitch), COH ? 1 : 0);
        return Raw3{v.x, v.y, v.z};
    };
    auto hinterp = [&](const Raw3& w) {
        HRow r;
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            uint32_t lo = w.d0, hi = w.d1;
            if(NEED & (1 << k))
            {
                const bool up = (t0.y >> k) & 1u;
                lo = up ? w.d1 : w.d0;
                hi = up ? w.d2 : w.d1;
            }
            const uint32_t pr = __builtin_amdgcn_perm(hi, lo, sel[k]);
            u16x2_t pv, cv;
            pv.x = (unsigned short)(pr & 0xFFFF);
            pv.y = (unsigned short)(pr >> 16);
            cv.x = (unsigned short)(coef[k] & 0xFFFF);
            cv.y = (unsigned short)(coef[k] >> 16);
            r.h[k] = EXACT ? __builtin_amdgcn_udot2(pv, cv, 0u, false) : __builtin_amdgcn_udot2(pv, cv, 0u, false) >> 4;
        }
        return r;
    };
    // source rows of the block's row i (resizeGeneric_Invoker clips the row index, not the weight)
    auto rows_of = [&](int i, int& sy0, int& sy1) {
        const int t = __builtin_amdgcn_readlane(my_y0, i);
        sy0 = max(0, min(t, a.sh - 1));
        sy1 = max(0, min(t + 1, a.sh - 1));
    };

    BlurRing st;
#pragma unroll
    for(int j = 0; j < 4; ++j)
        st.pr[4][j] = 0; // (row 0 pairs with "row -1": never looked at, but defined)
    HRow hA{}, hB{};
    int rowA = -1, rowB = -1; // wave-uniform: which source rows hA / hB hold
    // both source rows of every destination row are requested two rows ahead, unconditionally (static 3-deep ring: the
    // compiler can count its vmcnt waits); whether a row is interpolated again or taken over from hB is decided at use
    auto compute = [&](auto ph, auto emit, auto sraw, int i, bool raw, int sy0, int sy1, const Raw3& wA, const Raw3& wB) {
        constexpr int PH = decltype(ph)::value;
        constexpr bool EMIT = decltype(emit)::value;
        constexpr bool SRAW = decltype(sraw)::value; // false for block rows 0 .. 2: no raw store is emitted at all
        // bring (hA, hB) to (sy0, sy1).  All conditions are wave-uniform; the empty asm statements keep the compiler from
        // turning the branches into speculated work + selects.
        const bool reuse = sy1 == rowB && (sy0 == rowA || sy0 == rowB);
        if(!reuse)
        {
            if(sy0 == rowB)
            {
                asm volatile("");
                hA = hB; // one source row further (the common step at scale factors below 2)
            }
            else
            {
                asm volatile("");
                hA = hinterp(wA);
            }
            rowA = sy0;
            if(sy1 == sy0)
            {
                asm volatile("");
                hB = hA; // both clipped to the first / last source row
            }
            else
            {
                asm volatile("");
                hB = hinterp(wB);
            }
            rowB = sy1;
        }
        const uint32_t yc = (uint32_t)__builtin_amdgcn_readlane((int)my_yc, i);
        const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
        auto blend = [&](const HRow& h0, const HRow& h1) {
            uint32_t r = 0;
            if(EXACT)
            {
                uint32_t acc[4];
#pragma unroll
                for(int k = 0; k < 4; ++k) // h <= 65280, weights <= 256: the 24-bit multiply-adds are exact
                    acc[k] = __umul24(b1, h1.h[k]) + (__umul24(b0, h0.h[k]) + 32768u);
                r = __builtin_amdgcn_perm(acc[1], acc[0], 0x0C0C0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020C0Cu);
            }
            else
            {
                const uint32_t b0s = b0 << 16, b1s = b1 << 16; // (b * h) >> 16 as the high half of (b << 16) * h
                uint32_t v[4];
#pragma unroll
                for(int k = 0; k < 4; ++k)
                    v[k] = __umulhi(h0.h[k], b0s) + __umulhi(h1.h[k], b1s) + 2u; // <= 1022; the pixel is v >> 2
                // two sums per register, ONE shift per pair (a pair's low bits fall into bits 14, 15 of the other's field,
                // which no byte of the result is taken from), one byte gather
                const uint32_t p01 = (v[0] | (v[1] << 16)) >> 2, p23 = (v[2] | (v[3] << 16)) >> 2;
                r = __builtin_amdgcn_perm(p23, p01, 0x06040200u);
            }
            return r;
        };
        uint32_t g;
        if(sy0 == rowA)
        {
            asm volatile("");
            g = blend(hA, hB);
        }
        else // a re-used pair whose lower row is hB as well (up-scaling only)
        {
            asm volatile("");
            g = blend(hB, hB);
        }
        if(SRAW) // rows 3 .. R+2 need no reflection; a halo row is stored against an empty buffer (see k_gray_blur)
            __builtin_amdgcn_raw_buffer_store_b32(g, __builtin_amdgcn_make_buffer_rsrc(a.pyr, 0, __builtin_amdgcn_readfirstlane(raw ? (int)n_rec : 0), 0x00020000),
                                                  (int)raw_v, (y0 - 3 + i) * a.dpitch, 0);
        const uint32_t o = blur_feed<PH, EMIT>(st, g, e, a.bk);
        if(EMIT)
            __builtin_amdgcn_raw_buffer_store_b32(o, blur_rs, (int)blur_v,
                                                  (int)(TILED ? tiled_off((unsigned)a.dpitch, 0, y0 - 6 + i) : (unsigned)((y0 - 6 + i) * a.dpitch)), 0);
    };
    using std::integral_constant;
    if constexpr(DEEP > 0)
    {
        // (host: R + 6 == DEEP) both windows of every row of the block in flight at once, then the rows in order
        Raw3 allA[DEEP], allB[DEEP];
#pragma unroll
        for(int i = 0; i < DEEP; ++i)
        {
            int y0i, y1i;
            rows_of(i, y0i, y1i);
            allA[i] = load(y0i);
            allB[i] = load(y1i);
        }
        __builtin_amdgcn_sched_barrier(0); // (the scheduler would otherwise sink every load to its use)
        static_for([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr bool RAWROW = I >= 3 && I <= DEEP - 4; // rows 3 .. R + 2
            int sy0, sy1;
            rows_of(I, sy0, sy1);
            compute(integral_constant<int, I % 6>{}, integral_constant<bool, (I >= 6)>{}, integral_constant<bool, RAWROW>{}, I, RAWROW, sy0, sy1,
                    allA[I], allB[I]);
        }, std::make_integer_sequence<int, DEEP>{});
        return;
    }
    Raw3 RA[3], RB[3];
    // the source rows of block rows i and i + 1 are carried from the iterations that requested them (one table lookup per row)
    int c0_y0, c0_y1, c1_y0, c1_y1;
    rows_of(0, c0_y0, c0_y1);
    rows_of(1, c1_y0, c1_y1);
    RA[0] = load(c0_y0);
    RB[0] = load(c0_y1);
    RA[1] = load(c1_y0);
    RB[1] = load(c1_y1);
    auto row = [&](auto ph, auto emit, auto sraw, int i, bool raw) {
        constexpr int PH = decltype(ph)::value;
        const int sy0 = c0_y0, sy1 = c0_y1;
        {
            int ny0, ny1;
            rows_of(min(i + 2, R + 5), ny0, ny1);
            // hA of row i + 2 is taken over from hB when its upper source row is the lower one of row i + 1 (the common step at
            // scale factors below 2): that row's window is then never looked at — the load still issues (the compiler counts
            // its vmcnt waits statically) but against an empty buffer, which returns zeros without touching the cache
            // (0.72 -> 0.68 ms per 1000 frames; an empty-buffer load takes longer to return than a cache hit, so the handful-of-
            // frames launches, which run at the latency of one wave's walk, keep the real load: always_load)
            RA[(PH + 2) % 3] = load_if(ny0, ny0 != c1_y1 || a.always_load);
            RB[(PH + 2) % 3] = load(ny1);
            c0_y0 = c1_y0, c0_y1 = c1_y1;
            c1_y0 = ny0, c1_y1 = ny1;
        }
        compute(ph, emit, sraw, i, raw, sy0, sy1, RA[PH % 3], RB[PH % 3]);
    };
#define MSLAM_ROW(PH, EMIT, SRAW, I, RAW) row(integral_constant<int, PH>{}, integral_constant<bool, EMIT>{}, integral_constant<bool, SRAW>{}, I, RAW)
    MSLAM_ROW(0, false, false, 0, false);
    MSLAM_ROW(1, false, false, 1, false);
    MSLAM_ROW(2, false, false, 2, false);
    MSLAM_ROW(3, false, true, 3, true);
    MSLAM_ROW(4, false, true, 4, true);
    MSLAM_ROW(5, false, true, 5, R > 2); // (a 2-row block — k6 = 0, the single-frame launches — ends at row 4: row 5 is a halo row)
#pragma unroll 1
    for(int i0 = 6; i0 < R + 4; i0 += 6)
    {
        MSLAM_ROW(0, true, true, i0, true);
        MSLAM_ROW(1, true, true, i0 + 1, true);
        MSLAM_ROW(2, true, true, i0 + 2, true);
        MSLAM_ROW(3, true, true, i0 + 3, true);
        MSLAM_ROW(4, true, true, i0 + 4, true);
        MSLAM_ROW(5, true, true, i0 + 5, i0 + 5 < R + 3); // the block's last raw row is R + 2: in the last trip this one is a halo row
    }
    MSLAM_ROW(0, true, false, R + 4, false);
    MSLAM_ROW(1, true, false, R + 5, false);
#undef MSLAM_ROW
}

template <bool EXACT, int NEED, bool TILED, int DEEP>
__global__ __launch_bounds__(256) void k_resize_blur(ResizeBlurArgs a)
{
    // XCD-aware wave numbering, as in k_gray_blur
    const int gw = __builtin_amdgcn_readfirstlane((int)((blockIdx.x & 7) * a.waves_per_xcd + (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)));
    resize_blur_wave<EXACT, NEED, TILED, DEEP, false>(a, gw, (int)blockIdx.y);
}

// ---- the whole level chain of a group of frames in ONE launch (round 6) ------------------------------------------------
// The per-level launches above are each a single, partly filled generation of waves: a launch lasts as long as one wave's
// walk down its row block and the next level cannot start before the last wave of this one has drained (14 launches per
// 1000-frame step).  Here a small workgroup takes G frames through gray + blur and every resize + blur level in sequence:
// its waves deal the (column wave, row block) walks of a level among themselves, meet at a barrier (+ a workgroup-scope
// fence: the level just written is read back through the L2, writer and reader share the CU) and go on to the next level.
// Workgroups are independent, so a CU holds frames at different levels at the same time (the memory-bound level 0 of one
// beside the issue-bound upper levels of another) and the launch has a steady state.  The walks are the ones above.
template <bool EXACT, int NEED, bool TILED, int WAVES, bool COH>
__global__ __launch_bounds__(WAVES * 64) void k_level_chain(LevelChainArgs ch)
{
    // groups contiguous per XCD (workgroups go to the XCDs round-robin by linear id)
    const int grp = (int)(blockIdx.x & 7) * ch.groups_per_xcd + (int)(blockIdx.x >> 3);
    if(grp >= ch.n_groups)
        return; // (whole workgroup: before any barrier)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int f0 = grp * ch.G;
    const int nf = min(ch.G, ch.n_frames - f0);
    {
        GrayBlurArgs a = ch.g;
        a.frame0 = ch.frame0 + f0;
        a.n_frames = nf;
        const int R = 6 * a.k6 + 2;
        const int nwx = (nf * a.quads + 61) / 62, nby = (a.H + R - 1) / R;
#pragma unroll 1
        for(int item = wave; item < nwx * nby; item += WAVES)
        {
            const int by = item / nwx;
            gray_blur_wave<TILED, 0>(a, item - by * nwx, by);
        }
    }
#pragma unroll 1
    for(int l = 0; l < ch.n_lv; ++l)
    {
        // workgroup scope is enough — writer and reader waves share the CU: the stores are acknowledged by the L2 (vmcnt) before
        // the barrier, the next level's loads go past the L1 when COH (a device-scope fence would write the whole L2 back)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        ResizeBlurArgs a = ch.lv[l];
        a.frame0 = ch.frame0 + f0;
        a.n_frames = nf;
        const int R = 6 * a.k6 + 2;
        const int nwx = (nf * a.quads + 61) / 62, nby = (a.dh + R - 1) / R;
#pragma unroll 1
        for(int item = wave; item < nwx * nby; item += WAVES)
        {
            const int by = item / nwx;
            resize_blur_wave<EXACT, NEED, TILED, 0, COH>(a, item - by * nwx, by);
        }
    }
}

// rows per block of a chained level: the block height 6 k + 2 that minimises the longest wave's walk, (walks per wave) x (R + 6)
static int chain_k6(int quads, int rows, int G, int waves, int k6_max)
{
    const long nwx = ((long)G * quads + 61) / 62;
    int best = 1;
    long best_cost = -1;
    for(int k = 1; k <= k6_max && 6 * k + 2 <= rows; ++k)
    {
        const int R = 6 * k + 2;
        const long items = nwx * ((rows + R - 1) / R);
        const long cost = ((items + waves - 1) / waves) * (R + 6);
        if(best_cost < 0 || cost <= best_cost)
            best = k, best_cost = cost;
    }
    return best;
}

bool launch_level_chain(const GrayBlurArgs& g, const ResizeBlurArgs* lv, int n_lv, int frames_per_group, int waves, int k6_max, hipStream_t s)
{
    if(n_lv > kChainLevels - 1 || !g.blur_tiled)
        return false;
    LevelChainArgs ch{};
    ch.g = g;
    ch.G = frames_per_group;
    ch.frame0 = g.frame0;
    ch.n_frames = g.n_frames;
    ch.n_lv = n_lv;
    ch.g.k6 = chain_k6(g.quads, g.H, ch.G, waves, k6_max);
    int need = 0;
    for(int l = 0; l < n_lv; ++l)
    {
        if(lv[l].exact || !lv[l].blur_tiled)
            return false;
        ch.lv[l] = lv[l];
        ch.lv[l].always_load = 0;
        ch.lv[l].k6 = chain_k6(lv[l].quads, lv[l].dh, ch.G, waves, k6_max);
        need |= lv[l].need_mask;
    }
    ch.n_groups = (g.n_frames + ch.G - 1) / ch.G;
    ch.groups_per_xcd = (ch.n_groups + 7) / 8;
    const dim3 grid(8 * ch.groups_per_xcd);
    const bool n8 = (need & ~8) == 0;
#define MSLAM_LC(N, W) hipLaunchKernelGGL((k_level_chain<false, N, true, W, true>), grid, dim3(W * 64), 0, s, ch)
    if(waves == 8)
    {
        if(n8) MSLAM_LC(8, 8); else MSLAM_LC(15, 8);
    }
    else
    {
        if(n8) MSLAM_LC(8, 4); else MSLAM_LC(15, 4);
    }
#undef MSLAM_LC
    return true;
}

void launch_resize_blur(const ResizeBlurArgs& a, hipStream_t s)
{
    const int R = 6 * a.k6 + 2;
    const int n_waves = (a.n_frames * a.quads + 61) / 62;
    ResizeBlurArgs b = a;
    b.waves_per_xcd = ((n_waves + 31) / 32) * 4; // whole workgroups per XCD
    dim3 grid(8 * (b.waves_per_xcd / 4), (a.dh + R - 1) / R);
    const int need = (a.need_mask & ~8) == 0 ? (a.need_mask ? 8 : 0) : (a.need_mask & ~12) == 0 ? 12 : 15;
    // a handful of frames (always_load) with 2- or 8-row blocks: the deep-prefetch instances (MSLAM_HIP_LEVEL_DEEP=0: off)
    static const bool deep_env = [] { const char* e = getenv("MSLAM_HIP_LEVEL_DEEP"); return !e || atoi(e) != 0; }();
    const int deep = (a.always_load && deep_env && a.k6 <= 1) ? R + 6 : 0;
#define MSLAM_RB3(E, N, T) do { if(deep == 8) hipLaunchKernelGGL((k_resize_blur<E, N, T, 8>), grid, dim3(256), 0, s, b); \
                               else if(deep == 14) hipLaunchKernelGGL((k_resize_blur<E, N, T, 14>), grid, dim3(256), 0, s, b); \
                               else hipLaunchKernelGGL((k_resize_blur<E, N, T, 0>), grid, dim3(256), 0, s, b); } while(0)
#define MSLAM_RB(E, N) do { if(a.blur_tiled) MSLAM_RB3(E, N, true); else MSLAM_RB3(E, N, false); } while(0)
    if(a.exact)
    {
        if(need == 0) MSLAM_RB(true, 0); else if(need == 8) MSLAM_RB(true, 8); else if(need == 12) MSLAM_RB(true, 12); else MSLAM_RB(true, 15);
    }
    else
    {
        if(need == 0) MSLAM_RB(false, 0); else if(need == 8) MSLAM_RB(false, 8); else if(need == 12) MSLAM_RB(false, 12); else MSLAM_RB(false, 15);
    }
#undef MSLAM_RB3
#undef MSLAM_RB
}

void launch_gray_blur(const GrayBlurArgs& a, hipStream_t s)
{
    const int R = 6 * a.k6 + 2;
    const int n_waves = (a.n_frames * a.quads + 61) / 62;
    GrayBlurArgs b = a;
    b.waves_per_xcd = ((n_waves + 31) / 32) * 4; // whole workgroups per XCD
    dim3 grid(8 * (b.waves_per_xcd / 4), (a.H + R - 1) / R);
    static const bool deep_env = [] { const char* e = getenv("MSLAM_HIP_LEVEL_DEEP"); return !e || atoi(e) != 0; }();
    const int deep = (a.n_frames < 8 && deep_env && a.k6 <= 1) ? R + 6 : 0;
#define MSLAM_GB(T) do { if(deep == 8) hipLaunchKernelGGL((k_gray_blur<T, 8>), grid, dim3(256), 0, s, b); \
                        else if(deep == 14) hipLaunchKernelGGL((k_gray_blur<T, 14>), grid, dim3(256), 0, s, b); \
                        else hipLaunchKernelGGL((k_gray_blur<T, 0>), grid, dim3(256), 0, s, b); } while(0)
    if(a.blur_tiled)
        MSLAM_GB(true);
    else
        MSLAM_GB(false);
#undef MSLAM_GB
}

} // namespace mslam
