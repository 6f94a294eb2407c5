// k_cvorb.hip — the kernels only the cv::ORB detector mode needs (MSLAM_HIP_DETECTOR_CV_ORB).
//
// Replaces OrbOpenCvDetector::Pimpl::detect (reference orb_feature.cpp:33-65): toGrayScale, then
// cv::ORB::create(1000)->detectAndCompute (orb_feature.cpp:25,40).  Everything below that call is OpenCV 4.8.1
// (features2d/src/orb.cpp; not in the reference tree), restated stage by stage:
//   pyramid      resize(prev, sz, INTER_LINEAR_EXACT), chained          -> k_resize_exact
//   FAST         FastFeatureDetector(20, nonmax = true) on the WHOLE level -> k_fast_tiles (64x64 tiles staged in LDS:
//                scores incl. a 1-px ring, 3x3 strict NMS, runByImageBorder(31), per-level list)
//   selection    retainBest(2 n_l) by FAST score, HarrisResponses (7x7 block, k = 0.04), retainBest(n_l) by Harris
//                                                                       -> k_cv_select (one workgroup per level; first
//                puts the list into FAST's raster order)
//   orientation, blur, rBRIEF: k_blur and k_describe (shared with the in-tree detector; k_describe takes its
//                cos/sin from include/mslam_sincos.h in this mode and the Harris response as the keypoint response)
// retainBest keeps the SET {response >= n-th largest} (ties kept); its ORDER in the reference comes from
// std::nth_element / std::partition and is implementation-defined, so lists stay in FAST's raster order here.
#include "common.hpp"
#include "arc_score.hpp"

namespace mslam
{

// ---- INTER_LINEAR_EXACT (imgproc resize.cpp: resize_bitExact, interpolationLinear<uchar>, ufixedpoint16) ------
// horizontal: h = c0 * S[o] + c1 * S[o+1] in 8.8 (u16, exact); left of xmin / right of xmax: edge sample << 8.
// vertical  : (h0 * b0 + h1 * b1 + 2^15) >> 16; above ymin / below ymax: (h + 128) >> 8 of the first / last row.
// The host tables fold the edge rules in: a position left of xmin is (offset 0, coefficients 256 / 0), right of xmax
// (offset sw-2, 0 / 256) — the same 8.8 value as "edge sample << 8" — and a row above ymin / below ymax likewise, where
// (h * 256 + 2^15) >> 16 == (h + 128) >> 8.  The kernel is therefore branch-free: 4 pixels per lane, all loads first.
// The kernel is branch-free: 4 destination pixels and R = 4 destination rows per lane.  The 4 pixels draw on at most
// 12 consecutive source bytes (scale <= 2), fetched per source row as three aligned dwords; the pair (S[o], S[o+1])
// of pixel k is cut out with one v_perm_b32 whose selector depends on (o_k - window base) only, i.e. is computed once
// per lane, as are the horizontal coefficients.
template <bool WINDOW>
__global__ __launch_bounds__(256) void k_resize_exact(ExactResizeArgs a)
{
    constexpr int R = 4;
    const int qx = blockIdx.x * 64 + threadIdx.x;
    const int dy0 = (blockIdx.y * 4 + threadIdx.y) * R;
    if(dy0 >= a.dh || (qx << 2) >= a.dw)
        return;
    const size_t frame = blockIdx.z + a.frame0;
    const uint8_t* src = a.pyr + frame * a.slab + a.src_off;
    uint8_t* dst = a.pyr + frame * a.slab + a.dst_off;
    int o[4];
    uint32_t c0[4], c1[4], sel[4];
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        const int dx = min((qx << 2) + k, a.dw - 1);
        o[k] = a.xofs[dx];
        const uint32_t xc = a.xcoef[dx];
        c0[k] = xc & 0xFFFF, c1[k] = xc >> 16;
    }
    const int base = o[0] & ~3;
#pragma unroll
    for(int k = 0; k < 4; ++k)
    {
        // bytes (s, s+1) of the 12-byte window -> u16 lanes [S[o], 0, S[o+1], 0]: from dwords (0,1) when s <= 6, else (1,2)
        const uint32_t sft = (uint32_t)(o[k] - base);
        sel[k] = 0x0c010c00u + (sft > 6 ? sft - 4 : sft) * 0x00010001u;
    }
    for(int r = 0; r < R; ++r)
    {
        const int dy = dy0 + r;
        if(dy >= a.dh)
            break;
        const int r0 = a.yofs[dy];
        const uint32_t yc = a.ycoef[dy];
        const uint32_t b0 = yc & 0xFFFF, b1 = yc >> 16;
        const uint8_t* S0 = src + (size_t)r0 * a.spitch;
        const uint8_t* S1 = S0 + a.spitch;
        uint32_t h0[4], h1[4];
        if(WINDOW)
        {
            const uint32_t* W0 = reinterpret_cast<const uint32_t*>(S0 + base);
            const uint32_t* W1 = reinterpret_cast<const uint32_t*>(S1 + base);
            const uint32_t a0 = W0[0], a1 = W0[1], a2 = W0[2], e0 = W1[0], e1 = W1[1], e2 = W1[2];
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                const bool up = (uint32_t)(o[k] - base) > 6;
                const uint32_t pa = __builtin_amdgcn_perm(up ? a2 : a1, up ? a1 : a0, sel[k]); // S0[o] | S0[o+1] << 16
                const uint32_t pe = __builtin_amdgcn_perm(up ? e2 : e1, up ? e1 : e0, sel[k]);
                // every factor is below 2^16 and every product below 2^24: full-rate 24-bit multiplies
                h0[k] = __umul24(c0[k], pa & 0xFFFF) + __umul24(c1[k], pa >> 16);
                h1[k] = __umul24(c0[k], pe & 0xFFFF) + __umul24(c1[k], pe >> 16);
            }
        }
        else
        {
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                h0[k] = __umul24(c0[k], S0[o[k]]) + __umul24(c1[k], S0[o[k] + 1]);
                h1[k] = __umul24(c0[k], S1[o[k]]) + __umul24(c1[k], S1[o[k] + 1]);
            }
        }
        uint32_t out = 0;
#pragma unroll
        for(int k = 0; k < 4; ++k)
            out |= (((__umul24(h0[k], b0) + __umul24(h1[k], b1) + 32768u) >> 16) & 0xFFu) << (8 * k);
        *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dpitch + (qx << 2)) = out;
    }
}

void launch_resize_exact(const ExactResizeArgs& a, int n_frames, hipStream_t s)
{
    dim3 grid(((a.dw + 3) / 4 + 63) / 64, (a.dh + 15) / 16, n_frames);
    if(a.window12) // every quad's source pixels fit the aligned 12-byte window (checked on the host)
        hipLaunchKernelGGL(k_resize_exact<true>, grid, dim3(64, 4), 0, s, a);
    else
        hipLaunchKernelGGL(k_resize_exact<false>, grid, dim3(64, 4), 0, s, a);
}

// ---- whole-level FAST-9/16 with 3x3 non-max suppression, per 64x64 tile -------------------------------------------
// score(x, y) = cornerScore<16> when the pixel passes the 9-contiguous test at `thr`, else 0, for 3 <= x < w-3,
// 3 <= y < h-3 (FAST_t's tested range); a keypoint is a pixel whose score is strictly greater than its 8 neighbours'.
// One workgroup per 64x64 tile: the 72x72 pixels it needs are staged in LDS, scores are computed for the tile plus a
// one-pixel ring (66x66: the neighbours of the tile's own pixels), the 4-point compass test rejects most pixels, the
// survivors are compacted into an LDS list and scored densely (same arc-score formulation as k_fast_cells:
// S = max over the 16 arcs of 9 of min(d) resp. min(-d), minus 1; corner iff S >= thr).  Keypoints inside the
// runByImageBorder(edge) rectangle are appended to the level's list (one global atomic per wave); the list is put into
// FAST's raster order by k_cv_select.
constexpr int kSP = 96; // LDS tile pitch: six 16-byte LDS-DMA chunks from column X0 - 16 on (a 16-byte boundary); the 72 pixels the
constexpr int kSX = 12; // tile needs (X0 - 4 ...) start at byte kSX of a row
constexpr int kSc = 68; // score-map pitch (66 used)

__global__ __launch_bounds__(256) void k_fast_tiles(const uint8_t* __restrict__ pyr, Geometry g, int level, int tiles_x,
                                                    int n_tiles, int n_frames, int thr, CvSelectArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[72 * kSP];
    __shared__ __attribute__((aligned(16))) uint8_t sc[66 * kSc];
    __shared__ __attribute__((aligned(4))) uint16_t list[66 * 66 + 2]; // (at most every scored pixel survives: 20.1 KB in all, eight workgroups per CU)
    __shared__ uint32_t n_list;

    // XCD-aware mapping (as k_fast_cells): all tiles of a frame get ids with the same (id & 7) and meet in one L2.  Grid:
    // x = tile column * 8 + XCD, y = tile row, z = group of 8 frames (no divisions: two runtime divisions per wave were
    // a tenth of the kernel's vector instructions)
    const int xcd = blockIdx.x & 7;
    const int f_local = (int)blockIdx.z * 8 + xcd;
    if(f_local >= n_frames)
        return;
    const LevelGeom& lv = g.lv[level];
    const int w = lv.w, h = lv.h, pitch = lv.pitch;
    const int tx = (int)(blockIdx.x >> 3), ty = (int)blockIdx.y;
    const int X0 = tx * 64, Y0 = ty * 64; // the tile's own block; scores cover (X0-1, Y0-1) + 66x66, pixels (X0-4, Y0-4) + 72x72
    const size_t frame = (size_t)f_local + g.frame0;
    const uint8_t* src = pyr + frame * g.slab + lv.offset;
    const int tid = threadIdx.x, lane = tid & 63;

    // stage the 72 rows by LDS-DMA (as k_fast_cells: no registers, no vector instructions for the copy): chunk t = (row
    // t / 6, chunk t % 6) lands at tile + 16 t.  Rows outside the level repeat its first / last row, chunks outside the
    // row pitch are skipped: what they would hold is only ever read by pixels that are not tested.
#pragma unroll
    for(int p = 0; p < 2; ++p)
    {
        const int t = p * 256 + tid;
        const int r = (t * 10923) >> 16, c = t - r * 6; // t / 6 for t < 32768
        const int y = min(max(Y0 - 4 + r, 0), h - 1), x = X0 - 16 + 16 * c;
        if(t < 72 * 6 && x >= 0 && x < pitch)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)y * pitch + x),
                                             (__attribute__((address_space(3))) void*)&tile[(p * 256 + (tid & ~63)) * 16], 16, 0, 0);
    }
    for(int i = tid; i < 66 * kSc / 4; i += 256)
        reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if(tid == 0)
        n_list = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the wave's DMA chunks have landed
    __syncthreads();

    // compass test over the 66x66 scored region (a 9-arc contains two adjacent compass points of one polarity), FOUR
    // pixels per lane as in k_fast_cells: item i = (scored row r, tile dword q); the scored columns are tile columns
    // 3 .. 68, i.e. byte 3 of dword 0, dwords 1 .. 16 and byte 0 of dword 17.  Items are numbered row-major over the
    // 18-dword rows, so the five LDS reads of a wave are linear in the lane index.
    {
        const uint32_t* T = reinterpret_cast<const uint32_t*>(tile);
        const int tq = min(max(thr, 0), 254);
        const uint32_t kb = (uint32_t)(256 - ((tq + 256) >> 1)) * 0x01010101u; // bit 7 of lerp(q, kb): q >= (t + 256) >> 1
        const uint32_t kd = (uint32_t)(255 - ((254 - tq) >> 1)) * 0x01010101u; // bit 7 of lerp(q, kd): q >  (254 - t) >> 1
        // tested tile columns [c_lo, c_hi) and scored rows [r_lo, r_hi): FAST's 3-pixel frame of the level
        const int c_lo = max(3, 7 - X0), c_hi = min(69, w - 3 - (X0 - 4));
        const int r_lo = max(0, 4 - Y0), r_hi = min(66, h - 3 - (Y0 - 1));
        // (only the items of rows below r_hi: the tiles of a level's last tile row are partly outside FAST's frame)
        const int i_end = min(66, r_hi) * 18;
        uint16_t* seg = reinterpret_cast<uint16_t*>(sc) + (tid >> 6) * 320; // <= 5 rounds x 64 records
        uint32_t n_grp = 0;                                                  // (wave-uniform)
        for(int i0 = (r_lo * 18) & ~255; i0 < i_end; i0 += 256)
        {
            const int i = i0 + tid;
            const int r = (i * 3641) >> 16, q = i - r * 18; // i / 18 for i < 16384
            uint32_t keep = 0;
            if(i < 66 * 18 && r >= r_lo && r < r_hi)
            {
                const int lo4 = min(max(c_lo - 4 * q, 0), 4), hi4 = min(max(c_hi - 4 * q, 0), 4);
                const uint32_t m4 = ((1u << hi4) - 1u) & ~((1u << lo4) - 1u);          // pixels j of the quad that are tested
                const uint32_t colmask = ((m4 * 0x00204081u) & 0x01010101u) << 7;      // as the sign bits of the four bytes
                const uint32_t* row = T + (r + 3) * (kSP / 4) + (kSX / 4) + q;
                const uint32_t L = row[-1], C = row[0], R = row[1];
                const uint32_t U = row[-3 * (kSP / 4)], D = row[3 * (kSP / 4)];
                const uint32_t Lf = __builtin_amdgcn_alignbyte(C, L, 1); // columns x-3 of the four pixels
                const uint32_t Rt = __builtin_amdgcn_alignbyte(R, C, 3); // columns x+3
                // the byte-wise compass test of k_fast_cells (k_fast.hip, phase A): q = (p - c + 255) >> 1 of four pixels per
                // v_lerp_u8, thresholds as bit 7 of a second lerp; a superset of the exact test (the survivors get the exact
                // arc score below)
                const uint32_t nC = ~C;
                const uint32_t q0 = __builtin_amdgcn_lerp(D, nC, 0u), q8 = __builtin_amdgcn_lerp(U, nC, 0u);
                const uint32_t q4 = __builtin_amdgcn_lerp(Rt, nC, 0u), q12 = __builtin_amdgcn_lerp(Lf, nC, 0u);
                const uint32_t br = (__builtin_amdgcn_lerp(q0, kb, 0u) | __builtin_amdgcn_lerp(q8, kb, 0u)) &
                                    (__builtin_amdgcn_lerp(q4, kb, 0u) | __builtin_amdgcn_lerp(q12, kb, 0u));
                const uint32_t nd = (__builtin_amdgcn_lerp(q0, kd, 0u) & __builtin_amdgcn_lerp(q8, kd, 0u)) |
                                    (__builtin_amdgcn_lerp(q4, kd, 0u) & __builtin_amdgcn_lerp(q12, kd, 0u));
                keep = (br | ~nd) & colmask;
            }
            // compaction as in k_fast_cells: ONE ballot per round on "the quad has a flag"; a surviving quad leaves a 16-bit
            // record (item index << 4 | its four flags) in the wave's own segment, expanded to pixels below.  The segments
            // (<= 5 rounds x 64 records x 2 bytes per wave) live in the first 2560 bytes of the score map, which the waves
            // clear again when they are through with them.
            const bool any = keep != 0;
            const unsigned long long vote = __ballot(any);
            if(any)
            {
                const uint32_t nib = ((((keep >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
                seg[n_grp + __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u))] =
                    (uint16_t)(((uint32_t)i << 4) | nib);
            }
            n_grp += (uint32_t)__popcll(vote);
        }
        for(uint32_t g0 = 0; g0 < n_grp; g0 += 64)
        {
            const uint32_t rec = g0 + lane < n_grp ? (uint32_t)seg[g0 + lane] : 0u;
            const bool f0 = (rec & 1u) != 0, f1 = (rec & 2u) != 0, f2 = (rec & 4u) != 0, f3 = (rec & 8u) != 0;
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1), m2 = __ballot(f2), m3 = __ballot(f3);
            const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2), c3 = (uint32_t)__popcll(m3);
            uint32_t base = 0;
            if(lane == 0)
                base = atomicAdd(&n_list, c0 + c1 + c2 + c3);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t it = rec >> 4;
            const uint32_t rr = (it * 3641u) >> 16, qq = it - rr * 18u;          // item -> (scored row, tile dword)
            const uint32_t yx = ((rr + 3u) << 8) | (4u * qq);                     // (tile row, tile column) of the quad's pixel 0
            if(f0)
                list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = (uint16_t)yx;
            base += c0;
            if(f1)
                list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = (uint16_t)(yx + 1);
            base += c1;
            if(f2)
                list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u))] = (uint16_t)(yx + 2);
            base += c2;
            if(f3)
                list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, 0u))] = (uint16_t)(yx + 3);
        }
        for(int z = lane; z < 160; z += 64) // the wave's segment is score map again
            reinterpret_cast<uint32_t*>(seg)[z] = 0u;
    }
    __syncthreads();
    const uint32_t n = n_list;
    for(uint32_t j = tid; 2 * j < n; j += 256) // two listed pixels per lane (arc_score.hpp)
    {
        const uint32_t two = *reinterpret_cast<const uint32_t*>(&list[2 * j]);
        const uint32_t ca = two & 0xFFFFu, cb = 2 * j + 1 < n ? two >> 16 : ca;
        const int ay = (int)(ca >> 8), ax = (int)(ca & 0xFF), by = (int)(cb >> 8), bx = (int)(cb & 0xFF);
        const uint8_t* pa = &tile[ay * kSP + kSX + ax];
        const uint8_t* pb = &tile[by * kSP + kSX + bx];
        uint32_t e[16]; // circle pixel k of both candidates side by side (raw pixels: arc_score2_raw)
#define MSLAM_E(k, off) e[k] = (uint32_t)pa[off] | ((uint32_t)pb[off] << 16)
        MSLAM_E(0, 3 * kSP);
        MSLAM_E(1, 3 * kSP + 1);
        MSLAM_E(2, 2 * kSP + 2);
        MSLAM_E(3, kSP + 3);
        MSLAM_E(4, 3);
        MSLAM_E(5, -kSP + 3);
        MSLAM_E(6, -2 * kSP + 2);
        MSLAM_E(7, -3 * kSP + 1);
        MSLAM_E(8, -3 * kSP);
        MSLAM_E(9, -3 * kSP - 1);
        MSLAM_E(10, -2 * kSP - 2);
        MSLAM_E(11, -kSP - 3);
        MSLAM_E(12, -3);
        MSLAM_E(13, kSP - 3);
        MSLAM_E(14, 2 * kSP - 2);
        MSLAM_E(15, 3 * kSP - 1);
#undef MSLAM_E
        int sa, sb;
        arc_score2_raw(e, (int)pa[0], (int)pb[0], sa, sb);
        if(sa >= thr && sa > 0)
            sc[(ay - 3) * kSc + (ax - 3)] = (uint8_t)sa;
        if(sb >= thr && sb > 0)
            sc[(by - 3) * kSc + (bx - 3)] = (uint8_t)sb;
    }
    __syncthreads();
    // 3x3 strict NMS over the listed pixels of the tile's own 64x64 block + runByImageBorder(edge)
    const int e = a.edge;
    uint32_t* out = a.cand + (frame * g.n_levels + level) * (size_t)a.cand_cap;
    uint32_t* out_cnt = a.cand_cnt + frame * g.n_levels + level;
    for(uint32_t i0 = 0; i0 < n; i0 += 256)
    {
        const uint32_t i = i0 + tid;
        bool is_kp = false;
        uint32_t packed = 0;
        if(i < n)
        {
            const int idx = list[i];
            const int ly = (idx >> 8) - 3, lx = (idx & 0xFF) - 3;
            const int x = X0 - 1 + lx, y = Y0 - 1 + ly;
            const uint8_t* q = &sc[ly * kSc + lx];
            const int s = q[0];
            if(s != 0 && lx >= 1 && lx <= 64 && ly >= 1 && ly <= 64 && x >= e && x < w - e && y >= e && y < h - e)
            {
                const int m = max(max(max(q[-kSc - 1], q[-kSc]), max(q[-kSc + 1], q[-1])),
                                  max(max(q[1], q[kSc - 1]), max(q[kSc], q[kSc + 1])));
                is_kp = s > m;
                packed = pack_kp(x, y, s);
            }
        }
        const unsigned long long m = __ballot(is_kp);
        if(m)
        {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if((tid & 63) == leader)
                base = atomicAdd(out_cnt, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader);
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull));
            if(is_kp && pos < (uint32_t)a.cand_cap)
                out[pos] = packed;
        }
    }
}

// the per-level counters k_fast_tiles appends to.  A kernel, not hipMemsetAsync: the single-frame path replays this
// sequence from a HIP graph, and the memset node of a captured hipMemsetAsync was seen to write stale data on replay
// (ROCm 7.2: the counters came back holding pointer-like garbage once other contexts had run in between).
__global__ void k_zero_u32(uint32_t* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n)
        p[i] = 0u;
}
void launch_zero_u32(uint32_t* p, int n, hipStream_t s)
{
    if(n > 0)
        hipLaunchKernelGGL(k_zero_u32, dim3((n + 255) / 256), dim3(256), 0, s, p, n);
}

void launch_fast_tiles(const uint8_t* d_pyr, const Geometry& g, int thr, const CvSelectArgs& a, int frame0, int n_frames,
                       hipStream_t s)
{
    // one launch per level (ONE launch over the tiles of all levels measured slower: 1.75 vs 1.46 ms per 1000 frames)
    Geometry gg = g;
    gg.frame0 = frame0;
    for(int l = 0; l < g.n_levels; ++l)
    {
        const int tiles_x = (g.lv[l].w + 63) / 64, tiles_y = (g.lv[l].h + 63) / 64, n_tiles = tiles_x * tiles_y;
        const dim3 grid(8u * (unsigned)tiles_x, (unsigned)tiles_y, (unsigned)((n_frames + 7) / 8));
        hipLaunchKernelGGL(k_fast_tiles, grid, dim3(256), 0, s, d_pyr, gg, l, tiles_x, n_tiles, n_frames, thr, a);
    }
}

// ---- retainBest(2n) by FAST score -> Harris -> retainBest(n) by Harris: one workgroup per (level, frame) -------
__device__ __forceinline__ float harris_at(const uint8_t* img, int pitch, int x, int y)
{
    // orb.cpp HarrisResponses: blockSize 7, Sobel-like 3x3 gradients, integer sums, float response.
    // The 9x9 pixels around (x, y) come in as 27 (unaligned) dword loads — rows y-4 .. y+4, bytes x-4 .. x+7 — instead of
    // ~400 byte loads; the gradients are built separably in registers:
    //   V[r][c] = p[r-1][c] + 2 p[r][c] + p[r+1][c]  ->  Ix[r][j] = V[r][j+1] - V[r][j-1]
    //   H[r][c] = p[r][c-1] + 2 p[r][c] + p[r][c+1]  ->  Iy[r][j] = H[r+1][j] - H[r-1][j]
    // A sliding window of three source rows keeps the register count low: row r+1 arrives, V[r] and H[r+1] follow.
    const uint8_t* base = img + (size_t)(y - 4) * pitch + (x - 4);
    struct Raw
    {
        uint32_t d0, d1, d2;
    };
    auto load_raw = [&](int r) { // bytes x-4 .. x+7 of row y-4+r
        const uint8_t* rp = base + (size_t)r * pitch;
        Raw w;
        __builtin_memcpy(&w.d0, rp, 4);
        __builtin_memcpy(&w.d1, rp + 4, 4);
        __builtin_memcpy(&w.d2, rp + 8, 4);
        return w;
    };
    auto unpack = [&](const Raw& w, int* p) { // p[0..8] = pixels x-4 .. x+4
#pragma unroll
        for(int k = 0; k < 4; ++k)
        {
            p[k] = (int)((w.d0 >> (8 * k)) & 0xFFu);
            p[4 + k] = (int)((w.d1 >> (8 * k)) & 0xFFu);
        }
        p[8] = (int)(w.d2 & 0xFFu);
    };
    int pm[9], pc[9], pn[9]; // rows r-1, r, r+1
    int Hm[7], Hc[7], Hn[7]; // H of rows r-1, r, r+1 (columns j = -3 .. 3 -> index j+3, pixel index j+4)
    auto hrow = [&](const int* p, int* H) {
#pragma unroll
        for(int j = 0; j < 7; ++j)
            H[j] = p[j] + 2 * p[j + 1] + p[j + 2];
    };
    unpack(load_raw(0), pm);
    unpack(load_raw(1), pc);
    hrow(pm, Hm);
    hrow(pc, Hc);
    int a = 0, b = 0, c = 0;
    Raw nxt = load_raw(2);
    // a ROLLED loop with the next row's three dwords requested one iteration ahead: fully unrolled, the scheduler hoists
    // all 27 loads and unpacks them at once (235 VGPRs, 2 waves per SIMD, slower than the byte-load version)
#pragma unroll 1
    for(int r = 1; r <= 7; ++r) // window row i = r - 4 = -3 .. 3
    {
        const Raw cur = nxt;
        nxt = load_raw(min(r + 2, 8));
        unpack(cur, pn);
        hrow(pn, Hn);
#pragma unroll
        for(int j = 0; j < 7; ++j)
        {
            // pixel index of column j-3 is j+1; V at columns (j-3)+1 and (j-3)-1 -> pixel indices j+2 and j
            const int Vr = pm[j + 2] + 2 * pc[j + 2] + pn[j + 2], Vl = pm[j] + 2 * pc[j] + pn[j];
            const int Ix = Vr - Vl, Iy = Hn[j] - Hm[j];
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
#pragma unroll
        for(int k = 0; k < 9; ++k)
            pm[k] = pc[k], pc[k] = pn[k];
#pragma unroll
        for(int k = 0; k < 7; ++k)
            Hm[k] = Hc[k], Hc[k] = Hn[k];
    }
    const float harris_k = 0.04f;
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float s4 = __fmul_rn(__fmul_rn(__fmul_rn(scale, scale), scale), scale);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float sum = __fadd_rn(fa, fb);
    const float v = __fsub_rn(__fsub_rn(__fmul_rn(fa, fb), __fmul_rn(fc, fc)), __fmul_rn(__fmul_rn(harris_k, sum), sum));
    return __fmul_rn(v, s4);
}

// ---- KeyPointsFilter::retainBest in the ORDER a GCC build of the reference leaves -------------------------------------------
// retainBest(keypoints, n) = std::nth_element(begin, begin + n - 1, end, response greater) + std::partition(begin + n, end,
// response >= keypoints[n - 1].response) + resize (OpenCV features2d/src/keypoint.cpp): the surviving SET is the standard's,
// their ORDER is what libstdc++'s introselect and partition happen to do — and it is the order of the reference's output
// (orb_feature.cpp:25,40 -> orb.cpp computeKeyPoints), so of every DescriptorMatch index downstream.  Both library loops are
// Hoare-style pointer walks, and a Hoare pass is a PARALLEL operation in disguise: the left pointer only ever stops at
// elements that are "left stoppers" in the ORIGINAL array (response <= pivot), the right pointer at "right stoppers"
// (response >= pivot), swapped elements are never looked at again before the pointers cross, so pass = swap the k-th left
// stopper with the k-th right stopper (from the right) while the former lies left of the latter, and the cut is where the
// left pointer stops next.  Two ordered compactions (ranks by workgroup scans) + one parallel swap per pass; the O(1) pieces
// (median of three, the <= 3-element insertion sort) and the depth-limit fallback (heap select: adversarial inputs only) run
// on one thread, literally as in bits/stl_algo.h / stl_heap.h.  The CPU checker restates the same library code sequentially
// and is pinned against the real <algorithm> of the build image (tests/test_oracle_std_order.py).
// keys / resp: the n elements (both arrays are permuted together); lsp / rsp: scratch for n ranks; returns the new count.
template <class IDX>
__device__ __forceinline__ int retain_best_std(uint32_t* keys, float* resp, int n, int n_points, IDX* lsp, IDX* rsp, uint32_t* wsum,
                                               int* sh)
{
    if(n_points < 0 || n <= n_points)
        return n;
    if(n_points == 0)
        return 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto swap_el = [&](int i, int j) {
        const uint32_t k = keys[i];
        keys[i] = keys[j];
        keys[j] = k;
        const float r = resp[i];
        resp[i] = resp[j];
        resp[j] = r;
    };
    // ranks of the elements of [lo, hi) that satisfy fl (-> lsp, ascending) and fr (-> rsp, ascending); totals in nl, nr
    auto rank2 = [&](int lo, int hi, auto&& fl, auto&& fr, int& nl, int& nr) {
        uint32_t runl = 0, runr = 0;
        for(int base = lo; base < hi; base += 256)
        {
            const int i = base + tid;
            const float v = i < hi ? resp[i] : 0.f;
            const bool bl = i < hi && fl(v), br = i < hi && fr(v);
            const unsigned long long ml = __ballot(bl), mr = __ballot(br);
            if(lane == 0)
                wsum[wave] = (uint32_t)__popcll(ml) | ((uint32_t)__popcll(mr) << 16);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for(int k = 0; k < 4; ++k)
            {
                pre += k < wave ? wsum[k] : 0;
                tot += wsum[k];
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if(bl)
                lsp[runl + (pre & 0xFFFFu) + (uint32_t)__popcll(ml & below)] = (IDX)i;
            if(br)
                rsp[runr + (pre >> 16) + (uint32_t)__popcll(mr & below)] = (IDX)i;
            runl += tot & 0xFFFFu;
            runr += tot >> 16;
            __syncthreads();
        }
        nl = (int)runl;
        nr = (int)runr;
    };
    // number of k < min(nl, nr) with lsp[k] < rsp[nr - 1 - k] (the condition is monotone in k), swaps of those pairs
    auto pair_swaps = [&](int nl, int nr) -> int {
        const int m = min(nl, nr);
        if(tid == 0)
            sh[0] = 0;
        __syncthreads();
        int mine = 0;
        for(int k = tid; k < m; k += 256)
            mine += (int)lsp[k] < (int)rsp[nr - 1 - k] ? 1 : 0;
        for(int o = 32; o > 0; o >>= 1)
            mine += __shfl_xor(mine, o);
        if(lane == 0 && mine)
            atomicAdd(&sh[0], mine);
        __syncthreads();
        const int K = sh[0];
        for(int k = tid; k < K; k += 256)
            swap_el((int)lsp[k], (int)rsp[nr - 1 - k]);
        __syncthreads();
        return K;
    };
    auto comp = [&](int a, int b) { return resp[a] > resp[b]; }; // KeypointResponseGreater on positions

    // ---- std::nth_element(begin, begin + n_points - 1, end) = __introselect(..., 2 * __lg(n))
    int first = 0, last = n;
    const int nth = n_points - 1;
    int depth = 2 * (31 - __builtin_clz((unsigned)n));
    bool heap_done = false;
    while(last - first > 3)
    {
        if(depth == 0)
        {
            // __heap_select(first, nth + 1, last) + iter_swap(first, nth): one thread, as the library does it
            if(tid == 0)
            {
                const int len = nth + 1 - first;
                auto el_r = [&](int i) -> float& { return resp[first + i]; };
                auto el_k = [&](int i) -> uint32_t& { return keys[first + i]; };
                auto push_heap = [&](int hole, int top, float vr, uint32_t vk) {
                    int parent = (hole - 1) / 2;
                    while(hole > top && el_r(parent) > vr)
                    {
                        el_r(hole) = el_r(parent);
                        el_k(hole) = el_k(parent);
                        hole = parent;
                        parent = (hole - 1) / 2;
                    }
                    el_r(hole) = vr;
                    el_k(hole) = vk;
                };
                auto adjust_heap = [&](int hole, int hlen, float vr, uint32_t vk) {
                    const int top = hole;
                    int second = hole;
                    while(second < (hlen - 1) / 2)
                    {
                        second = 2 * (second + 1);
                        if(el_r(second) > el_r(second - 1))
                            second--;
                        el_r(hole) = el_r(second);
                        el_k(hole) = el_k(second);
                        hole = second;
                    }
                    if((hlen & 1) == 0 && second == (hlen - 2) / 2)
                    {
                        second = 2 * (second + 1);
                        el_r(hole) = el_r(second - 1);
                        el_k(hole) = el_k(second - 1);
                        hole = second - 1;
                    }
                    push_heap(hole, top, vr, vk);
                };
                if(len >= 2) // __make_heap
                    for(int parent = (len - 2) / 2;; --parent)
                    {
                        adjust_heap(parent, len, el_r(parent), el_k(parent));
                        if(parent == 0)
                            break;
                    }
                for(int i = nth + 1; i < last; ++i)
                    if(resp[i] > resp[first]) // __pop_heap(first, middle, i)
                    {
                        const float vr = resp[i];
                        const uint32_t vk = keys[i];
                        resp[i] = resp[first];
                        keys[i] = keys[first];
                        adjust_heap(0, len, vr, vk);
                    }
                swap_el(first, nth);
            }
            __syncthreads();
            heap_done = true;
            break;
        }
        --depth;
        // __unguarded_partition_pivot: median of (first + 1, mid, last - 1) to first ...
        if(tid == 0)
        {
            const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
            int m;
            if(comp(a, b))
                m = comp(b, c) ? b : comp(a, c) ? c : a;
            else
                m = comp(a, c) ? a : comp(b, c) ? c : b;
            swap_el(first, m);
        }
        __syncthreads();
        // ... then __unguarded_partition(first + 1, last, pivot = *first)
        const float pv = resp[first];
        int nl, nr;
        rank2(first + 1, last, [&](float v) { return !(v > pv); }, [&](float v) { return !(pv > v); }, nl, nr);
        const int K = pair_swaps(nl, nr);
        // where the left pointer stops next: the next original left stopper if it lies before the last swapped right position,
        // else that position (it holds a left stopper now)
        int cut;
        if(K < nl && (K == 0 || (int)lsp[K] < (int)rsp[nr - K]))
            cut = (int)lsp[K];
        else
            cut = (int)rsp[nr - K]; // = rsp[nr - 1 - (K - 1)]; K >= 1 here (the median guarantees a left stopper)
        __syncthreads(); // lsp / rsp are rewritten by the next pass
        if(cut <= nth)
            first = cut;
        else
            last = cut;
    }
    if(!heap_done)
    {
        // __insertion_sort(first, last) on at most three elements
        if(tid == 0)
            for(int i = first + 1; i < last; ++i)
            {
                const float vr = resp[i];
                const uint32_t vk = keys[i];
                if(vr > resp[first])
                {
                    for(int j = i; j > first; --j)
                        resp[j] = resp[j - 1], keys[j] = keys[j - 1];
                    resp[first] = vr;
                    keys[first] = vk;
                }
                else
                {
                    int j = i;
                    while(vr > resp[j - 1])
                    {
                        resp[j] = resp[j - 1];
                        keys[j] = keys[j - 1];
                        --j;
                    }
                    resp[j] = vr;
                    keys[j] = vk;
                }
            }
        __syncthreads();
    }
    // ---- std::partition(begin + n_points, end, response >= ambiguous): the k-th element from the left that fails with the
    //      k-th from the right that passes, while the former lies left of the latter; the new end = n_points + #passing
    const float amb = resp[n_points - 1];
    int nf, np;
    rank2(n_points, n, [&](float v) { return !(v >= amb); }, [&](float v) { return v >= amb; }, nf, np);
    pair_swaps(nf, np);
    return n_points + np;
}

// The body of k_cv_select on one (level, frame): `keys` holds the level's n keypoints (any order) and is sorted in place,
// `kept` / `resp` receive the survivors of the first retainBest and their Harris responses (kept may alias keys: the
// compaction only writes positions it has already read).  Force-inlined into two callers so that in the common case
// every array is a known LDS object.
template <class IDX>
__device__ __forceinline__ void cv_select_run(uint32_t* keys, uint32_t* kept, float* resp, int n, int quota, const uint8_t* img,
                                              int pitch, uint32_t* sel, float* sresp, uint32_t* sel_cnt, uint32_t* hist,
                                              uint32_t* wsum, int& s_thr, uint32_t* sorted_out, int std_order, IDX* lsp, IDX* rsp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The tile kernel appended the level's keypoints in arrival order; FAST's own order — the order everything
    // downstream is defined in — is raster (y, then x), which is ascending order of the packed words: bitonic sort.
    {
        int np2 = 1;
        while(np2 < n)
            np2 <<= 1;
        // normalised bitonic network (every comparator ascending; the first step of a merge mirrors the upper half), so a
        // length that is not a power of two only needs comparators reaching beyond n to be skipped
        for(int lsize = 1; (1 << lsize) <= np2; ++lsize)
            for(int ls = lsize - 1; ls >= 0; --ls) // stride = 1 << ls: shifts and masks, no integer division
            {
                const int stride = 1 << ls, size = 1 << lsize;
                for(int t = tid; t < (np2 >> 1); t += 256)
                {
                    const int grp = t >> ls, off = t & (stride - 1);
                    int lo, hi;
                    if(ls == lsize - 1)
                    {
                        lo = (grp << lsize) + off;
                        hi = (grp << lsize) + size - 1 - off;
                    }
                    else
                    {
                        lo = (grp << (ls + 1)) + off;
                        hi = lo + stride;
                    }
                    if(hi < n)
                    {
                        const uint32_t x = keys[lo], y = keys[hi];
                        if(x > y)
                        {
                            keys[lo] = y;
                            keys[hi] = x;
                        }
                    }
                }
                __syncthreads();
            }
    }
    // the level's list in FAST order stays readable in global memory (mslam_hip_debug_read) when it was sorted elsewhere
    if(sorted_out != nullptr)
        for(int i = tid; i < n; i += 256)
            sorted_out[i] = keys[i];
    // ordered compaction over [0, count): `pred(i, v)` sees element i's value v = load(i), read BEFORE any store of the
    // same round, so the destination may alias the source; `emit(i, v, pos)` stores it.  Returns the total.
    auto compact = [&](int count, auto&& load, auto&& pred, auto&& emit) -> int {
        uint32_t running = 0;
        for(int base = 0; base < count; base += 256)
        {
            const int i = base + tid;
            const uint32_t v = i < count ? load(i) : 0u;
            const bool ok = i < count && pred(i, v);
            const unsigned long long m = __ballot(ok);
            if(lane == 0)
                wsum[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t pre = 0, tot = 0;
            for(int k = 0; k < 4; ++k)
            {
                pre += k < wave ? wsum[k] : 0;
                tot += wsum[k];
            }
            if(ok)
                emit(i, v, running + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
            running += tot;
            __syncthreads();
        }
        return (int)running;
    };

    if(std_order)
    {
        // the reference's order (a GCC build): both retainBest calls as libstdc++ runs them, in place on (keys, resp)
        __syncthreads();
        for(int i = tid; i < n; i += 256)
            resp[i] = (float)kp_score(keys[i]); // KeyPoint::response after FAST
        __syncthreads();
        const int m1 = retain_best_std<IDX>(keys, resp, n, 2 * quota, lsp, rsp, wsum, &s_thr);
        __syncthreads();
        for(int i = tid; i < m1; i += 256)
        {
            const uint32_t p = keys[i];
            resp[i] = harris_at(img, pitch, kp_x(p), kp_y(p));
        }
        __syncthreads();
        const int m2 = retain_best_std<IDX>(keys, resp, m1, quota, lsp, rsp, wsum, &s_thr);
        __syncthreads();
        for(int i = tid; i < m2; i += 256)
        {
            const uint32_t p = keys[i];
            sel[i] = pack_kp(kp_x(p) - kBorder, kp_y(p) - kBorder, kp_score(p));
            sresp[i] = resp[i];
        }
        if(tid == 0)
            *sel_cnt = (uint32_t)m2;
        return;
    }
    // 1. retainBest(2 * quota) by FAST score (integer 1..255): threshold = the (2 quota)-th largest score
    int thr = 0;
    if(n > 2 * quota)
    {
        hist[tid] = 0;
        __syncthreads();
        for(int i = tid; i < n; i += 256)
            atomicAdd(&hist[kp_score(keys[i])], 1u);
        __syncthreads();
        if(tid == 0)
        {
            int acc = 0, t = 255;
            for(; t > 0; --t)
            {
                acc += (int)hist[t];
                if(acc >= 2 * quota)
                    break;
            }
            s_thr = 2 * quota == 0 ? 256 : t;
        }
        __syncthreads();
        thr = s_thr;
    }
    const int m1 = compact(
        n, [&](int i) { return keys[i]; }, [&](int, uint32_t v) { return kp_score(v) >= thr; },
        [&](int, uint32_t v, uint32_t pos) { kept[pos] = v; });
    __syncthreads();
    // 2. Harris responses (the writes above are visible to the whole workgroup after the barrier)
    for(int i = tid; i < m1; i += 256)
    {
        const uint32_t p = kept[i];
        resp[i] = harris_at(img, pitch, kp_x(p), kp_y(p));
    }
    __syncthreads();
    // 3. retainBest(quota) by Harris: keep iff fewer than `quota` responses are strictly greater (= response >= the
    //    quota-th largest; ties kept)
    const int m2 = compact(
        m1, [&](int i) { return kept[i]; },
        [&](int i, uint32_t) {
            if(m1 <= quota)
                return true;
            if(quota == 0)
                return false;
            const float r = resp[i];
            int greater = 0;
            for(int j = 0; j < m1; ++j)
                greater += resp[j] > r ? 1 : 0;
            return greater < quota;
        },
        [&](int i, uint32_t p, uint32_t pos) {
            // k_describe takes coordinates relative to the (19, 19) origin of the in-tree detector's lists
            sel[pos] = pack_kp(kp_x(p) - kBorder, kp_y(p) - kBorder, kp_score(p));
            sresp[pos] = resp[i];
        });
    if(tid == 0)
        *sel_cnt = (uint32_t)m2;
}

constexpr int kSelLds = 4096;      // levels with at most this many FAST keypoints are selected entirely in LDS
constexpr int kSelLdsSmall = 1024; // ... and those with at most this many by the small instance (9 KB of LDS instead of 34)

// Two instances on the same grid (as k_quadtree's classes): a (level, frame) pair is taken by the instance whose LDS
// arrays its keypoints fit — the passes are barrier + LDS latency, so workgroups per CU set the rate — and the other
// instance leaves at once (the count is one load).
template <int KP, bool SMALL>
__global__ __launch_bounds__(256) void k_cv_select(const uint8_t* __restrict__ pyr, Geometry g, CvSelectArgs a, int take_all)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ int s_thr;
    __shared__ uint32_t l_keys[KP];
    __shared__ float l_resp[KP];
    __shared__ uint16_t l_lsp[KP], l_rsp[KP]; // stopper ranks of retain_best_std
    // Workgroups go to the 8 XCDs round-robin by linear id = frame * n_levels + blockIdx.x: with level = blockIdx.x and the
    // usual 8 levels XCD 0 would get every level-0 pair (the heaviest: thousands of keys to sort) and XCD 7 every level-7
    // one; rotating the level by the frame index gives every XCD the same mix.
    const int level = (int)((blockIdx.x + blockIdx.y) % (unsigned)g.n_levels);
    const size_t frame = blockIdx.y + g.frame0;
    const LevelGeom& lv = g.lv[level];
    const size_t slot = frame * g.n_levels + level;
    const int tid = threadIdx.x;
    const int n_raw = (int)a.cand_cnt[slot];
    if(!take_all && SMALL != (n_raw <= kSelLdsSmall))
        return; // the other instance's pair
    uint32_t* cand = a.cand + slot * (size_t)a.cand_cap;
    uint32_t* sel = a.sel + slot * (size_t)a.cand_cap;
    float* sresp = a.sel_resp + slot * (size_t)a.cand_cap;
    const uint8_t* img = pyr + frame * g.slab + lv.offset;
    if(n_raw > a.cand_cap && tid == 0)
        atomicOr(a.flags, kFlagCandOverflow);
    const int n = min(n_raw, a.cand_cap);
    const int quota = a.quota[level];
    if(n <= KP)
    {
        for(int i = tid; i < n; i += 256)
            l_keys[i] = cand[i];
        __syncthreads();
        // a sort stage is a barrier + LDS traffic here; on the global arrays it was a round trip to L2 (55 stages for
        // 1024 keypoints: the whole kernel ran at ~250 us per workgroup)
        cv_select_run<uint16_t>(l_keys, l_keys, l_resp, n, quota, img, lv.pitch, sel, sresp, a.sel_cnt + slot, hist, wsum, s_thr, cand,
                                a.std_order, l_lsp, l_rsp);
    }
    else // (global arrays; the stopper ranks of the library-order form borrow tmp_kp and — until the final emission — sel)
        cv_select_run<uint32_t>(cand, a.std_order ? cand : a.tmp_kp + slot * (size_t)a.cand_cap, a.tmp_resp + slot * (size_t)a.cand_cap, n,
                                quota, img, lv.pitch, sel, sresp, a.sel_cnt + slot, hist, wsum, s_thr, nullptr, a.std_order,
                                a.tmp_kp + slot * (size_t)a.cand_cap, sel);
}

void launch_cv_select(const uint8_t* d_pyr, const Geometry& g, const CvSelectArgs& a, int frame0, int n_frames,
                      hipStream_t s)
{
    Geometry gg = g;
    gg.frame0 = frame0;
    // a handful of frames (the synchronous single-frame call): one launch, the large instance takes every pair
    const int take_all = n_frames < 8 ? 1 : 0;
    hipLaunchKernelGGL((k_cv_select<kSelLds, false>), dim3(g.n_levels, n_frames), dim3(256), 0, s, d_pyr, gg, a, take_all);
    if(!take_all)
        hipLaunchKernelGGL((k_cv_select<kSelLdsSmall, true>), dim3(g.n_levels, n_frames), dim3(256), 0, s, d_pyr, gg, a, 0);
}

} // namespace mslam
