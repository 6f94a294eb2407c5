// common.hpp — shared declarations of the gfx950 kernels and the context behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

namespace mslam
{

constexpr int kMaxLevels = 16;
constexpr int kBorder = 19;    // orb_patch_radius_, reference distributed_cv_feature.cpp:699
constexpr int kCell = 64;      // cell_size, :853
constexpr int kOverlap = 6;    // overlap,   :852
constexpr int kCellCap = 1024; // NMS keeps at most one corner per 2x2 block of a 64x64 tested area

// One pyramid level inside a frame's slab.  All fields are filled on the host at context creation.
struct LevelGeom
{
    int w, h;       // level size in pixels (reference :836-837)
    int pitch;      // row pitch in bytes (multiple of 16)
    int offset;     // byte offset of the level inside the frame slab
    float scale;    // scale_factors_[l], float32 chain (:411-420)
    int bw, bh;     // bordered width/height = w-38, h-38
    int cell_base;  // index of this level's first cell in the cell table
    int n_cells;    // cells kept after the skip rule (:881-905)
    int bsx;        // blur (k_blur2): 4-px strips per row = ceil(w / 4)
    // quadtree initial grid (:1031-1052)
    int nxg, nyg;
    double delta_x, delta_y;
};

struct Geometry
{
    int n_levels;
    int W, H;
    int n_cells;
    int frame0;    // first frame this launch works on (blockIdx is relative to it)
    unsigned slab; // bytes per frame (all levels)
    int blur_tiled; // the blurred slab is stored in kTileW x kTileH pixel tiles (tiled_off) instead of rows: set when every level comes from k_level.hip
    LevelGeom lv[kMaxLevels];
};

// Tiled plane layout: a level plane of pitch p (a multiple of kTileW) and h rows (stored as whole tile rows) is cut into
// tiles of kTileW pixels x kTileH rows = 128 bytes = one cache line, tiles of a tile row side by side, rows of a tile one
// after the other.  A window of r rows x c columns then touches about (r/kTileH + 1)(c/kTileW + 1) lines instead of
// r (1 + c/128): what k_describe's window gathers pay for.
#ifndef MSLAM_TILE_W
#define MSLAM_TILE_W 64
#endif
constexpr int kTileW = MSLAM_TILE_W, kTileH = 128 / kTileW;
constexpr int kTileWLog = kTileW == 64 ? 6 : kTileW == 32 ? 5 : 4, kTileHLog = 7 - kTileWLog;
static_assert(kTileW == 16 || kTileW == 32 || kTileW == 64, "tile width");
__host__ __device__ __forceinline__ unsigned tiled_off(unsigned pitch, int x, int y)
{
    return (unsigned)(y >> kTileHLog) * (pitch * (unsigned)kTileH) + (unsigned)(x >> kTileWLog) * 128u +
           (unsigned)(y & (kTileH - 1)) * (unsigned)kTileW + (unsigned)(x & (kTileW - 1));
}
// an LDS pointer from a 32-bit LDS byte offset (device pass: LDS pointers are 32 bits wide; the host pass only parses this)
template <class P>
__device__ __forceinline__ P lds_ptr(uint32_t off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (P)off;
#else
    return (P)(uintptr_t)off;
#endif
}
// FAST cell: a sub-image [x0, x0+cw) x [y0, y0+ch) of one level (:880-905).
struct CellDesc
{
    int16_t level, cw, ch, pad;
    int16_t x0, y0; // absolute level coordinates of the sub-image origin (min_x, min_y)
    int16_t ox, oy; // j*64, i*64: added to cell-relative coordinates (:940-941)
};

constexpr int kBlurRows = 32; // output rows per blur strip thread

// candidate / selected keypoint: (Y << 20) | (X << 8) | score, X/Y relative to the (19,19) border origin
__host__ __device__ inline uint32_t pack_kp(int x, int y, int score) { return ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)score; }
__host__ __device__ inline int kp_x(uint32_t p) { return (int)((p >> 8) & 0xFFFu); }
__host__ __device__ inline int kp_y(uint32_t p) { return (int)(p >> 20); }
__host__ __device__ inline int kp_score(uint32_t p) { return (int)(p & 0xFFu); }

// status flags written by kernels (device word, OR-ed)
enum : uint32_t
{
    kFlagCandOverflow = 1u,  // more FAST candidates on a level than max_candidates
    kFlagKpOverflow = 2u,    // more keypoints in a frame than max_keypoints
    kFlagQuadNoConverge = 4u, // quadtree pass limit hit (cannot happen for sane sizes)
    kFlagBowPackOverflow = 8u, // a BoW vector has more words than the exchange format's k_max
    kFlagDbFull = 16u,         // the BoW database's posting log is full (mslam_hip_bow_db_reserve)
    kFlagPackOverflow = 32u    // mslam_hip_pack_batch_dev: the packed results do not fit the caller's buffer
};

// ---- kernel launchers (each enqueues on `s`, no synchronisation) ----------------------------------
// every launcher works on frames [frame0, frame0 + n_frames) of the batch buffers
void launch_gray(const uint8_t* d_bgr, uint8_t* d_pyr, const Geometry& g, int frame0, int n_frames, hipStream_t s);
void launch_resize(uint8_t* d_pyr, const Geometry& g, int level, const int32_t* d_xofs, const uint32_t* d_xcoef,
                   const int32_t* d_yofs, const uint32_t* d_ycoef, int frame0, int n_frames, hipStream_t s);
struct ResizeColArgs
{
    uint8_t* pyr;
    unsigned slab;
    int src_off, sh, spitch, dst_off, dw, dh, dpitch;
    const uint4* qt;       // [quads][3]: {byte offset, upper flags, sel0, sel1} {sel2, sel3, coef0, coef1} {coef2, coef3, -, -}
    const int32_t* yofs;   // [dh]
    const uint32_t* ycoef; // [dh] b0 | b1 << 16
    int frame0, n_frames;
    int quads;
    float inv_quads;
    int R;         // destination rows per lane
    int need_mask; // bit k: pixel k of some quad takes its pair from dwords (1,2)
    int exact;     // 0: INTER_LINEAR (11-bit weights, cv::resize's two-step rounding); 1: INTER_LINEAR_EXACT (8.8 weights)
};
void launch_resize_col(const ResizeColArgs& a, hipStream_t s);
void launch_fast(const uint8_t* d_pyr, const Geometry& g, const CellDesc* d_cells, uint32_t* d_cell_cnt,
                 uint32_t* d_cell_kp, int ini_thr, int min_thr, int frame0, int n_frames, hipStream_t s);
// writes 1 to *d_out when v_pk_minimum3_f16 / v_pk_maximum3_f16 order f16 denormal bit patterns like integers in this build
// (what the FAST score kernels rely on: arc_score.hpp); checked once per context by mslam_hip_create
void launch_denorm_selfcheck(uint32_t* d_out, hipStream_t s);
struct QuadArgs
{
    const uint32_t* cell_cnt; // [B][n_cells]
    const uint32_t* cell_kp;  // [B][n_cells][kCellCap]
    uint32_t* cand;           // [B][L][cand_cap]   packed candidates in FAST order
    uint32_t* cand_cnt;       // [B][L]
    uint32_t* sel;            // [B][L][cand_cap]   packed selected keypoints in node-list order
    uint32_t* sel_cnt;        // [B][L]
    // scratch, per (frame, level) block:
    uint32_t* kp_node;  // [B][L][cand_cap]
    uint2* nodes_a;     // [B][L][cand_cap]  (bx|by<<16, ex|ey<<16)
    uint2* nodes_b;
    uint32_t* ncnt_a;   // [B][L][cand_cap]  keypoints per node
    uint32_t* ncnt_b;
    uint32_t* child_cnt; // [B][L][cand_cap*4]
    uint32_t* ninfo;     // [B][L][cand_cap]
    uint32_t* best;      // [B][L][cand_cap]
    uint32_t* flags;     // status word
    int cand_cap;
    unsigned min_size;
};
void launch_quadtree(const Geometry& g, const QuadArgs& a, int frame0, int n_frames, hipStream_t s);
// the 7 Gaussian taps arranged for the kernels (k_blur2, k_level.hip)
struct BlurK
{
    uint32_t ta[3], tb[4], tc[4]; // horizontal taps positioned for output pixel j on window dwords A / B / C (tc[0] unused)
    uint32_t t01, t23, t45, t6;   // vertical taps as u16 pairs
};
BlurK make_blur_k();
// k_level.hip: level 0 = gray + blur in one pass
struct GrayBlurArgs
{
    const uint8_t* bgr; // batch of BGR frames
    uint8_t* pyr;       // raw slab of frame 0 (level 0 starts it)
    uint8_t* blur;      // blurred slab of frame 0
    int W, H, pitch;
    unsigned slab;
    int n_frames, frame0;
    int quads;          // W / 4
    float inv_quads;
    int k6;             // rows per block = 6 k6 + 2
    int waves_per_xcd;  // filled in by the launcher
    int blur_tiled;     // Geometry::blur_tiled
    BlurK bk;
};
void launch_gray_blur(const GrayBlurArgs& a, hipStream_t s);
// k_level.hip: level l > 0 = resize (k_resize_col's tables) + blur in one pass
struct ResizeBlurArgs
{
    uint8_t* pyr;
    uint8_t* blur;
    unsigned slab;
    int src_off, sh, spitch, dst_off, dw, dh, dpitch;
    const uint4* qt;       // [quads][3], as ResizeColArgs
    const int32_t* yofs;   // [dh]
    const uint32_t* ycoef; // [dh] b0 | b1 << 16
    int frame0, n_frames;
    int quads;
    float inv_quads;
    int k6;        // rows per block = 6 k6 + 2 (<= 58: the block's row table lives in lane registers)
    int need_mask; // bit k: pixel k of some quad takes its pair from dwords (1,2)
    int exact;     // 0: INTER_LINEAR, 1: INTER_LINEAR_EXACT
    int waves_per_xcd; // filled in by the launcher
    int blur_tiled;    // Geometry::blur_tiled
    int always_load;   // 1: load the upper source-row window even where the walk does not look at it (latency-bound launches)
    BlurK bk;
};
void launch_resize_blur(const ResizeBlurArgs& a, hipStream_t s);
// k_level.hip: gray + blur and every resize + blur level of a batch in ONE launch (a small workgroup walks the levels of its
// frames in sequence); false = this pyramid cannot take the chain (the caller launches per level)
constexpr int kChainLevels = 8;
struct LevelChainArgs
{
    GrayBlurArgs g;
    ResizeBlurArgs lv[kChainLevels - 1];
    int n_lv;           // resize levels (pyramid levels - 1)
    int G;              // frames per workgroup
    int n_groups, groups_per_xcd;
    int frame0, n_frames;
};
bool launch_level_chain(const GrayBlurArgs& g, const ResizeBlurArgs* lv, int n_lv, int frames_per_group, int waves, int k6_max, hipStream_t s);
// k_blur2: one descriptor per wave of a frame (level-uniform waves of 64 consecutive (band, strip) items)
struct BlurWave
{
    int level;
    int item0;       // first item: item = band * strips_per_row + strip (4-px strips, 32-row bands)
    int generic;     // 1: the level is lower than 38 rows, every row takes the per-lane reflect path
    float inv_bsx;   // 1 / strips per row
};
void launch_blur(const uint8_t* d_pyr, uint8_t* d_blur, const Geometry& g, const BlurWave* d_waves, int wpf, int frame0, int n_frames,
                 hipStream_t s);
struct DescArgs
{
    const uint8_t* pyr;
    const uint8_t* blur;
    const uint32_t* sel;
    const uint32_t* sel_cnt;
    const uint32_t* orient_w; // [2][256] disc weights: u bytes, then v bytes (api.hip: build_orient_weights)
    int cand_cap;
    int max_kp;
    float* xy;
    uint8_t* desc;
    int32_t* octave;
    float* angle;
    float* response;
    int32_t* count;
    uint32_t* flags;
    int n_frames = 0; // filled in by the launcher
    // cv::ORB mode (orb_feature.cpp:25 -> OpenCV orb.cpp): the keypoint response is the Harris response of the
    // selection kernel, and cos / sin of the angle follow computeOrbDescriptors (float degree -> radian product,
    // include/mslam_sincos.h in place of the host libm) instead of the in-tree util::cos / util::sin
    const float* sel_resp = nullptr; // [B][L][cand_cap], parallel to sel
    // the synchronous single-frame call: the page-locked, device-mapped result block of the context (count, flags, then xy /
    // descriptors / octave / angle / response at capacity strides).  When set, k_describe writes every result there as well —
    // its stores are the transfer, no packing kernel follows (5 us of the call)
    uint8_t* h_mirror = nullptr;
    int cv_mode = 0;
};
void launch_describe(const Geometry& g, const DescArgs& a, int frame0, int n_frames, hipStream_t s);

// ---- cv::ORB detector mode (k_cvorb.hip) ----------------------------------------------------------------------
struct ExactResizeArgs
{
    uint8_t* pyr;
    unsigned slab;
    int src_off, sw, sh, spitch, dst_off, dw, dh, dpitch;
    const int32_t* xofs;   // [dw]
    const uint32_t* xcoef; // [dw] c0 | c1 << 16 (8.8 fixed point, c0 + c1 = 256)
    const int32_t* yofs;   // [dh]
    const uint32_t* ycoef; // [dh]
    int xmin, xmax, ymin, ymax; // destination positions outside [min, max) copy the first / last source sample (folded into the tables)
    int window12;               // 1: the 4 pixels of every destination quad read inside one aligned 12-byte source window
    int frame0;
};
void launch_resize_exact(const ExactResizeArgs& a, int n_frames, hipStream_t s);
struct CvSelectArgs
{
    uint32_t* cand;     // [B][L][cand_cap] FAST keypoints after NMS + border filter, raster order, ABSOLUTE coordinates
    uint32_t* cand_cnt; // [B][L]
    uint32_t* tmp_kp;   // [B][L][cand_cap] scratch
    float* tmp_resp;    // [B][L][cand_cap] scratch
    uint32_t* sel;      // [B][L][cand_cap] final keypoints, coordinates relative to (19, 19) (k_describe's convention)
    float* sel_resp;    // [B][L][cand_cap] Harris responses
    uint32_t* sel_cnt;  // [B][L]
    uint32_t* flags;
    int cand_cap;
    int edge;           // edgeThreshold (31)
    int quota[kMaxLevels]; // nfeaturesPerLevel
    int std_order;         // 1: retainBest leaves its survivors where libstdc++'s nth_element + partition put them (the reference's order), 0: raster order
};
void launch_zero_u32(uint32_t* p, int n, hipStream_t s);
void launch_fast_tiles(const uint8_t* d_pyr, const Geometry& g, int thr, const CvSelectArgs& a, int frame0, int n_frames,
                       hipStream_t s);
void launch_cv_select(const uint8_t* d_pyr, const Geometry& g, const CvSelectArgs& a, int frame0, int n_frames,
                      hipStream_t s);

// knn-2 Hamming match for `n_pairs` independent (from, to) pairs.  Descriptor sets are addressed as
// base + pair_index * stride; counts come from device arrays (or fixed values when the pointer is null).
struct MatchArgs
{
    const uint8_t* from_desc;
    const uint8_t* to_desc;
    long long from_stride, to_stride; // bytes between consecutive pairs
    const int32_t* from_cnt;          // [n_pairs] (stride 1) or nullptr -> n_from_fixed
    const int32_t* to_cnt;
    int n_from_fixed, n_to_fixed;
    int cap;          // per-pair capacity of the outputs (max n_to)
    int32_t* idx0;    // [n_pairs][cap]
    int32_t* idx1;
    int32_t* dist0;
    int32_t* dist1;
    int cap_from = 0;                 // upper bound of from_cnt[] (per-pair capacity of the train side)
    int popcount_only = 0;            // 1: the xor/popcount kernel whatever the train size (mslam_hip_set_matcher)
    // matrix-core kernel only, a handful of pairs: the train set is cut into n_slices, scanned by separate workgroups, and
    // merged by k_match_merge.  partial: [n_pairs][n_slices][2][cap] top-2 keys; n_slices <= 1 or partial == nullptr: off
    uint32_t* partial = nullptr;
    int n_slices = 0;
    int n_pairs = 0, wg_per_pair = 0; // filled in by the launcher
};
int launch_match_knn2(const MatchArgs& a, int n_pairs, hipStream_t s); // returns the kernel taken: 1 = matrix cores, 2 = xor/popcount
// ratio test + ordered compaction (orb_feature.cpp:99-114).  thr[d1] = largest d0 accepted + 1.
struct RatioArgs
{
    const int32_t* idx0;
    const int32_t* dist0;
    const int32_t* dist1;
    const int32_t* from_cnt;
    const int32_t* to_cnt;
    int n_from_fixed, n_to_fixed;
    int cap;
    const int32_t* thr; // [257]: accept iff d0 < thr[d1]
    int32_t* from_idx;  // [n_pairs][cap]
    int32_t* to_idx;
    int32_t* n_out;     // [n_pairs]
};
void launch_ratio_compact(const RatioArgs& a, int n_pairs, hipStream_t s);
// ONE pair, matrix-core kernel with sliced train tiles, then merge + ratio test + compaction in one launch (mslam_hip_match);
// false: the arguments do not fit this form and nothing was launched
bool launch_match_ratio_single(const MatchArgs& a, const RatioArgs& r, hipStream_t s);

} // namespace mslam
