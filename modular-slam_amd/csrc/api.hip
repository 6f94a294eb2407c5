// api.hip — C ABI (include/mslam_hip.h): context, host-built geometry/tables, launch sequencing.
//
// Host logic mirrors the scalar set-up code of the reference:
//   level scales / sizes  : distributed_cv_feature.cpp:411-420, :836-837
//   cell grid + skip rule : :862-905
//   quadtree initial grid : :1031-1052
//   resize coefficients   : cv::resize(INTER_LINEAR) table construction (OpenCV imgproc/resize.cpp)
//   blur taps             : cv::getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (8.8)
//   ratio-test table      : orb_feature.cpp:101 evaluated for every integer distance pair
#include "context.hpp"

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>

using namespace mslam;

static thread_local std::string g_create_error;

#define HIPCHK(c, call)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if(e_ != hipSuccess)                                                                                           \
        {                                                                                                              \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                              \
            return MSLAM_HIP_E_RUNTIME;                                                                                \
        }                                                                                                              \
    } while(0)

// HIP's current device is per host thread: make the context's device current on every entry so that
// allocations and launches land on it no matter what the caller did in between.
#define ENTER(c)                                                                                                       \
    do                                                                                                                 \
    {                                                                                                                  \
        if(!(c))                                                                                                       \
            return MSLAM_HIP_E_INVALID;                                                                                \
        HIPCHK(c, hipSetDevice((c)->p.device));                                                                        \
    } while(0)

static int fail(mslam_hip_ctx* c, int code, const std::string& msg)
{
    c->err = msg;
    return code;
}

// ---- host helpers ---------------------------------------------------------------------------------
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}
static inline uint32_t sat_coef(float v)
{
    int iv = cv_round_f(v);
    iv = iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv;
    return (uint32_t)(iv & 0xFFFF);
}

// xofs/alpha (clamp = true) or yofs/beta (clamp = false) of cv::resize INTER_LINEAR
static void resize_table(int ssize, int dsize, bool clamp, std::vector<int32_t>& ofs, std::vector<uint32_t>& coef)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for(int d = 0; d < dsize; ++d)
    {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        if(clamp)
        {
            if(s < 0)
                f = 0, s = 0;
            if(s >= ssize - 1)
                f = 0, s = ssize - 1;
        }
        ofs.push_back(s);
        coef.push_back(sat_coef((1.f - f) * 2048) | (sat_coef(f * 2048) << 16));
    }
}

static void gaussian_taps_fixed(int taps[7])
{
    const int n = 7, n2 = 3;
    const double sigma = 2.0, scale2x = -0.125 / (sigma * sigma);
    double values[3], sum = 0;
    for(int i = 0, x = 1 - n; i < n2; ++i, x += 2)
    {
        values[i] = std::exp((double)(x * x) * scale2x);
        sum += values[i];
    }
    sum = sum * 2 + 1;
    const double mul1 = 1.0 / sum;
    double err = 0;
    long total = 0;
    for(int i = 0; i < n2; ++i)
    {
        const double adj = values[i] * mul1 * 256.0 + err;
        const long v0 = lrint(adj);
        err = adj - (double)v0;
        taps[i] = taps[n - 1 - i] = (int)v0;
        total += 2 * v0;
    }
    taps[n2] = (int)(256 - total);
}

static bool umax_table(int* out16)
{
    // orb_impl ctor, distributed_cv_feature.cpp:522-541
    static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    int u[17] = {0};
    const int hp = 15;
    const unsigned vmax = (unsigned)std::floor(hp * std::sqrt(2.0) / 2 + 1);
    const unsigned vmin = (unsigned)std::ceil(hp * std::sqrt(2.0) / 2);
    for(unsigned v = 0; v <= vmax; ++v)
        u[v] = (int)std::round(std::sqrt((double)(hp * hp) - (double)(v * v)));
    for(unsigned v = hp, v0 = 0; vmin <= v; --v)
    {
        while(u[v0] == u[v0 + 1])
            ++v0;
        u[v] = (int)v0;
        ++v0;
    }
    std::memcpy(out16, u, sizeof(expect));
    return std::memcmp(u, expect, sizeof(expect)) == 0;
}

// Disc weights for the intensity centroid (ic_angle, :543-570): entry t = 8*row + column group covers
// the 4 pixels u = -15+4c .. -12+4c of row v = t/8 - 15; byte = u (first table) / v (second table) when
// the pixel lies in the radius-15 disc (|u| <= u_max[|v|]), else 0.
static void build_orient_weights(const int* umax, uint32_t* w /* [2][256] */)
{
    std::memset(w, 0, 2 * 256 * sizeof(uint32_t));
    for(int t = 0; t < 31 * 8; ++t)
    {
        const int v = t / 8 - 15, c = t % 8;
        for(int j = 0; j < 4; ++j)
        {
            const int u = -15 + 4 * c + j;
            const int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
            if(au <= 15 && au <= umax[av])
            {
                w[t] |= (uint32_t)(uint8_t)(int8_t)u << (8 * j);
                w[256 + t] |= (uint32_t)(uint8_t)(int8_t)v << (8 * j);
            }
        }
    }
}

static int build_geometry(mslam_hip_ctx* c)
{
    const mslam_hip_params& p = c->p;
    Geometry& g = c->geom;
    g.n_levels = p.n_levels;
    g.W = p.width;
    g.H = p.height;
    g.blur_tiled = 0;
    float scale[kMaxLevels];
    scale[0] = 1.0f;
    for(int l = 1; l < p.n_levels; ++l)
        scale[l] = p.scale_factor * scale[l - 1];
    unsigned offset = 0;
    c->cells.clear();
    for(int l = 0; l < p.n_levels; ++l)
    {
        LevelGeom& lv = g.lv[l];
        const double s = scale[l];
        lv.w = l == 0 ? p.width : (int)std::round(p.width * 1.0 / s);
        lv.h = l == 0 ? p.height : (int)std::round(p.height * 1.0 / s);
        lv.scale = scale[l];
        if(lv.w <= 2 * kBorder + kOverlap || lv.h <= 2 * kBorder + kOverlap)
            return fail(c, MSLAM_HIP_E_INVALID,
                        "pyramid level " + std::to_string(l) + " is smaller than the 19-px border + one cell");
        if(lv.w > 4095 + kBorder || lv.h > 4095 + kBorder)
            return fail(c, MSLAM_HIP_E_INVALID, "frames larger than 4096 px per side are not supported");
        lv.pitch = (lv.w + kTileW - 1) & ~(kTileW - 1); // whole tiles (common.hpp: tiled_off); kTileW is a multiple of 16
        lv.offset = (int)offset;
        offset += ((unsigned)lv.pitch * ((lv.h + kTileH - 1) & ~(kTileH - 1)) + 255u) & ~255u; // whole tile rows
        lv.bw = lv.w - 2 * kBorder;
        lv.bh = lv.h - 2 * kBorder;

        // cells (:866-905)
        const unsigned min_b = kBorder, max_bx = lv.w - kBorder, max_by = lv.h - kBorder;
        const unsigned num_cols = (max_bx - min_b) / kCell + 1, num_rows = (max_by - min_b) / kCell + 1;
        lv.cell_base = (int)c->cells.size();
        for(unsigned i = 0; i < num_rows; ++i)
        {
            const unsigned min_y = min_b + i * kCell;
            if(max_by - kOverlap <= min_y)
                continue;
            unsigned max_y = min_y + kCell + kOverlap;
            if(max_by < max_y)
                max_y = max_by;
            for(unsigned j = 0; j < num_cols; ++j)
            {
                const unsigned min_x = min_b + j * kCell;
                if(max_bx - kOverlap <= min_x)
                    continue;
                unsigned max_x = min_x + kCell + kOverlap;
                if(max_bx < max_x)
                    max_x = max_bx;
                CellDesc cd{};
                cd.level = (int16_t)l;
                cd.cw = (int16_t)(max_x - min_x);
                cd.ch = (int16_t)(max_y - min_y);
                cd.x0 = (int16_t)min_x;
                cd.y0 = (int16_t)min_y;
                cd.ox = (int16_t)(j * kCell);
                cd.oy = (int16_t)(i * kCell);
                c->cells.push_back(cd);
            }
        }
        lv.n_cells = (int)c->cells.size() - lv.cell_base;
        if(lv.n_cells > 2048)
            return fail(c, MSLAM_HIP_E_INVALID, "more than 2048 FAST cells on one level");

        lv.bsx = (lv.w + 3) / 4; // 4-column strips per row (k_blur2)

        // quadtree initial grid (:1031-1052) on the bordered rectangle [19, w-19) x [19, h-19)
        const int min_x = kBorder, max_x = lv.w - kBorder, min_y = kBorder, max_y = lv.h - kBorder;
        const double ratio = (double)(max_x - min_x) / (max_y - min_y);
        if(ratio > 1)
        {
            lv.nxg = (int)std::round(ratio);
            lv.nyg = 1;
            lv.delta_x = (double)(max_x - min_x) / lv.nxg;
            lv.delta_y = max_y - min_y;
        }
        else
        {
            lv.nxg = 1;
            lv.nyg = (int)std::round(1 / ratio);
            lv.delta_x = max_x - min_y; // sic (:1050)
            lv.delta_y = (double)(max_y - min_y) / lv.nyg;
        }
        if(lv.nxg * lv.nyg > 64)
            return fail(c, MSLAM_HIP_E_INVALID, "aspect ratio beyond 64:1 is not supported");
    }
    g.slab = offset + 256;
    g.n_cells = (int)c->cells.size();
    return MSLAM_HIP_OK;
}

// cv::ORB mode geometry (OpenCV orb.cpp): layerScale[l] = (float)pow(scaleFactor, l) with the double member holding
// the float argument; sizes cvRound(cols * (1.0f / scale)) in float; nfeaturesPerLevel as computeKeyPoints does.
static int build_geometry_cv(mslam_hip_ctx* c)
{
    const mslam_hip_params& p = c->p;
    Geometry& g = c->geom;
    g.n_levels = p.n_levels;
    g.W = p.width;
    g.H = p.height;
    g.blur_tiled = 0;
    const double sf = (double)p.scale_factor;
    unsigned offset = 0;
    c->cells.clear();
    for(int l = 0; l < p.n_levels; ++l)
    {
        LevelGeom& lv = g.lv[l];
        lv = LevelGeom{};
        lv.scale = (float)std::pow(sf, (double)l);
        const float inv = 1.0f / lv.scale;
        lv.w = cv_round_f((float)p.width * inv);
        lv.h = cv_round_f((float)p.height * inv);
        if(lv.w < 16 || lv.h < 16)
            return fail(c, MSLAM_HIP_E_INVALID, "pyramid level " + std::to_string(l) + " is smaller than 16 px");
        if(lv.w > 4095 || lv.h > 4095)
            return fail(c, MSLAM_HIP_E_INVALID, "frames larger than 4095 px per side are not supported");
        lv.pitch = (lv.w + kTileW - 1) & ~(kTileW - 1); // whole tiles (common.hpp: tiled_off); kTileW is a multiple of 16
        lv.offset = (int)offset;
        offset += ((unsigned)lv.pitch * ((lv.h + kTileH - 1) & ~(kTileH - 1)) + 255u) & ~255u; // whole tile rows
        lv.bw = lv.w - 2 * kBorder;
        lv.bh = lv.h - 2 * kBorder;
        lv.bsx = (lv.w + 3) / 4;
    }
    g.slab = offset + 256;
    g.n_cells = 0;
    const float factor = (float)(1.0 / sf);
    float desired = (float)p.n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.n_levels));
    int sum = 0;
    for(int l = 0; l < p.n_levels - 1; ++l)
    {
        c->cv_quota[l] = cv_round_f(desired);
        sum += c->cv_quota[l];
        desired *= factor;
    }
    c->cv_quota[p.n_levels - 1] = std::max(p.n_features - sum, 0);
    return MSLAM_HIP_OK;
}

// interpolationLinear<uchar>::getCoeffs (imgproc resize.cpp): offsets, 8.8 coefficients, interpolating range
static void exact_axis(int ssize, int dsize, std::vector<int32_t>& ofs, std::vector<uint32_t>& coef, int& dmin, int& dmax)
{
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    dmin = 0;
    dmax = dsize;
    for(int d = 0; d < dsize; ++d)
    {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        const int ival = (int)std::floor(fval);
        int o = 0;
        uint32_t c0 = 256, c1 = 0;
        if(ival >= 0 && ssize > 1)
        {
            if(ival < ssize - 1)
            {
                o = ival;
                c1 = (uint32_t)lrint((fval - (double)ival) * 256.0);
                c0 = 256 - c1;
            }
            else
            {
                // right of the range: the last sample (hlineResizeCn), stored as offset ssize-2 with weights 0 / 256 so that
                // the kernel's unconditional second read stays inside the row
                o = ssize - 2;
                c0 = 0, c1 = 256;
                dmax = std::min(dmax, d);
            }
        }
        else
            dmin = std::max(dmin, d + 1); // left of the range: the first sample = offset 0, weights 256 / 0 (the defaults)
        ofs.push_back(o);
        coef.push_back(c0 | (c1 << 16));
    }
}

template <typename T>
static hipError_t dmalloc(T*& p, size_t n)
{
    return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T));
}

// ---- C ABI ------------------------------------------------------------------------------------------
extern "C" {

void mslam_hip_default_params(mslam_hip_params* p)
{
    std::memset(p, 0, sizeof(*p));
    p->width = 640;
    p->height = 480;
    p->max_batch = 1;
    p->n_levels = 8;        // distributed_cv_feature.cpp:1184
    p->scale_factor = 1.2f; // :1184
    p->ini_fast_thr = 20;
    p->min_fast_thr = 7;
    p->min_node_area = 1000; // :1186
    p->max_keypoints = 8192;
    p->max_candidates = 16384;
    p->device = 0;
    p->stream = nullptr;
    p->detector = MSLAM_HIP_DETECTOR_DISTRIBUTED;
    p->n_features = 1000;    // orb_feature.cpp:25
    p->edge_threshold = 31;  // cv::ORB::create default
}

int mslam_hip_abi_version(void) { return MSLAM_HIP_ABI_VERSION; }

const char* mslam_hip_last_error(const mslam_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void mslam_hip_destroy(mslam_hip_ctx* c)
{
    if(!c)
        return;
    if(c->stream)
        (void)hipStreamSynchronize(c->stream);
    void* bufs[] = {c->d_cells,   c->d_rs_ofs, c->d_rs_coef, c->d_rs_qt, c->d_ratio_thr, c->d_orient_w, c->d_stage,  c->d_pyr,
                    c->d_blur,    c->d_cv_ofs, c->d_cv_coef, c->d_cell_cnt, c->d_cell_kp, c->quad.cand, c->quad.cand_cnt, c->quad.sel,
                    c->quad.sel_cnt, c->quad.kp_node, c->quad.nodes_a, c->quad.nodes_b, c->quad.ncnt_a, c->quad.ncnt_b,
                    c->quad.child_cnt, c->quad.ninfo, c->quad.best, c->d_flags, c->d_hm_from, c->d_hm_out, c->d_hm_partial,
                    c->d_xyz, c->d_valid, c->d_pnp_obj, c->d_pnp_img, c->d_pnp_n, c->d_pnp_counts, c->d_pnp_hyp, c->d_pnp_out,
                    c->d_pnp_mask, c->d_blur_waves, c->d_pnp1_obj, c->d_pnp1_img, c->d_pnp1_hyp, c->d_pnp1_out,
                    c->d_pnp1_counts, c->d_pnp1_mask};
    for(void* b : bufs)
        if(b)
            (void)hipFree(b);
    if(c->h_out)
        (void)hipHostFree(c->h_out);
    if(c->h_hm)
        (void)hipHostFree(c->h_hm);
    if(c->h_stage)
        (void)hipHostFree(c->h_stage);
    if(c->match_graph)
        (void)hipGraphExecDestroy(c->match_graph);
    for(auto& e : c->detect_graph)
        if(e)
            (void)hipGraphExecDestroy(e);
    if(c->stream_m)
        (void)hipStreamSynchronize(c->stream_m);
    for(auto& o : c->out)
    {
        void* ob[] = {o.xy, o.desc, o.octave, o.angle, o.response, o.count, o.idx0, o.idx1, o.dist0, o.dist1, o.mfrom, o.mto, o.mcount};
        for(void* b : ob)
            if(b)
                (void)hipFree(b);
        if(o.ev_detect)
            (void)hipEventDestroy(o.ev_detect);
        if(o.ev_match)
            (void)hipEventDestroy(o.ev_match);
    }
    if(c->stream_m)
        (void)hipStreamDestroy(c->stream_m);
    for(auto& t : c->timers)
    {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for(int k = 0; k < 4; ++k)
    {
        if(c->blur_stream[k])
        {
            (void)hipStreamSynchronize(c->blur_stream[k]);
            (void)hipStreamDestroy(c->blur_stream[k]);
        }
        if(c->ev_blur_fork[k])
            (void)hipEventDestroy(c->ev_blur_fork[k]);
        if(c->ev_blur_join[k])
            (void)hipEventDestroy(c->ev_blur_join[k]);
    }
    if(c->ev_fork)
        (void)hipEventDestroy(c->ev_fork);
    for(int k = 0; k < 4; ++k)
    {
        if(c->ev_join[k])
            (void)hipEventDestroy(c->ev_join[k]);
        if(c->side[k])
        {
            (void)hipStreamSynchronize(c->side[k]);
            (void)hipStreamDestroy(c->side[k]);
        }
    }
    if(c->bow)
        bow_destroy(c->bow);
    if(c->own_stream && c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

static void select_set(mslam_hip_ctx* c, int k)
{
    const mslam_out_set& o = c->out[k];
    c->cur = k;
    c->d_xy = o.xy, c->d_desc = o.desc, c->d_octave = o.octave, c->d_angle = o.angle, c->d_response = o.response;
    c->d_count = o.count, c->d_idx0 = o.idx0, c->d_idx1 = o.idx1, c->d_dist0 = o.dist0, c->d_dist1 = o.dist1;
    c->d_mfrom = o.mfrom, c->d_mto = o.mto, c->d_mcount = o.mcount;
}

// Quad table of one level for k_resize_col (both interpolation flavours): every 4 destination pixels share one aligned
// 12-byte source window; per quad {window byte offset, upper-pair flags, 4 v_perm selectors, 4 packed weight pairs}.
// (The last quad's window may run up to 8 bytes past its row: into the next row, or into the >= 256 bytes of padding
// behind every level of the slab.)  Leaves `start` at SIZE_MAX when a quad does not fit the window (very large scale factors) or the batch is beyond the
// kernel's exact index split.
static void build_quad_table(const std::vector<int32_t>& ofs, const std::vector<uint32_t>& coef, size_t x0, int dw,
                             int max_batch, std::vector<uint4>& qt, size_t& start_out, int& need_out)
{
    const int nq = (dw + 3) / 4;
    const size_t start = qt.size() / 3;
    bool ok = true;
    int need = 0;
    for(int q = 0; q < nq && ok; ++q)
    {
        const int first = ofs[x0 + 4 * q];
        const uint32_t base = (uint32_t)first & ~3u;
        uint32_t sel[4], cf[4], flags = 0;
        for(int k = 0; k < 4; ++k)
        {
            const int dx = std::min(4 * q + k, dw - 1);
            const int shift = ofs[x0 + dx] - (int)base;
            const uint32_t a0 = coef[x0 + dx] & 0xFFFF, a1 = coef[x0 + dx] >> 16;
            if(shift < 0 || shift > 10 || a0 > 0xFFF || a1 > 0xFFF)
                ok = false;
            // bytes shift, shift+1 of the 12-byte window: in dwords (0,1) when shift <= 6, else in (1,2)
            const int upper = shift > 6 ? 1 : 0;
            flags |= (uint32_t)upper << k;
            // selector bytes [i0, zero, i0+1, zero]: the pair lands as two u16 lanes
            sel[k] = 0x0c010c00u + (uint32_t)(shift - 4 * upper) * 0x00010001u;
            cf[k] = a0 | (a1 << 16);
        }
        need |= (int)flags;
        qt.push_back(make_uint4(base, flags, sel[0], sel[1]));
        qt.push_back(make_uint4(sel[2], sel[3], cf[0], cf[1]));
        qt.push_back(make_uint4(cf[2], cf[3], 0, 0));
    }
    if(!ok || (size_t)max_batch * nq >= (1u << 22)) // (beyond the exact range of the kernel's float index split)
    {
        qt.resize(start * 3);
        start_out = SIZE_MAX;
        return;
    }
    start_out = start;
    need_out = need;
}

static int create_impl(mslam_hip_ctx* c)
{
    const mslam_hip_params& p = c->p;
    const bool has_detector = !(p.width == 0 && p.height == 0); // 0 x 0: matcher / BoW only, no pyramid buffers
    if((has_detector && (p.width <= 0 || p.height <= 0)) || p.max_batch < 1 || p.n_levels < 1 || p.n_levels > kMaxLevels ||
       !(p.scale_factor > 1.0f) || p.ini_fast_thr < 0 || p.ini_fast_thr > 255 || p.min_fast_thr < 0 ||
       p.min_fast_thr > p.ini_fast_thr || p.max_keypoints < 1 || p.max_keypoints > 65535 || p.max_candidates < 1 ||
       p.max_candidates > (1 << 22))
        return fail(c, MSLAM_HIP_E_INVALID, "invalid parameters");
    const bool cv_mode = p.detector == MSLAM_HIP_DETECTOR_CV_ORB;
    if((p.detector != MSLAM_HIP_DETECTOR_DISTRIBUTED && !cv_mode) ||
       (cv_mode && (p.n_features < 0 || p.edge_threshold < 19 || p.edge_threshold > 1024)))
        return fail(c, MSLAM_HIP_E_INVALID, "invalid detector parameters (cv::ORB mode needs edge_threshold >= 19: "
                                            "the descriptor pattern reaches 19 px)");
    int umax[16];
    if(!umax_table(umax))
        return fail(c, MSLAM_HIP_E_INVALID, "u_max table self-check failed");
    int taps[7];
    gaussian_taps_fixed(taps);
    static const int expect_taps[7] = {18, 34, 48, 56, 48, 34, 18};
    if(std::memcmp(taps, expect_taps, sizeof(taps)) != 0)
        return fail(c, MSLAM_HIP_E_INVALID, "gaussian tap self-check failed");
    set_blur_taps(taps);

    int rc = !has_detector ? MSLAM_HIP_OK : cv_mode ? build_geometry_cv(c) : build_geometry(c);
    if(rc)
        return rc;
    const Geometry& g = c->geom;

    int n_dev = 0;
    if(hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(c, MSLAM_HIP_E_RUNTIME, "no HIP device available (the product path has no CPU fallback)");
    HIPCHK(c, hipSetDevice(p.device));
    if(p.stream)
        c->stream = (hipStream_t)p.stream;
    else
    {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }

    {
        if(const char* ge = getenv("MSLAM_HIP_GRAPH"))
            c->use_graph = atoi(ge) != 0;
        const char* e = getenv("MSLAM_HIP_STREAMS");
        c->n_side = e ? std::max(1, std::min(atoi(e), 4)) : 2; // measured in one run (batch 250): 1 / 2 / 3 / 4 chunks = 366 / 375 / 369 / 361 M kp/s
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for(int k = 0; k < c->n_side; ++k)
        {
            HIPCHK(c, hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming));
        }
        const char* fb = getenv("MSLAM_HIP_FORK_BLUR");
        c->fork_blur = fb && atoi(fb) != 0;
        // (only when asked for: every stream of a process is dealt to one of a few hardware queues, and streams that share a
        // queue run one after the other — idle streams here can push a later context's chunk / matcher streams onto one queue)
        for(int k = 0; k < 4 && c->fork_blur; ++k)
        {
            HIPCHK(c, hipStreamCreateWithFlags(&c->blur_stream[k], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_blur_fork[k], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_blur_join[k], hipEventDisableTiming));
        }
    }
    // tables
    if(has_detector && cv_mode)
    {
        std::vector<int32_t> ofs;
        std::vector<uint32_t> coef;
        c->cv_x.assign(p.n_levels, 0);
        c->cv_y.assign(p.n_levels, 0);
        c->cv_range.assign(p.n_levels, std::array<int, 4>{0, 0, 0, 0});
        c->cv_window12.assign(p.n_levels, 0);
        for(int l = 1; l < p.n_levels; ++l)
        {
            c->cv_x[l] = ofs.size();
            exact_axis(g.lv[l - 1].w, g.lv[l].w, ofs, coef, c->cv_range[l][0], c->cv_range[l][1]);
            c->cv_y[l] = ofs.size();
            exact_axis(g.lv[l - 1].h, g.lv[l].h, ofs, coef, c->cv_range[l][2], c->cv_range[l][3]);
            // the window form needs, for every destination quad, o3 + 1 - (o0 & ~3) <= 11, the window inside the pitch and
            // the kernel's unconditional "next row" read inside the level
            bool ok = true;
            const int dw = g.lv[l].w;
            for(int q = 0; q * 4 < dw && ok; ++q)
            {
                const int base = ofs[c->cv_x[l] + 4 * q] & ~3;
                for(int k = 0; k < 4; ++k)
                {
                    const int o = ofs[c->cv_x[l] + std::min(4 * q + k, dw - 1)];
                    ok = ok && o >= base && o + 1 - base <= 11;
                }
                ok = ok && base + 12 <= g.lv[l - 1].pitch;
            }
            c->cv_window12[l] = ok ? 1 : 0;
        }
        // the column-walking kernel of the in-tree detector serves INTER_LINEAR_EXACT too (k_resize_col<true>); levels
        // whose quads do not fit its 12-byte window fall back to k_resize_exact
        {
            std::vector<uint4> qt;
            c->rs_q.assign(p.n_levels, SIZE_MAX);
            c->rs_need.assign(p.n_levels, 0);
            for(int l = 1; l < p.n_levels; ++l)
                build_quad_table(ofs, coef, c->cv_x[l], g.lv[l].w, p.max_batch, qt, c->rs_q[l], c->rs_need[l]);
            qt.push_back(make_uint4(0, 0, 0, 0));
            HIPCHK(c, dmalloc(c->d_rs_qt, qt.size()));
            HIPCHK(c, hipMemcpy(c->d_rs_qt, qt.data(), qt.size() * 16, hipMemcpyHostToDevice));
        }
        ofs.push_back(0);
        coef.push_back(0);
        HIPCHK(c, dmalloc(c->d_cv_ofs, ofs.size()));
        HIPCHK(c, dmalloc(c->d_cv_coef, coef.size()));
        HIPCHK(c, hipMemcpy(c->d_cv_ofs, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_cv_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    }
    if(has_detector && !cv_mode)
    {
    HIPCHK(c, dmalloc(c->d_cells, c->cells.size()));
    HIPCHK(c, hipMemcpy(c->d_cells, c->cells.data(), c->cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> ofs;
        std::vector<uint32_t> coef;
        c->rs_x.assign(p.n_levels, 0);
        c->rs_y.assign(p.n_levels, 0);
        for(int l = 1; l < p.n_levels; ++l)
        {
            c->rs_x[l] = ofs.size();
            resize_table(g.lv[l - 1].w, g.lv[l].w, true, ofs, coef);
            c->rs_y[l] = ofs.size();
            resize_table(g.lv[l - 1].h, g.lv[l].h, false, ofs, coef);
        }
        // quad tables for k_resize_col: 4 destination pixels share one aligned 12-byte source window
        std::vector<uint4> qt;
        c->rs_q.assign(p.n_levels, SIZE_MAX);
        c->rs_need.assign(p.n_levels, 0);
        for(int l = 1; l < p.n_levels; ++l)
            build_quad_table(ofs, coef, c->rs_x[l], g.lv[l].w, p.max_batch, qt, c->rs_q[l], c->rs_need[l]);
        qt.push_back(make_uint4(0, 0, 0, 0));
        HIPCHK(c, dmalloc(c->d_rs_qt, qt.size()));
        HIPCHK(c, hipMemcpy(c->d_rs_qt, qt.data(), qt.size() * 16, hipMemcpyHostToDevice));
        ofs.push_back(0);
        coef.push_back(0);
        HIPCHK(c, dmalloc(c->d_rs_ofs, ofs.size()));
        HIPCHK(c, dmalloc(c->d_rs_coef, coef.size()));
        HIPCHK(c, hipMemcpy(c->d_rs_ofs, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_rs_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    }
    } // has_detector
    HIPCHK(c, dmalloc(c->d_ratio_thr, 257));
    {
        uint32_t w[2 * 256];
        build_orient_weights(umax, w);
        HIPCHK(c, dmalloc(c->d_orient_w, 2 * 256));
        HIPCHK(c, hipMemcpy(c->d_orient_w, w, sizeof(w), hipMemcpyHostToDevice));
    }

    const size_t B = (size_t)p.max_batch, L = (size_t)p.n_levels, cap = (size_t)p.max_candidates;
    const size_t K = (size_t)p.max_keypoints;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_out), 16 + K * 52, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_out), c->h_out, 0));
    HIPCHK(c, dmalloc(c->d_flags, 1));
    HIPCHK(c, hipMemset(c->d_flags, 0, 4));
    if(has_detector)
    {
    {
        // the FAST score kernels order f16 denormal bit patterns (arc_score.hpp): checked once, on the device, in this build
        uint32_t ok = 0;
        launch_denorm_selfcheck(c->d_flags, c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&ok, c->d_flags, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemset(c->d_flags, 0, 4));
        if(ok != 1u)
            return fail(c, MSLAM_HIP_E_RUNTIME, "v_pk_minimum3_f16 / v_pk_maximum3_f16 do not preserve f16 denormals in this build "
                                                "(the FAST score kernels need .amdhsa_float_denorm_mode_16_64 3): refusing to run");
    }
    HIPCHK(c, dmalloc(c->d_stage, (size_t)p.width * p.height * 3));
    // + 64: the patch loads of k_describe may run a few bytes past the last row of the last frame
    HIPCHK(c, dmalloc(c->d_pyr, B * g.slab + 256));
    HIPCHK(c, dmalloc(c->d_blur, B * g.slab + 256));
    {
        // k_level.hip (fused gray + blur): needs dword columns, 32-bit batch offsets and the exact float index split
        {
            const char* e = getenv("MSLAM_HIP_FUSED_LEVELS");
            const int want = e ? atoi(e) : kMaxLevels;
            const char* k = getenv("MSLAM_HIP_LEVEL_K6");
            const char* ks = getenv("MSLAM_HIP_LEVEL_K6_SMALL");
            c->level_k6_small = ks ? std::max(0, atoi(ks)) : 0; // a handful of frames: 2-row blocks (a wave walks 8 rows instead of 14)
            {
                // the one-launch level chain (k_level_chain; off by default: measured equal in place, 4 % slower alone — DESIGN.md §8.1):
                // MSLAM_HIP_LEVEL_CHAIN=1, frames per workgroup, waves per workgroup (4 or 8), longest row block
                const char* lc = getenv("MSLAM_HIP_LEVEL_CHAIN");
                const char* lf = getenv("MSLAM_HIP_LEVEL_CHAIN_FRAMES");
                const char* lw = getenv("MSLAM_HIP_LEVEL_CHAIN_WAVES");
                const char* lk = getenv("MSLAM_HIP_LEVEL_CHAIN_K6");
                c->level_chain = lc ? atoi(lc) : 0;
                c->level_chain_frames = lf ? std::max(1, std::min(64, atoi(lf))) : 2;
                c->level_chain_waves = lw ? atoi(lw) : 8;
                if(c->level_chain_waves != 4)
                    c->level_chain_waves = 8;
                c->level_chain_k6 = lk ? std::max(1, std::min(9, atoi(lk))) : 9;
            }
            c->level_k6 = k ? std::max(1, atoi(k)) : 9; // 9 -> 56-row blocks: the halo re-reads cost 11 % instead of 19 % (32 rows); the step time is the same
            const bool fits = (p.width & 3) == 0 && (double)B * p.width * p.height * 3 < 4294967296.0 &&
                              (double)B * g.slab < 4294901760.0 /* below k_level.hip's kDropLane */ && (size_t)B * (p.width / 4) < (1u << 22) && p.height >= 8;
            int n_fused = fits ? 1 : 0;
            // levels l > 0 (resize + blur): the level must be on the 12-byte-window tables of k_resize_col
            while(n_fused >= 1 && n_fused < p.n_levels && c->rs_q[n_fused] != SIZE_MAX && g.lv[n_fused].h >= 8 &&
                  (size_t)B * ((g.lv[n_fused].w + 3) / 4) < (1u << 22))
                ++n_fused;
            c->fused_levels = std::max(0, std::min(want, n_fused));
            // every level from k_level.hip: the blurred slab (written there, read by k_describe only) is kept in tiles (common.hpp)
            const char* t = getenv("MSLAM_HIP_TILED_BLUR");
            c->geom.blur_tiled = c->fused_levels == p.n_levels && !(t && atoi(t) == 0) ? 1 : 0;
        }
        std::vector<BlurWave> bw;
        build_blur_waves(g, c->fused_levels, bw);
        c->blur_wpf = (int)bw.size();
        if(!bw.empty())
        {
            HIPCHK(c, dmalloc(c->d_blur_waves, bw.size()));
            HIPCHK(c, hipMemcpy(c->d_blur_waves, bw.data(), bw.size() * sizeof(BlurWave), hipMemcpyHostToDevice));
        }
    }
    HIPCHK(c, dmalloc(c->d_cell_cnt, B * g.n_cells));
    HIPCHK(c, dmalloc(c->d_cell_kp, B * g.n_cells * (size_t)kCellCap));
    QuadArgs& q = c->quad;
    HIPCHK(c, dmalloc(q.cand, B * L * cap));
    HIPCHK(c, dmalloc(q.cand_cnt, B * L));
    HIPCHK(c, dmalloc(q.sel, B * L * cap));
    HIPCHK(c, dmalloc(q.sel_cnt, B * L));
    HIPCHK(c, dmalloc(q.kp_node, B * L * cap));
    HIPCHK(c, dmalloc(q.nodes_a, B * L * cap));
    HIPCHK(c, dmalloc(q.nodes_b, B * L * cap));
    HIPCHK(c, dmalloc(q.ncnt_a, B * L * cap));
    HIPCHK(c, dmalloc(q.ncnt_b, B * L * cap));
    HIPCHK(c, dmalloc(q.child_cnt, B * L * cap * 4));
    HIPCHK(c, dmalloc(q.ninfo, B * L * cap));
    HIPCHK(c, dmalloc(q.best, B * L * cap));
    q.cell_cnt = c->d_cell_cnt;
    q.cell_kp = c->d_cell_kp;
    q.flags = c->d_flags;
    q.cand_cap = p.max_candidates;
    q.min_size = p.min_node_area;
    } // has_detector

    for(auto& o : c->out)
    {
        HIPCHK(c, dmalloc(o.xy, (B + 1) * K * 2));
        HIPCHK(c, dmalloc(o.desc, (B + 1) * K * 32));
        HIPCHK(c, dmalloc(o.octave, (B + 1) * K));
        HIPCHK(c, dmalloc(o.angle, (B + 1) * K));
        HIPCHK(c, dmalloc(o.response, (B + 1) * K));
        HIPCHK(c, dmalloc(o.count, B + 1));
        HIPCHK(c, hipMemset(o.count, 0, (B + 1) * 4));
        HIPCHK(c, dmalloc(o.idx0, B * K));
        HIPCHK(c, dmalloc(o.idx1, B * K));
        HIPCHK(c, dmalloc(o.dist0, B * K));
        HIPCHK(c, dmalloc(o.dist1, B * K));
        HIPCHK(c, dmalloc(o.mfrom, B * K));
        HIPCHK(c, dmalloc(o.mto, B * K));
        HIPCHK(c, dmalloc(o.mcount, B));
        HIPCHK(c, hipMemset(o.mcount, 0, B * 4));
        HIPCHK(c, hipEventCreateWithFlags(&o.ev_detect, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&o.ev_match, hipEventDisableTiming));
    }
    // (the chunk streams and this one should sit on different hardware queues: streams that share a queue run one after the
    // other.  Giving them different PRIORITY classes to force that was measured — cfg5 380 -> 308 M, a second 640x480 context
    // 625 -> 491 M keypoints/s: the low class starves — and is not done.)
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream_m, hipStreamNonBlocking));
    {
        const char* e = getenv("MSLAM_HIP_OVERLAP_MATCH");
        c->overlap_match = !(e && atoi(e) == 0);
        // knobs of the synchronous single-frame calls: read per context, at creation (a test or a tool that sets them before
        // creating its context gets what it asked for)
        auto on = [](const char* name) { const char* v = getenv(name); return !v || atoi(v) != 0; };
        c->knob_mirror_results = on("MSLAM_HIP_MIRROR_RESULTS");
        c->knob_zero_copy = on("MSLAM_HIP_ZERO_COPY_FRAME");
        c->knob_match_graph = on("MSLAM_HIP_MATCH_GRAPH");
        const char* zm = getenv("MSLAM_HIP_ZERO_COPY_MAX_BYTES");
        // measured (round 6, tools/latency.py, detect median in us, zero-copy / copy): 640x480 146 / 162, 1280x720 374 / 302,
        // 1920x1080 818 / 620 — the kernel's PCIe reads (39 GB/s) beat the DMA's fixed cost only for small frames
        c->zero_copy_max_bytes = zm ? (size_t)atoll(zm) : (size_t)1200000;
        const char* m = getenv("MSLAM_HIP_MATCHER");
        c->matcher_kind = (m && std::strcmp(m, "popcount") == 0) ? MSLAM_HIP_MATCHER_POPCOUNT : MSLAM_HIP_MATCHER_AUTO;
    }
    select_set(c, 0);
    HIPCHK(c, hipDeviceSynchronize());
    return MSLAM_HIP_OK;
}

int mslam_hip_create(const mslam_hip_params* p, mslam_hip_ctx** out)
{
    if(!p || !out)
    {
        g_create_error = "null argument";
        return MSLAM_HIP_E_INVALID;
    }
    *out = nullptr;
    mslam_hip_ctx* c = new mslam_hip_ctx();
    c->p = *p;
    const int rc = create_impl(c);
    if(rc != MSLAM_HIP_OK)
    {
        g_create_error = c->err;
        mslam_hip_destroy(c);
        return rc;
    }
    *out = c;
    return MSLAM_HIP_OK;
}

constexpr int kMinChunk = 8; // do not cut batches into chunks smaller than this many frames

static int check_flags(mslam_hip_ctx* c)
{
    uint32_t f = 0;
    HIPCHK(c, hipStreamSynchronize(c->stream_m));
    HIPCHK(c, hipMemcpyAsync(&f, c->d_flags, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if(f == 0)
        return MSLAM_HIP_OK;
    HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 4, c->stream));
    std::string m = "capacity exceeded:";
    if(f & kFlagCandOverflow)
        m += " FAST candidates on a level > max_candidates;";
    if(f & kFlagKpOverflow)
        m += " keypoints in a frame > max_keypoints;";
    if(f & kFlagQuadNoConverge)
        m += " quadtree pass limit;";
    if(f & kFlagDbFull)
        m += " the BoW database is full (mslam_hip_bow_db_reserve);";
    if(f & kFlagBowPackOverflow)
        m += " a BoW vector has more words than the exchange format's k_max;";
    if(f & kFlagPackOverflow)
        m += " packed batch results larger than the buffer given to mslam_hip_pack_batch_dev (header.bytes);";
    return fail(c, MSLAM_HIP_E_CAPACITY, m);
}

int mslam_hip_sync(mslam_hip_ctx* c)
{
    ENTER(c);
    return check_flags(c);
}

// The single-frame call's results go back to the host in ONE step: this kernel writes the count, the status flags and
// exactly `count` records of every output array into the page-locked, device-mapped staging block (h_out) — no copy
// nodes in the graph (seven capacity-sized device-to-host copies cost 37 us of the 185 us call), nothing beyond the
// keypoints that exist crosses PCIe.
__global__ __launch_bounds__(256) void k_pack_results(const float* __restrict__ xy, const uint8_t* __restrict__ desc,
                                                      const int32_t* __restrict__ octave, const float* __restrict__ angle,
                                                      const float* __restrict__ response, const int32_t* __restrict__ count,
                                                      const uint32_t* __restrict__ flags, uint8_t* __restrict__ h, int K)
{
    const int n = min(max(*count, 0), K);
    const int tid = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    if(tid == 0)
    {
        reinterpret_cast<int32_t*>(h)[0] = *count;
        reinterpret_cast<uint32_t*>(h)[1] = *flags;
    }
    uint2* h_xy = reinterpret_cast<uint2*>(h + 16);
    uint2* h_desc = reinterpret_cast<uint2*>(h + 16 + (size_t)K * 8); // (8-byte aligned for every K; 16-byte only for even K)
    uint32_t* h_oct = reinterpret_cast<uint32_t*>(h + 16 + (size_t)K * 40);
    uint32_t* h_ang = h_oct + K;
    uint32_t* h_resp = h_ang + K;
    for(int i = tid; i < 4 * n; i += nt)
        h_desc[i] = reinterpret_cast<const uint2*>(desc)[i];
    for(int i = tid; i < n; i += nt)
    {
        h_xy[i] = reinterpret_cast<const uint2*>(xy)[i];
        h_oct[i] = reinterpret_cast<const uint32_t*>(octave)[i];
        h_ang[i] = reinterpret_cast<const uint32_t*>(angle)[i];
        h_resp[i] = reinterpret_cast<const uint32_t*>(response)[i];
    }
}

// slot 0 of the next output set = the last frame of the previous batch (its descriptors are the matcher's train side)
__global__ __launch_bounds__(256) void k_carry_prev(uint4* __restrict__ dst, const uint4* __restrict__ src, int32_t* __restrict__ dst_count,
                                                    const int32_t* __restrict__ src_count, int K)
{
    const int n = min(max(*src_count, 0), K);
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if(tid == 0)
        *dst_count = *src_count;
    for(int i = tid; i < 2 * n; i += gridDim.x * 256)
        dst[i] = src[i];
}

// one level l > 0 produced and blurred in one pass (k_level.hip)
static void enqueue_resize_blur(mslam_hip_ctx* c, int l, const int32_t* yofs, const uint32_t* ycoef, int exact, int f0, int nf,
                                int k6, hipStream_t cs, ResizeBlurArgs* chain_out = nullptr)
{
    const Geometry& g = c->geom;
    const LevelGeom &sl = g.lv[l - 1], &dl = g.lv[l];
    ResizeBlurArgs ra{};
    ra.pyr = c->d_pyr;
    ra.blur = c->d_blur;
    ra.slab = g.slab;
    ra.src_off = sl.offset, ra.sh = sl.h, ra.spitch = sl.pitch;
    ra.dst_off = dl.offset, ra.dw = dl.w, ra.dh = dl.h, ra.dpitch = dl.pitch;
    ra.qt = c->d_rs_qt + 3 * c->rs_q[l];
    ra.yofs = yofs;
    ra.ycoef = ycoef;
    ra.frame0 = f0, ra.n_frames = nf;
    ra.quads = (dl.w + 3) / 4;
    ra.inv_quads = 1.0f / (float)ra.quads;
    // (k6 = 0, the single-frame launches: 2-row blocks pay on the smaller levels only — measured per level, 640x480 pyramid:
    // 333 rows and below 6.0-8.0 -> 4.7-7.6 us, 400 rows and above 7.9-8.2 -> 9.6-10.3 us)
    if(k6 == 0 && dl.h >= 380)
        k6 = 1;
    ra.k6 = std::max(k6 == 0 ? 0 : 1, std::min(std::min(k6, 9), (dl.h - 2) / 6));
    // small levels: a launch of long row blocks is a single wave per SIMD or less and runs as long as ONE wave's walk.
    // Shorten the blocks (more halo rows, more waves) until the launch has about two waves per SIMD.
    {
        const long waves_x = ((long)nf * ra.quads + 61) / 62;
        while(ra.k6 > 2 && waves_x * ((dl.h + 6 * ra.k6 + 1) / (6 * ra.k6 + 2)) < 2048)
            --ra.k6;
    }
    ra.need_mask = c->rs_need[l];
    ra.exact = exact;
    ra.blur_tiled = g.blur_tiled;
    ra.always_load = nf < 8 ? 1 : 0;
    ra.bk = make_blur_k();
    if(chain_out) // a level of the one-launch chain (k_level_chain picks its own rows per block)
    {
        *chain_out = ra;
        return;
    }
    launch_resize_blur(ra, cs);
}

// the kernel sequence of one detect batch into the current output set (no set switching, no events)
static int enqueue_detect(mslam_hip_ctx* c, const uint8_t* d_bgr, int n_frames)
{
    const Geometry& g = c->geom;
    const size_t K = (size_t)c->p.max_keypoints;
    hipStream_t s = c->stream;
    // rows per block of the level kernels (6 k6 + 2): long blocks for throughput (fewer halo rows), short ones for a
    // handful of frames, where a launch is a few waves and its duration is the length of one wave's walk
    const int k6_batch = n_frames >= 8 ? c->level_k6 : c->level_k6_small;
    // Frames are independent until the matcher, so the batch is cut into chunks that run the same
    // kernel sequence on separate HIP streams: the latency-bound kernels of one chunk (quadtree, the
    // tails of every launch) overlap the throughput-bound kernels of the other.  With profiling on,
    // everything runs on the context's stream so that the per-stage events are meaningful.
    const int n_chunks = (c->profiling || n_frames < 2 * kMinChunk) ? 1 : std::min<int>(c->n_side, n_frames / kMinChunk);
    if(n_chunks > 1)
        HIPCHK(c, hipEventRecord(c->ev_fork, s));
    for(int k = 0; k < n_chunks; ++k)
    {
        const int f0 = (int)((long long)n_frames * k / n_chunks), f1 = (int)((long long)n_frames * (k + 1) / n_chunks);
        const int nf = f1 - f0;
        hipStream_t cs = n_chunks > 1 ? c->side[k] : s;
        if(n_chunks > 1)
            HIPCHK(c, hipStreamWaitEvent(cs, c->ev_fork, 0));
        const bool cv_mode = c->p.detector == MSLAM_HIP_DETECTOR_CV_ORB;
        // gray + every level of the chunk in ONE launch (k_level_chain): batches whose levels all come from k_level.hip
        bool chained = false;
        if(c->level_chain && !cv_mode && nf >= 8 && c->fused_levels == g.n_levels && g.n_levels <= kChainLevels && g.blur_tiled)
        {
            StageScope t(c, "levels", cs);
            GrayBlurArgs ga{};
            ga.bgr = d_bgr;
            ga.pyr = c->d_pyr;
            ga.blur = c->d_blur;
            ga.W = g.W, ga.H = g.H, ga.pitch = g.lv[0].pitch;
            ga.slab = g.slab;
            ga.n_frames = nf, ga.frame0 = f0;
            ga.quads = g.W / 4;
            ga.inv_quads = 1.0f / (float)ga.quads;
            ga.k6 = 1;
            ga.blur_tiled = g.blur_tiled;
            ga.bk = make_blur_k();
            ResizeBlurArgs lv[kChainLevels - 1];
            for(int l = 1; l < g.n_levels; ++l)
                enqueue_resize_blur(c, l, c->d_rs_ofs + c->rs_y[l], c->d_rs_coef + c->rs_y[l], 0, f0, nf, k6_batch, cs, &lv[l - 1]);
            chained = launch_level_chain(ga, lv, g.n_levels - 1, c->level_chain_frames, c->level_chain_waves, c->level_chain_k6, cs);
        }
        if(!chained)
        {
            StageScope t(c, "gray", cs);
            if(c->fused_levels >= 1)
            {
                GrayBlurArgs ga{};
                ga.bgr = d_bgr;
                ga.pyr = c->d_pyr;  // (level 0 starts the slab)
                ga.blur = c->d_blur;
                ga.W = g.W, ga.H = g.H, ga.pitch = g.lv[0].pitch;
                ga.slab = g.slab;
                ga.n_frames = nf, ga.frame0 = f0;
                ga.quads = g.W / 4;
                ga.inv_quads = 1.0f / (float)ga.quads;
                ga.k6 = std::max(k6_batch == 0 && g.H < 380 ? 0 : 1, std::min(k6_batch, (g.H - 2) / 6));
                ga.blur_tiled = g.blur_tiled;
                ga.bk = make_blur_k();
                launch_gray_blur(ga, cs);
            }
            else
                launch_gray(d_bgr, c->d_pyr, g, f0, nf, cs);
        }
        if(cv_mode)
        {
            // OrbOpenCvDetector (orb_feature.cpp:33-65 -> OpenCV orb.cpp detectAndCompute), see k_cvorb.hip
            {
                StageScope t(c, "resize", cs);
                for(int l = 1; l < g.n_levels; ++l)
                {
                    const LevelGeom &sl = g.lv[l - 1], &dl = g.lv[l];
                    if(l < c->fused_levels)
                    {
                        enqueue_resize_blur(c, l, c->d_cv_ofs + c->cv_y[l], c->d_cv_coef + c->cv_y[l], 1, f0, nf, k6_batch, cs);
                        continue;
                    }
                    if(c->rs_q[l] != SIZE_MAX)
                    {
                        ResizeColArgs ca{};
                        ca.pyr = c->d_pyr;
                        ca.slab = g.slab;
                        ca.src_off = sl.offset, ca.sh = sl.h, ca.spitch = sl.pitch;
                        ca.dst_off = dl.offset, ca.dw = dl.w, ca.dh = dl.h, ca.dpitch = dl.pitch;
                        ca.qt = c->d_rs_qt + 3 * c->rs_q[l];
                        ca.yofs = c->d_cv_ofs + c->cv_y[l];
                        ca.ycoef = c->d_cv_coef + c->cv_y[l];
                        ca.frame0 = f0, ca.n_frames = nf;
                        ca.quads = (dl.w + 3) / 4;
                        ca.inv_quads = 1.0f / (float)ca.quads;
                        ca.R = 8;
                        ca.need_mask = c->rs_need[l];
                        ca.exact = 1;
                        launch_resize_col(ca, cs);
                        continue;
                    }
                    ExactResizeArgs ra{};
                    ra.pyr = c->d_pyr;
                    ra.slab = g.slab;
                    ra.src_off = sl.offset, ra.sw = sl.w, ra.sh = sl.h, ra.spitch = sl.pitch;
                    ra.dst_off = dl.offset, ra.dw = dl.w, ra.dh = dl.h, ra.dpitch = dl.pitch;
                    ra.xofs = c->d_cv_ofs + c->cv_x[l];
                    ra.xcoef = c->d_cv_coef + c->cv_x[l];
                    ra.yofs = c->d_cv_ofs + c->cv_y[l];
                    ra.ycoef = c->d_cv_coef + c->cv_y[l];
                    ra.xmin = c->cv_range[l][0], ra.xmax = c->cv_range[l][1];
                    ra.ymin = c->cv_range[l][2], ra.ymax = c->cv_range[l][3];
                    ra.window12 = c->cv_window12[l];
                    ra.frame0 = f0;
                    launch_resize_exact(ra, nf, cs);
                }
            }
            CvSelectArgs sa{};
            sa.cand = c->quad.cand, sa.cand_cnt = c->quad.cand_cnt;
            sa.tmp_kp = c->quad.kp_node, sa.tmp_resp = reinterpret_cast<float*>(c->quad.ninfo);
            sa.sel = c->quad.sel, sa.sel_resp = reinterpret_cast<float*>(c->quad.best), sa.sel_cnt = c->quad.sel_cnt;
            sa.flags = c->d_flags;
            sa.cand_cap = c->p.max_candidates;
            sa.edge = c->p.edge_threshold;
            sa.std_order = c->cv_order == MSLAM_HIP_CV_ORDER_LIBSTDCXX ? 1 : 0;
            for(int l = 0; l < g.n_levels; ++l)
                sa.quota[l] = c->cv_quota[l];
            {
                StageScope t(c, "fast", cs);
                // the tiles of a level append to its list with atomics: counts start at zero
                launch_zero_u32(c->quad.cand_cnt + (size_t)f0 * g.n_levels, nf * g.n_levels, cs);
                launch_fast_tiles(c->d_pyr, g, c->p.ini_fast_thr, sa, f0, nf, cs);
            }
            {
                StageScope t(c, "select", cs);
                launch_cv_select(c->d_pyr, g, sa, f0, nf, cs);
            }
        }
        else
        {
        if(!chained)
        {
            StageScope t(c, "resize", cs);
            for(int l = 1; l < g.n_levels; ++l)
            {
                if(l < c->fused_levels)
                {
                    enqueue_resize_blur(c, l, c->d_rs_ofs + c->rs_y[l], c->d_rs_coef + c->rs_y[l], 0, f0, nf, k6_batch, cs);
                    continue;
                }
                if(c->rs_q[l] != SIZE_MAX)
                {
                    const LevelGeom &sl = g.lv[l - 1], &dl = g.lv[l];
                    ResizeColArgs ra{};
                    ra.pyr = c->d_pyr;
                    ra.slab = g.slab;
                    ra.src_off = sl.offset;
                    ra.sh = sl.h;
                    ra.spitch = sl.pitch;
                    ra.dst_off = dl.offset;
                    ra.dw = dl.w;
                    ra.dh = dl.h;
                    ra.dpitch = dl.pitch;
                    ra.qt = c->d_rs_qt + 3 * c->rs_q[l];
                    ra.yofs = c->d_rs_ofs + c->rs_y[l];
                    ra.ycoef = c->d_rs_coef + c->rs_y[l];
                    ra.frame0 = f0;
                    ra.n_frames = nf;
                    ra.quads = (dl.w + 3) / 4;
                    ra.inv_quads = 1.0f / (float)ra.quads;
                    ra.R = 8; // measured 4 / 8 / 16 / 32 rows per lane: 0.156 / 0.149 / 0.157 / 0.185 ms per 250 frames
                    ra.need_mask = c->rs_need[l];
                    launch_resize_col(ra, cs);
                }
                else // windows wider than 12 bytes (very large scale factors): generic per-pixel kernel
                    launch_resize(c->d_pyr, g, l, c->d_rs_ofs + c->rs_x[l], c->d_rs_coef + c->rs_x[l],
                                  c->d_rs_ofs + c->rs_y[l], c->d_rs_coef + c->rs_y[l], f0, nf, cs);
            }
        }
        {
            StageScope t(c, "fast", cs);
            launch_fast(c->d_pyr, g, c->d_cells, c->d_cell_cnt, c->d_cell_kp, c->p.ini_fast_thr, c->p.min_fast_thr, f0,
                        nf, cs);
        }
        const bool blur_needed = c->fused_levels < g.n_levels;
        const bool forked = c->fork_blur && !c->profiling && blur_needed;
        if(forked)
        {
            // quadtree (latency-bound, LDS-heavy) and blur (vector-ALU-bound, no LDS) only read the pyramid: side by side
            HIPCHK(c, hipEventRecord(c->ev_blur_fork[k], cs));
            HIPCHK(c, hipStreamWaitEvent(c->blur_stream[k], c->ev_blur_fork[k], 0));
            {
                StageScope t(c, "blur", c->blur_stream[k]);
                launch_blur(c->d_pyr, c->d_blur, g, c->d_blur_waves, c->blur_wpf, f0, nf, c->blur_stream[k]);
            }
            HIPCHK(c, hipEventRecord(c->ev_blur_join[k], c->blur_stream[k]));
        }
        {
            StageScope t(c, "quadtree", cs);
            launch_quadtree(g, c->quad, f0, nf, cs);
        }
        if(forked)
            HIPCHK(c, hipStreamWaitEvent(cs, c->ev_blur_join[k], 0));
        else if(blur_needed)
        {
            StageScope t(c, "blur", cs);
            launch_blur(c->d_pyr, c->d_blur, g, c->d_blur_waves, c->blur_wpf, f0, nf, cs);
        }
        } // in-tree detector
        if(cv_mode && c->fused_levels < g.n_levels)
        {
            StageScope t(c, "blur", cs);
            launch_blur(c->d_pyr, c->d_blur, g, c->d_blur_waves, c->blur_wpf, f0, nf, cs);
        }
        {
            StageScope t(c, "describe", cs);
            DescArgs a{};
            a.pyr = c->d_pyr;
            a.blur = c->d_blur;
            a.sel = c->quad.sel;
            a.sel_cnt = c->quad.sel_cnt;
            a.orient_w = c->d_orient_w;
            a.cand_cap = c->p.max_candidates;
            a.max_kp = c->p.max_keypoints;
            a.xy = c->d_xy + K * 2;
            a.desc = c->d_desc + K * 32;
            a.octave = c->d_octave + K;
            a.angle = c->d_angle + K;
            a.response = c->d_response + K;
            a.count = c->d_count + 1;
            a.flags = c->d_flags;
            if(cv_mode)
            {
                a.sel_resp = reinterpret_cast<const float*>(c->quad.best);
                a.cv_mode = 1;
            }
            if(c->mirror_results && n_frames == 1)
                a.h_mirror = c->d_h_out;
            launch_describe(g, a, f0, nf, cs);
        }
        if(n_chunks > 1)
        {
            HIPCHK(c, hipEventRecord(c->ev_join[k], cs));
            HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[k], 0));
        }
    }
    HIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

// set switching + carry of the previous batch's last frame; shared by the device and the host entry points
static int detect_prologue(mslam_hip_ctx* c)
{
    const size_t K = (size_t)c->p.max_keypoints;
    hipStream_t s = c->stream;
    if(!c->inplace_timing)
        c->timers_used = 0;
    // This batch goes into the other output set, so that a matcher still running on the previous batch (on
    // its own stream) is not disturbed.  The set we are about to fill was last read by the matcher of two
    // batches ago: wait for it.
    if(c->n_last > 0)
    {
        const int prev = c->cur, nxt = prev ^ 1;
        if(c->out[nxt].match_pending)
        {
            HIPCHK(c, hipStreamWaitEvent(s, c->out[nxt].ev_match, 0));
            c->out[nxt].match_pending = false;
        }
        // carry the last frame of the previous batch into slot 0 (predecessor of the new frame 0)
        const size_t last = (size_t)c->n_last;
        // (one small kernel: the count and exactly that many descriptors, instead of two copy launches of the capacity)
        hipLaunchKernelGGL(k_carry_prev, dim3(16), dim3(256), 0, s, reinterpret_cast<uint4*>(c->out[nxt].desc),
                           reinterpret_cast<const uint4*>(c->out[prev].desc + last * K * 32), c->out[nxt].count,
                           c->out[prev].count + last, (int)K);
        HIPCHK(c, hipGetLastError());
        c->have_prev = true;
        select_set(c, nxt);
    }
    return MSLAM_HIP_OK;
}

int mslam_hip_detect_batch_dev(mslam_hip_ctx* c, const uint8_t* d_bgr, int n_frames)
{
    ENTER(c);
    if(c->p.width == 0)
        return fail(c, MSLAM_HIP_E_INVALID, "detect_batch_dev: this context was created without a detector (0 x 0)");
    if(!d_bgr || n_frames < 1 || n_frames > c->p.max_batch)
        return fail(c, MSLAM_HIP_E_INVALID, "detect_batch_dev: n_frames outside [1, max_batch]");
    int rc = detect_prologue(c);
    if(rc)
        return rc;
    rc = enqueue_detect(c, d_bgr, n_frames);
    if(rc)
        return rc;
    HIPCHK(c, hipEventRecord(c->out[c->cur].ev_detect, c->stream));
    c->n_last = n_frames;
    ++c->detect_seq;
    return MSLAM_HIP_OK;
}

int mslam_hip_get_batch_view(mslam_hip_ctx* c, mslam_hip_batch_view* v)
{
    if(!c || !v)
        return MSLAM_HIP_E_INVALID;
    const size_t K = (size_t)c->p.max_keypoints;
    v->n_frames = c->n_last;
    v->capacity = c->p.max_keypoints;
    v->xy = c->d_xy + K * 2;
    v->desc = c->d_desc + K * 32;
    v->octave = c->d_octave + K;
    v->angle = c->d_angle + K;
    v->response = c->d_response + K;
    v->count = c->d_count + 1;
    v->match_from = c->d_mfrom;
    v->match_to = c->d_mto;
    v->match_count = c->d_mcount;
    return MSLAM_HIP_OK;
}

int mslam_hip_detect(mslam_hip_ctx* c, const uint8_t* bgr, int width, int height, int max_out, float* xy,
                     uint8_t* desc, int32_t* octave, float* angle, float* response, int* n_out)
{
    ENTER(c);
    if(n_out)
        *n_out = 0;
    if(!bgr || !n_out || max_out < 0 || (max_out > 0 && (!xy || !desc)))
        return fail(c, MSLAM_HIP_E_INVALID, "detect: null argument");
    if(c->p.width == 0)
        return fail(c, MSLAM_HIP_E_INVALID, "detect: this context was created without a detector (0 x 0)");
    if(width != c->p.width || height != c->p.height)
        return fail(c, MSLAM_HIP_E_INVALID, "detect: frame size differs from the context's");
    const size_t K = (size_t)c->p.max_keypoints;
    // Results come back through one pinned staging block: count, flags and all five arrays are queued as
    // asynchronous copies behind the kernels and ONE synchronisation covers them (a synchronous hipMemcpy per
    // array costs a round trip each), then the first n entries are handed to the caller's buffers.
    uint8_t* h = c->h_out;
    uint8_t* h_xy = h + 16;
    uint8_t* h_desc = h_xy + K * 8;
    uint8_t* h_oct = h_desc + K * 32;
    uint8_t* h_ang = h_oct + K * 4;
    uint8_t* h_resp = h_ang + K * 4;
    // (MSLAM_HIP_MIRROR_RESULTS=0: the packing kernel of round 3 instead of k_describe's own stores into the mapped block)
    const bool mirror_env = c->knob_mirror_results; // (read at context creation)
    struct MirrorScope
    {
        mslam_hip_ctx* c;
        MirrorScope(mslam_hip_ctx* cc, bool on) : c(cc) { c->mirror_results = on; }
        ~MirrorScope() { c->mirror_results = false; }
    } mirror_scope(c, mirror_env);
    auto enqueue_results = [&]() -> int {
        if(c->mirror_results)
            return MSLAM_HIP_OK;
        hipLaunchKernelGGL(k_pack_results, dim3(32), dim3(256), 0, c->stream, c->d_xy + K * 2, c->d_desc + K * 32, c->d_octave + K,
                           c->d_angle + K, c->d_response + K, c->d_count + 1, c->d_flags, c->d_h_out, (int)K);
        HIPCHK(c, hipGetLastError());
        return MSLAM_HIP_OK;
    };
    HIPCHK(c, hipStreamSynchronize(c->stream_m));
    // (the carry kernel of the prologue is queued first: it runs while the host stages the pageable frame for the copy)
    int rc = detect_prologue(c);
    if(rc)
        return rc;
    // The frame is copied by the CPU into page-locked, device-mapped memory and the gray kernel reads it from there over PCIe
    // (its deep-prefetch instance has every row of a block in flight at once: one round trip per block).  A copy from the
    // caller's pageable buffer blocks the host for the same CPU copy into the runtime's own staging area and only then starts
    // a DMA of ~20 us in front of the first kernel: detect 164 -> 148 us.  MSLAM_HIP_ZERO_COPY_FRAME=0: the copy.
    // The knob is read at context creation; frames above zero_copy_max_bytes (1.2 MB, MSLAM_HIP_ZERO_COPY_MAX_BYTES) take the
    // copy: at 1280x720 and 1920x1080 the kernel's PCIe reads lose to the DMA (numbers at the knob, create_impl).
    const bool zero_copy = c->knob_zero_copy && (size_t)width * height * 3 <= c->zero_copy_max_bytes;
    const uint8_t* frame_src = c->d_stage;
    if(zero_copy)
    {
        if(!c->h_stage)
        {
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), (size_t)width * height * 3, hipHostMallocMapped));
            HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_stage), c->h_stage, 0));
        }
        std::memcpy(c->h_stage, bgr, (size_t)width * height * 3);
        frame_src = c->d_h_stage;
    }
    else
        HIPCHK(c, hipMemcpyAsync(c->d_stage, bgr, (size_t)width * height * 3, hipMemcpyHostToDevice, c->stream));
    // The single-frame call is launch-bound (11 small kernels), so its fixed sequence — kernels + the result packing
    // kernel — is captured once per output set into a HIP graph and replayed.  With stage timing on, the plain path runs.
    if(!c->profiling && !c->inplace_timing && c->use_graph)
    {
        hipGraphExec_t& exec = c->detect_graph[c->cur];
        if(!exec)
        {
            hipGraph_t graph = nullptr;
            HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
            rc = enqueue_detect(c, frame_src, 1);
            if(!rc)
                rc = enqueue_results();
            const hipError_t e = hipStreamEndCapture(c->stream, &graph);
            // the captured graph is destroyed on every way out
            const hipError_t e_inst = (!rc && e == hipSuccess) ? hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) : hipSuccess;
            if(graph)
                (void)hipGraphDestroy(graph);
            if(e_inst != hipSuccess)
                exec = nullptr;
            if(rc)
                return rc;
            HIPCHK(c, e);
            HIPCHK(c, e_inst);
        }
        HIPCHK(c, hipGraphLaunch(exec, c->stream));
    }
    else
    {
        rc = enqueue_detect(c, frame_src, 1);
        if(rc)
            return rc;
        rc = enqueue_results();
        if(rc)
            return rc;
    }
    HIPCHK(c, hipEventRecord(c->out[c->cur].ev_detect, c->stream));
    c->n_last = 1;
    ++c->detect_seq;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t flags;
    int32_t n;
    std::memcpy(&n, h, 4);
    std::memcpy(&flags, h + 4, 4);
    if(flags)
        return check_flags(c); // reads the flags again, clears them and reports
    *n_out = n;
    if(n > max_out)
        return fail(c, MSLAM_HIP_E_CAPACITY, "detect: max_out smaller than the number of keypoints");
    if(n == 0)
        return MSLAM_HIP_OK;
    std::memcpy(xy, h_xy, (size_t)n * 8);
    std::memcpy(desc, h_desc, (size_t)n * 32);
    if(octave)
        std::memcpy(octave, h_oct, (size_t)n * 4);
    if(angle)
        std::memcpy(angle, h_ang, (size_t)n * 4);
    if(response)
        std::memcpy(response, h_resp, (size_t)n * 4);
    return MSLAM_HIP_OK;
}

// ---- matcher ----------------------------------------------------------------------------------------
static int upload_ratio_table(mslam_hip_ctx* c, double ratio)
{
    if(ratio == c->ratio_cached)
        return MSLAM_HIP_OK;
    // thr[d1] = number of integer d0 in [0,256] with (double)(float)d0 < ratio*(double)(float)d1
    // (orb_feature.cpp:101; DMatch::distance is a float holding the integer Hamming distance)
    int32_t thr[257];
    for(int d1 = 0; d1 <= 256; ++d1)
    {
        int n = 0;
        for(int d0 = 0; d0 <= 256; ++d0)
            if((double)(float)d0 < ratio * (double)(float)d1)
                ++n;
            else
                break;
        thr[d1] = n;
    }
    // a ratio test of an earlier batch may still be reading the table on the matcher stream
    HIPCHK(c, hipStreamSynchronize(c->stream_m));
    HIPCHK(c, hipMemcpyAsync(c->d_ratio_thr, thr, sizeof(thr), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // thr is a stack buffer
    c->ratio_cached = ratio;
    return MSLAM_HIP_OK;
}

int mslam_hip_match_batch_dev(mslam_hip_ctx* c, double ratio, int chain_previous)
{
    ENTER(c);
    if(c->n_last < 1)
        return fail(c, MSLAM_HIP_E_INVALID, "match_batch_dev: no detect batch to match");
    int rc = upload_ratio_table(c, ratio);
    if(rc)
        return rc;
    const size_t K = (size_t)c->p.max_keypoints;
    const int first = (chain_previous && c->have_prev) ? 0 : 1; // first frame that has a predecessor
    const int n_pairs = c->n_last - first;
    c->match_seq = c->detect_seq;
    // the matcher runs on its own stream behind the detect batch it reads, so the next detect batch can
    // start right away (with profiling on, everything stays on the context's stream)
    const bool own = c->overlap_match && !c->profiling;
    hipStream_t s = own ? c->stream_m : c->stream;
    if(own)
        HIPCHK(c, hipStreamWaitEvent(s, c->out[c->cur].ev_detect, 0));
    if(first == 1)
        HIPCHK(c, hipMemsetAsync(c->d_mcount, 0, 4, s));
    if(n_pairs > 0)
    {
        MatchArgs m{};
        // pair t: from = frame t (slot t+1), to = frame t-1 (slot t)
        m.from_desc = c->d_desc + (size_t)(first + 1) * K * 32;
        m.to_desc = c->d_desc + (size_t)first * K * 32;
        m.from_stride = m.to_stride = (long long)K * 32;
        m.from_cnt = c->d_count + first + 1;
        m.to_cnt = c->d_count + first;
        m.cap = c->p.max_keypoints;
        m.cap_from = c->p.max_keypoints;
        m.popcount_only = c->matcher_kind == MSLAM_HIP_MATCHER_POPCOUNT;
        m.idx0 = c->d_idx0 + (size_t)first * K;
        m.idx1 = c->d_idx1 + (size_t)first * K;
        m.dist0 = c->d_dist0 + (size_t)first * K;
        m.dist1 = c->d_dist1 + (size_t)first * K;
        {
            StageScope t(c, "match_knn2", s);
            c->last_match_kernel = launch_match_knn2(m, n_pairs, s);
        }
        RatioArgs r{};
        r.idx0 = m.idx0;
        r.dist0 = m.dist0;
        r.dist1 = m.dist1;
        r.from_cnt = m.from_cnt;
        r.to_cnt = m.to_cnt;
        r.cap = m.cap;
        r.thr = c->d_ratio_thr;
        r.from_idx = c->d_mfrom + (size_t)first * K;
        r.to_idx = c->d_mto + (size_t)first * K;
        r.n_out = c->d_mcount + first;
        {
            StageScope t(c, "ratio_compact", s);
            launch_ratio_compact(r, n_pairs, s);
        }
    }
    HIPCHK(c, hipGetLastError());
    if(own)
    {
        HIPCHK(c, hipEventRecord(c->out[c->cur].ev_match, s));
        c->out[c->cur].match_pending = true;
    }
    return MSLAM_HIP_OK;
}

constexpr int kHostMatchSlices = 8;
static int host_match_prepare(mslam_hip_ctx* c, const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to)
{
    // device descriptors: ONE buffer [train rows | query rows], so that one copy fills both
    if(n_from > c->hm_from_cap || n_to > c->hm_to_cap)
    {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int from_cap = std::max(n_from, std::max(c->hm_from_cap, 2048)), to_cap = std::max(n_to, std::max(c->hm_to_cap, 2048));
        if(c->d_hm_from)
            (void)hipFree(c->d_hm_from);
        if(c->d_hm_out)
            (void)hipFree(c->d_hm_out);
        if(c->h_hm)
            (void)hipHostFree(c->h_hm);
        c->d_hm_from = nullptr;
        c->d_hm_to = nullptr;
        c->d_hm_out = nullptr;
        c->h_hm = nullptr;
        c->d_h_hm = nullptr;
        c->hm_from_cap = c->hm_to_cap = 0;
        HIPCHK(c, dmalloc(c->d_hm_from, (size_t)(from_cap + to_cap) * 32 + 16)); // (+ 16: the captured form's counts ride behind the rows)
        HIPCHK(c, dmalloc(c->d_hm_out, (size_t)to_cap * 6 + 4));
        if(c->d_hm_partial)
            (void)hipFree(c->d_hm_partial);
        c->d_hm_partial = nullptr;
        HIPCHK(c, dmalloc(c->d_hm_partial, (size_t)kHostMatchSlices * 2 * to_cap));
        // page-locked, device-mapped: [descriptor staging (from | to) | from_idx | to_idx | n_out] — the caller's
        // (pageable) descriptors are copied here by the CPU and go up in ONE asynchronous copy (two blocking pageable
        // copies cost 25 us of the 84 us call); the ratio kernel writes its compacted pairs straight into the block
        const size_t bytes = (size_t)(from_cap + to_cap) * 32 + 16 + ((size_t)to_cap * 2 + 4) * 4;
        if(c->match_graph)
        {
            (void)hipGraphExecDestroy(c->match_graph); // captured on the buffers just freed
            c->match_graph = nullptr;
        }
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_hm), bytes, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_h_hm), c->h_hm, 0));
        c->hm_from_cap = from_cap;
        c->hm_to_cap = to_cap;
    }
    c->d_hm_to = c->d_hm_from + (size_t)n_from * 32; // packed right behind the train rows of THIS call
    if(n_from > 0)
        std::memcpy(c->h_hm, from_desc, (size_t)n_from * 32);
    std::memcpy(c->h_hm + (size_t)n_from * 32, to_desc, (size_t)n_to * 32);
    HIPCHK(c, hipMemcpyAsync(c->d_hm_from, c->h_hm, (size_t)(n_from + n_to) * 32, hipMemcpyHostToDevice, c->stream));
    return MSLAM_HIP_OK;
}

static void host_match_args(mslam_hip_ctx* c, int n_from, int n_to, MatchArgs& m)
{
    const size_t cap = (size_t)c->hm_to_cap;
    m = MatchArgs{};
    m.from_desc = c->d_hm_from;
    m.to_desc = c->d_hm_to;
    m.n_from_fixed = n_from;
    m.n_to_fixed = n_to;
    m.cap = n_to;
    m.idx0 = c->d_hm_out;
    m.idx1 = c->d_hm_out + cap;
    m.dist0 = c->d_hm_out + 2 * cap;
    m.dist1 = c->d_hm_out + 3 * cap;
    m.popcount_only = c->matcher_kind == MSLAM_HIP_MATCHER_POPCOUNT;
    // one pair: the call's latency is the kernel's latency, so the train tiles are scanned by several workgroups side by side
    // (about 8 tiles each) and merged (k_match.hip)
    m.partial = c->d_hm_partial;
    m.n_slices = std::max(1, std::min(kHostMatchSlices, (n_from + 255) / 256));
}

// mslam_hip_match as ONE graph launch (descriptor upload, matcher, merge, ratio test): the single synchronous call is bound
// by its launches, not by its kernels (three launches + a copy: ~17 us of host time for ~22 us of kernels).  What varies from
// call to call — the two row counts — rides behind the descriptors in the upload (the kernels take their counts from device
// words, as the batched path does), the launch shapes come from the staging capacities, the train rows sit at a fixed
// offset: the graph captured once serves every call until the capacities or the matcher kind change.
static int host_match_graph(mslam_hip_ctx* c, const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to, int* n_out,
                            int32_t* from_idx, int32_t* to_idx)
{
    const size_t fcap = (size_t)c->hm_from_cap, tcap = (size_t)c->hm_to_cap;
    uint8_t* h_cnt = c->h_hm + (fcap + tcap) * 32;
    if(n_from > 0)
        std::memcpy(c->h_hm, from_desc, (size_t)n_from * 32);
    if(n_to > 0)
        std::memcpy(c->h_hm + fcap * 32, to_desc, (size_t)n_to * 32);
    const int32_t counts[4] = {n_from, n_to, 0, 0};
    std::memcpy(h_cnt, counts, sizeof(counts));
    int32_t* res_dev = reinterpret_cast<int32_t*>(c->d_h_hm + (fcap + tcap) * 32 + 16);
    const int32_t* res = reinterpret_cast<const int32_t*>(c->h_hm + (fcap + tcap) * 32 + 16);
    if(!c->match_graph || c->match_graph_from_cap != c->hm_from_cap || c->match_graph_to_cap != c->hm_to_cap ||
       c->match_graph_kind != c->matcher_kind)
    {
        if(c->match_graph)
            (void)hipGraphExecDestroy(c->match_graph);
        c->match_graph = nullptr;
        hipGraph_t graph = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        hipError_t e = hipMemcpyAsync(c->d_hm_from, c->h_hm, (fcap + tcap) * 32 + 16, hipMemcpyHostToDevice, c->stream);
        const int32_t* d_cnt = reinterpret_cast<const int32_t*>(c->d_hm_from + (fcap + tcap) * 32);
        MatchArgs m{};
        m.from_desc = c->d_hm_from;
        m.to_desc = c->d_hm_from + fcap * 32;
        m.from_cnt = d_cnt;
        m.to_cnt = d_cnt + 1;
        m.cap = (int)tcap;
        m.cap_from = (int)fcap;
        m.idx0 = c->d_hm_out;
        m.idx1 = c->d_hm_out + tcap;
        m.dist0 = c->d_hm_out + 2 * tcap;
        m.dist1 = c->d_hm_out + 3 * tcap;
        m.popcount_only = c->matcher_kind == MSLAM_HIP_MATCHER_POPCOUNT;
        m.partial = c->d_hm_partial;
        m.n_slices = kHostMatchSlices;
        RatioArgs r{};
        r.idx0 = m.idx0;
        r.dist0 = m.dist0;
        r.dist1 = m.dist1;
        r.from_cnt = m.from_cnt;
        r.to_cnt = m.to_cnt;
        r.cap = (int)tcap;
        r.thr = c->d_ratio_thr;
        r.from_idx = res_dev;
        r.to_idx = res_dev + tcap;
        r.n_out = res_dev + 2 * tcap;
        if(launch_match_ratio_single(m, r, c->stream))
            c->last_match_kernel = 1;
        else
        {
            c->last_match_kernel = launch_match_knn2(m, 1, c->stream);
            launch_ratio_compact(r, 1, c->stream);
        }
        const hipError_t e_launch = hipGetLastError(); // a launch made inside the capture that was rejected
        const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
        // the captured graph is destroyed on EVERY way out of here, instantiated or not
        hipError_t e3 = (e == hipSuccess && e_launch == hipSuccess && e2 == hipSuccess)
                            ? hipGraphInstantiate(&c->match_graph, graph, nullptr, nullptr, 0)
                            : hipSuccess;
        if(graph)
            (void)hipGraphDestroy(graph);
        if(e3 != hipSuccess)
            c->match_graph = nullptr;
        HIPCHK(c, e);
        HIPCHK(c, e_launch);
        HIPCHK(c, e2);
        HIPCHK(c, e3);
        c->match_graph_from_cap = c->hm_from_cap;
        c->match_graph_to_cap = c->hm_to_cap;
        c->match_graph_kind = c->matcher_kind;
        c->match_graph_kernel = c->last_match_kernel;
    }
    c->last_match_kernel = c->match_graph_kernel;
    HIPCHK(c, hipGraphLaunch(c->match_graph, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int32_t n = res[2 * tcap];
    if(n < 0 || n > n_to || (size_t)n > tcap) // (a count from mapped memory sizes the copies below: never trust it blindly)
        return fail(c, MSLAM_HIP_E_RUNTIME, "match: the kernel reported an impossible match count");
    std::memcpy(from_idx, res, (size_t)n * 4);
    std::memcpy(to_idx, res + tcap, (size_t)n * 4);
    *n_out = n;
    return MSLAM_HIP_OK;
}

int mslam_hip_join_matcher(mslam_hip_ctx* c)
{
    ENTER(c);
    for(auto& o : c->out)
        if(o.match_pending)
            HIPCHK(c, hipStreamWaitEvent(c->stream, o.ev_match, 0)); // stays pending: detect_prologue still orders on it
    return MSLAM_HIP_OK;
}

int mslam_hip_set_matcher(mslam_hip_ctx* c, int kind)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(kind != MSLAM_HIP_MATCHER_AUTO && kind != MSLAM_HIP_MATCHER_POPCOUNT)
        return fail(c, MSLAM_HIP_E_INVALID, "set_matcher: unknown kind");
    c->matcher_kind = kind;
    return MSLAM_HIP_OK;
}

int mslam_hip_get_matcher(const mslam_hip_ctx* c) { return c ? c->matcher_kind : -1; }

int mslam_hip_set_cv_keypoint_order(mslam_hip_ctx* c, int order)
{
    ENTER(c);
    if(order != MSLAM_HIP_CV_ORDER_LIBSTDCXX && order != MSLAM_HIP_CV_ORDER_RASTER)
        return fail(c, MSLAM_HIP_E_INVALID, "set_cv_keypoint_order: unknown order");
    if(order == c->cv_order)
        return MSLAM_HIP_OK;
    // the captured single-frame sequences carry the order as a kernel argument: drop them, the next call captures again
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for(auto& e : c->detect_graph)
        if(e)
        {
            (void)hipGraphExecDestroy(e);
            e = nullptr;
        }
    c->cv_order = order;
    return MSLAM_HIP_OK;
}
int mslam_hip_last_match_kernel(const mslam_hip_ctx* c) { return c ? c->last_match_kernel : -1; }

int mslam_hip_match_knn2(mslam_hip_ctx* c, const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to,
                         int32_t* idx0, int32_t* idx1, int32_t* dist0, int32_t* dist1)
{
    ENTER(c);
    if(n_from < 0 || n_to < 0 || (n_from > 0 && !from_desc) || (n_to > 0 && (!to_desc || !idx0 || !idx1 || !dist0 || !dist1)))
        return fail(c, MSLAM_HIP_E_INVALID, "match_knn2: bad argument");
    if(n_from > 65535)
        return fail(c, MSLAM_HIP_E_INVALID, "match_knn2: more than 65535 train descriptors are not supported");
    if(n_to == 0)
        return MSLAM_HIP_OK;
    int rc = host_match_prepare(c, from_desc, n_from, to_desc, n_to);
    if(rc)
        return rc;
    MatchArgs m;
    host_match_args(c, n_from, n_to, m);
    c->last_match_kernel = launch_match_knn2(m, 1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(idx0, m.idx0, (size_t)n_to * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(idx1, m.idx1, (size_t)n_to * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dist0, m.dist0, (size_t)n_to * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(dist1, m.dist1, (size_t)n_to * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSLAM_HIP_OK;
}

int mslam_hip_match(mslam_hip_ctx* c, const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to,
                    double ratio, int32_t* from_idx, int32_t* to_idx, int* n_out)
{
    ENTER(c);
    if(n_out)
        *n_out = 0;
    if(!n_out || n_from < 0 || n_to < 0 || (n_from > 0 && !from_desc) || (n_to > 0 && (!to_desc || !from_idx || !to_idx)))
        return fail(c, MSLAM_HIP_E_INVALID, "match: bad argument");
    if(n_from > 65535)
        return fail(c, MSLAM_HIP_E_INVALID, "match: more than 65535 train descriptors are not supported");
    if(n_to == 0 || n_from < 2)
        return MSLAM_HIP_OK; // reference: UB for n_from < 2 (orb_feature.cpp:101); defined here as no matches
    int rc = upload_ratio_table(c, ratio);
    if(rc)
        return rc;
    // the captured form: when the staging capacities are at most twice what this call needs (the upload always moves whole
    // capacities) and the matrix-core kernel's train range holds them
    const bool graph_env = c->knob_match_graph; // (read at context creation)
    if(graph_env && c->use_graph && !c->profiling && !c->inplace_timing && c->hm_from_cap >= n_from && c->hm_to_cap >= n_to &&
       c->hm_from_cap + c->hm_to_cap <= 2 * (n_from + n_to) + 2048 && c->hm_from_cap <= 32736)
        return host_match_graph(c, from_desc, n_from, to_desc, n_to, n_out, from_idx, to_idx);
    rc = host_match_prepare(c, from_desc, n_from, to_desc, n_to);
    if(rc)
        return rc;
    if(graph_env && c->use_graph && !c->profiling && !c->inplace_timing && c->hm_from_cap <= 32736 &&
       c->hm_from_cap + c->hm_to_cap <= 2 * (n_from + n_to) + 2048)
    {
        // (first call / the buffers have just grown: the plain path's copy is queued — let it drain, then take the captured form)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return host_match_graph(c, from_desc, n_from, to_desc, n_to, n_out, from_idx, to_idx);
    }
    MatchArgs m;
    host_match_args(c, n_from, n_to, m);
    c->last_match_kernel = launch_match_knn2(m, 1, c->stream);
    const size_t cap = (size_t)c->hm_to_cap;
    RatioArgs r{};
    r.idx0 = m.idx0;
    r.dist0 = m.dist0;
    r.dist1 = m.dist1;
    r.n_from_fixed = n_from;
    r.n_to_fixed = n_to;
    r.cap = n_to;
    r.thr = c->d_ratio_thr;
    // the compacted pairs and their count land in the mapped host block (they are small: 8 bytes per match)
    int32_t* res_dev = reinterpret_cast<int32_t*>(c->d_h_hm + (size_t)(c->hm_from_cap + c->hm_to_cap) * 32 + 16);
    const int32_t* res = reinterpret_cast<const int32_t*>(c->h_hm + (size_t)(c->hm_from_cap + c->hm_to_cap) * 32 + 16);
    r.from_idx = res_dev;
    r.to_idx = res_dev + cap;
    r.n_out = res_dev + 2 * cap;
    launch_ratio_compact(r, 1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int32_t n = res[2 * cap];
    std::memcpy(from_idx, res, (size_t)n * 4);
    std::memcpy(to_idx, res + cap, (size_t)n * 4);
    *n_out = n;
    return MSLAM_HIP_OK;
}

// ---- debug / test access ----------------------------------------------------------------------------
int mslam_hip_level_geometry(mslam_hip_ctx* c, int* widths, int* heights, float* scales)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    for(int l = 0; l < c->geom.n_levels; ++l)
    {
        if(widths)
            widths[l] = c->geom.lv[l].w;
        if(heights)
            heights[l] = c->geom.lv[l].h;
        if(scales)
            scales[l] = c->geom.lv[l].scale;
    }
    return MSLAM_HIP_OK;
}

int mslam_hip_debug_read(mslam_hip_ctx* c, int what, int frame, int level, void* dst, size_t dst_bytes,
                         size_t* n_items)
{
    ENTER(c);
    if(c->p.width == 0 || frame < 0 || frame >= c->p.max_batch || level < 0 || level >= c->geom.n_levels || !dst || !n_items)
        return fail(c, MSLAM_HIP_E_INVALID, "debug_read: bad argument");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const LevelGeom& lv = c->geom.lv[level];
    if(what == MSLAM_HIP_DBG_PYRAMID || what == MSLAM_HIP_DBG_BLURRED)
    {
        const size_t need = (size_t)lv.w * lv.h;
        *n_items = need;
        if(dst_bytes < need)
            return fail(c, MSLAM_HIP_E_CAPACITY, "debug_read: buffer too small");
        const uint8_t* src = (what == MSLAM_HIP_DBG_PYRAMID ? c->d_pyr : c->d_blur) + (size_t)frame * c->geom.slab + lv.offset;
        if(what == MSLAM_HIP_DBG_BLURRED && c->geom.blur_tiled)
        {
            // the blurred slab is stored in tiles: copy the plane's tile rows and put the rows back together here
            std::vector<uint8_t> tmp((size_t)lv.pitch * ((lv.h + kTileH - 1) & ~(kTileH - 1)));
            HIPCHK(c, hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost));
            uint8_t* out = static_cast<uint8_t*>(dst);
            for(int y = 0; y < lv.h; ++y)
                for(int x = 0; x < lv.w; ++x)
                    out[(size_t)y * lv.w + x] = tmp[tiled_off((unsigned)lv.pitch, x, y)];
            return MSLAM_HIP_OK;
        }
        HIPCHK(c, hipMemcpy2D(dst, lv.w, src, lv.pitch, lv.w, lv.h, hipMemcpyDeviceToHost));
        return MSLAM_HIP_OK;
    }
    if(what == MSLAM_HIP_DBG_CANDIDATES || what == MSLAM_HIP_DBG_SELECTED)
    {
        const size_t slot = (size_t)frame * c->geom.n_levels + level;
        const bool cand = what == MSLAM_HIP_DBG_CANDIDATES;
        uint32_t n = 0;
        HIPCHK(c, hipMemcpy(&n, (cand ? c->quad.cand_cnt : c->quad.sel_cnt) + slot, 4, hipMemcpyDeviceToHost));
        n = std::min<uint32_t>(n, (uint32_t)c->p.max_candidates); // an overflowing level keeps counting (flagged separately)
        *n_items = n;
        if(dst_bytes < (size_t)n * 12)
            return fail(c, MSLAM_HIP_E_CAPACITY, "debug_read: buffer too small");
        std::vector<uint32_t> tmp(n);
        if(n)
            HIPCHK(c, hipMemcpy(tmp.data(), (cand ? c->quad.cand : c->quad.sel) + slot * (size_t)c->p.max_candidates,
                                (size_t)n * 4, hipMemcpyDeviceToHost));
        float* out = static_cast<float*>(dst);
        for(uint32_t i = 0; i < n; ++i)
        {
            out[3 * i] = (float)kp_x(tmp[i]);
            out[3 * i + 1] = (float)kp_y(tmp[i]);
            out[3 * i + 2] = (float)kp_score(tmp[i]);
        }
        if(c->p.detector == MSLAM_HIP_DETECTOR_CV_ORB && !cand)
        {
            // cv::ORB mode: absolute level coordinates (the list is kept relative to (19, 19) for k_describe) and the
            // Harris response
            std::vector<float> resp(n);
            if(n)
                HIPCHK(c, hipMemcpy(resp.data(), reinterpret_cast<const float*>(c->quad.best) + slot * (size_t)c->p.max_candidates,
                                    (size_t)n * 4, hipMemcpyDeviceToHost));
            for(uint32_t i = 0; i < n; ++i)
            {
                out[3 * i] += (float)kBorder;
                out[3 * i + 1] += (float)kBorder;
                out[3 * i + 2] = resp[i];
            }
        }
        return MSLAM_HIP_OK;
    }
    return fail(c, MSLAM_HIP_E_INVALID, "debug_read: unknown item");
}

int mslam_hip_debug_counts(mslam_hip_ctx* c, int what, int32_t* out, int n_frames)
{
    ENTER(c);
    if(c->p.width == 0 || !out || n_frames < 1 || (what != MSLAM_HIP_DBG_CANDIDATES && what != MSLAM_HIP_DBG_SELECTED))
        return fail(c, MSLAM_HIP_E_INVALID, "debug_counts: bad argument");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the scratch arrays are indexed by the frame's position in the batch: [frame][level]; never more rows than the
    // caller has room for
    const int rows = std::min(n_frames, std::max(c->n_last, 1));
    HIPCHK(c, hipMemcpy(out, what == MSLAM_HIP_DBG_CANDIDATES ? c->quad.cand_cnt : c->quad.sel_cnt,
                        (size_t)rows * c->geom.n_levels * 4, hipMemcpyDeviceToHost));
    return MSLAM_HIP_OK;
}

int mslam_hip_copy_to_host(mslam_hip_ctx* c, void* dst_host, const void* src_dev, size_t bytes)
{
    ENTER(c);
    if(!dst_host || !src_dev)
        return fail(c, MSLAM_HIP_E_INVALID, "copy_to_host: null pointer");
    HIPCHK(c, hipStreamSynchronize(c->stream_m));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return MSLAM_HIP_OK;
}

int mslam_hip_set_profiling(mslam_hip_ctx* c, int enable)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    c->profiling = enable == 1;
    c->inplace_timing = enable == 2;
    c->timers_used = 0;
    return MSLAM_HIP_OK;
}

int mslam_hip_get_stage_times(mslam_hip_ctx* c, const char** names, float* ms, int cap, int* n)
{
    ENTER(c);
    if(!n)
        return fail(c, MSLAM_HIP_E_INVALID, "get_stage_times: null output");
    HIPCHK(c, hipStreamSynchronize(c->stream_m));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int k = 0;
    for(size_t i = 0; i < c->timers_used && k < cap; ++i, ++k)
    {
        float t = 0;
        HIPCHK(c, hipEventElapsedTime(&t, c->timers[i].start, c->timers[i].stop));
        if(names)
            names[k] = c->timers[i].name;
        if(ms)
            ms[k] = t;
    }
    *n = k;
    if(c->inplace_timing)
        c->timers_used = 0; // entries accumulate across calls in this mode; reading them starts a new window
    return MSLAM_HIP_OK;
}

} // extern "C"
