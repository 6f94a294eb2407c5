// context.hpp — the object behind mslam_hip_ctx: host-built tables, device buffers, stream.
#pragma once
#include "common.hpp"
#include "../../include/mslam_hip.h"
#include <array>
#include <string>
#include <vector>

namespace mslam
{
struct BowState; // k_bow.hip
void bow_destroy(BowState*);
void set_blur_taps(const int* taps);
void build_blur_waves(const Geometry& g, int first_level, std::vector<BlurWave>& out);

struct StageTimer
{
    const char* name;
    hipEvent_t start, stop;
};
} // namespace mslam

// One set of per-batch output buffers.  Two sets alternate so that the matcher of batch i (on its own
// stream) can run while the detector of batch i+1 fills the other set.
struct mslam_out_set
{
    float* xy = nullptr;
    uint8_t* desc = nullptr;
    int32_t* octave = nullptr;
    float* angle = nullptr;
    float* response = nullptr;
    int32_t* count = nullptr;
    int32_t *idx0 = nullptr, *idx1 = nullptr, *dist0 = nullptr, *dist1 = nullptr;
    int32_t *mfrom = nullptr, *mto = nullptr, *mcount = nullptr;
    hipEvent_t ev_detect = nullptr, ev_match = nullptr;
    bool match_pending = false;
};

struct mslam_hip_ctx
{
    mslam_hip_params p{};
    mslam::Geometry geom{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side streams for chunked batches (see mslam_hip_detect_batch_dev)
    int n_side = 2;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // the blur of a chunk runs on a stream of its own beside the chunk's quadtree (both only read the pyramid)
    hipStream_t blur_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_blur_fork[4] = {nullptr, nullptr, nullptr, nullptr}, ev_blur_join[4] = {nullptr, nullptr, nullptr, nullptr};
    bool fork_blur = false;
    std::string err;

    // host copies of the tables
    std::vector<mslam::CellDesc> cells;

    // device tables
    mslam::CellDesc* d_cells = nullptr;
    int32_t* d_rs_ofs = nullptr;   // resize offsets, all levels
    uint32_t* d_rs_coef = nullptr; // resize coefficients, all levels
    std::vector<size_t> rs_x, rs_y; // per-level start index into d_rs_*
    uint4* d_rs_qt = nullptr;       // quad tables (k_resize_col), all levels, 3 x uint4 per quad
    std::vector<size_t> rs_q;       // per-level start (in quads) into the quad tables; SIZE_MAX = use the generic kernel
    std::vector<int> rs_need;       // per level: which pixel positions of a quad ever use the upper dword pair
    // cv::ORB mode: INTER_LINEAR_EXACT tables (all levels) and per-level quota
    int32_t* d_cv_ofs = nullptr;
    uint32_t* d_cv_coef = nullptr;
    std::vector<size_t> cv_x, cv_y;           // per-level start index into d_cv_*
    std::vector<std::array<int, 4>> cv_range; // per level: xmin, xmax, ymin, ymax
    std::vector<int> cv_window12;             // per level: k_resize_exact may use its 12-byte-window form
    int cv_quota[mslam::kMaxLevels] = {0};
    int32_t* d_ratio_thr = nullptr; // [257]
    uint32_t* d_orient_w = nullptr; // [2][256] intensity-centroid disc weights
    hipGraphExec_t detect_graph[2] = {nullptr, nullptr}; // mslam_hip_detect's kernel + copy sequence, per output set
    bool use_graph = true;
    int cv_order = 0; // MSLAM_HIP_CV_ORDER_*: the cv::ORB mode's keypoint order inside a level (mslam_hip_set_cv_keypoint_order)
    bool mirror_results = false; // set by mslam_hip_detect around its enqueue: k_describe writes the results into h_out as well (no packing kernel)
    // mslam_hip_match's sequence (descriptor upload + matcher + merge + ratio test) as a graph: sizes come from a mapped word pair,
    // the launch shapes from the staging capacities, so one graph serves every call until the capacities or the matcher change
    hipGraphExec_t match_graph = nullptr;
    int match_graph_from_cap = 0, match_graph_to_cap = 0, match_graph_kind = -1, match_graph_kernel = 0;
    uint8_t* d_h_out = nullptr;     // the device address of h_out (page-locked, mapped)
    uint8_t* h_out = nullptr;       // pinned staging of mslam_hip_detect's results: [count, flags | xy | desc | octave | angle | response] for K keypoints
    double ratio_cached = -1.0;

    // device working set (sized for max_batch frames)
    uint8_t* d_stage = nullptr; // one frame of BGR for the host-pointer entry point
    uint8_t* h_stage = nullptr; // the same as page-locked, device-mapped host memory (zero-copy upload, MSLAM_HIP_ZERO_COPY_FRAME)
    uint8_t* d_h_stage = nullptr;
    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;
    mslam::BlurWave* d_blur_waves = nullptr; // k_blur2 wave descriptors of one frame
    int blur_wpf = 0;
    int fused_levels = 0;  // levels 0 .. fused_levels-1 are produced and blurred by k_level.hip; k_blur2 takes the rest
    int level_k6 = 9;      // k_level.hip: rows per block = 6 k6 + 2
    bool knob_mirror_results = true, knob_zero_copy = true, knob_match_graph = true; // MSLAM_HIP_MIRROR_RESULTS / _ZERO_COPY_FRAME / _MATCH_GRAPH at creation
    size_t zero_copy_max_bytes = 1200000; // frames above this size are copied by DMA instead of read over PCIe by the gray kernel
    int level_chain = 0, level_chain_frames = 2, level_chain_waves = 8, level_chain_k6 = 9; // k_level_chain (k_level.hip)
    int level_k6_small = 1; // the same for batches of fewer than 8 frames (latency: one wave's walk is the launch's duration)
    uint32_t* d_cell_cnt = nullptr;
    uint32_t* d_cell_kp = nullptr;
    mslam::QuadArgs quad{};
    uint32_t* d_flags = nullptr;

    // two alternating output sets; the d_* members below alias the set of the last detect batch
    mslam_out_set out[2];
    int cur = 0;
    hipStream_t stream_m = nullptr; // matcher stream
    bool overlap_match = true;
    int matcher_kind = 0; // MSLAM_HIP_MATCHER_*
    int last_match_kernel = 0; // kernel of the last matcher launch: 0 none yet, 1 matrix cores, 2 xor/popcount
    // outputs: slot 0 = last frame of the previous batch, slots 1..max_batch = current batch
    float* d_xy = nullptr;
    uint8_t* d_desc = nullptr;
    int32_t* d_octave = nullptr;
    float* d_angle = nullptr;
    float* d_response = nullptr;
    int32_t* d_count = nullptr;
    int n_last = 0;         // frames in the last detect batch
    unsigned long long detect_seq = 0; // counts detect batches; points_seq = the batch the back-projected points belong to
    unsigned long long points_seq = ~0ull;
    unsigned long long match_seq = ~0ull; // the detect batch mslam_hip_match_batch_dev last matched (mslam_hip_pack_batch_dev packs no stale pairs)
    bool have_prev = false; // slot 0 holds a real predecessor of the current batch

    // matcher
    int32_t *d_idx0 = nullptr, *d_idx1 = nullptr, *d_dist0 = nullptr, *d_dist1 = nullptr;
    int32_t *d_mfrom = nullptr, *d_mto = nullptr, *d_mcount = nullptr;
    // host-pointer matcher scratch (grown on demand)
    uint8_t *d_hm_from = nullptr, *d_hm_to = nullptr;
    int32_t* d_hm_out = nullptr; // 6 arrays x cap + 1
    int hm_from_cap = 0, hm_to_cap = 0;
    uint32_t* d_hm_partial = nullptr; // per-slice top-2 keys of the sliced single-pair matcher
    uint8_t *h_hm = nullptr, *d_h_hm = nullptr; // page-locked, mapped staging of the host-pointer matcher (host / device address)

    // RGB-D back-projection outputs (allocated on first use)
    double* d_xyz = nullptr;
    uint8_t* d_valid = nullptr;
    // batched PnP (allocated on first use)
    float *d_pnp_obj = nullptr, *d_pnp_img = nullptr;
    int32_t *d_pnp_n = nullptr, *d_pnp_counts = nullptr;
    double *d_pnp_hyp = nullptr, *d_pnp_out = nullptr;
    uint8_t* d_pnp_mask = nullptr;
    int pnp_iterations = 0;
    double pnp_confidence = 0.99; // cv_ransac_pnp.cpp:57 (mslam_hip_pnp_set_confidence)
    // single-problem PnP scratch (mslam_hip_pnp_ransac), grown on demand
    float *d_pnp1_obj = nullptr, *d_pnp1_img = nullptr;
    double *d_pnp1_hyp = nullptr, *d_pnp1_out = nullptr;
    int32_t* d_pnp1_counts = nullptr;
    uint8_t* d_pnp1_mask = nullptr;
    int pnp1_n_cap = 0, pnp1_it_cap = 0;
    bool pnp_attr_set = false; // the > 64 KB dynamic-LDS attribute of the PnP kernels, per context (= per device)

    mslam::BowState* bow = nullptr;

    bool profiling = false;      // mode 1: every stage timed, everything serialised on the context's stream
    bool inplace_timing = false; // mode 2: every stage launch is timed in place on the stream it runs on
    std::vector<mslam::StageTimer> timers;
    size_t timers_used = 0;
};

namespace mslam
{
// Records a [start, stop] HIP-event pair around a stage.  Mode 1 (profiling): everything runs on the context's
// stream, so the events go there and the stages do not overlap.  Mode 2 (inplace_timing): the events are
// recorded on the stream the stage is launched on, which does not change the schedule; entries accumulate until
// mslam_hip_get_stage_times reads them.
struct StageScope
{
    mslam_hip_ctx* c;
    mslam::StageTimer* t = nullptr;
    hipStream_t st;
    StageScope(mslam_hip_ctx* ctx, const char* name, hipStream_t launch_stream = nullptr) : c(ctx)
    {
        if(!c->profiling && !c->inplace_timing)
            return;
        if(c->timers_used >= 8192)
            return; // mode 2 accumulates until read: bound the number of live events
        st = (c->profiling || !launch_stream) ? c->stream : launch_stream;
        if(c->timers_used == c->timers.size())
        {
            mslam::StageTimer nt{name, nullptr, nullptr};
            if(hipEventCreate(&nt.start) != hipSuccess || hipEventCreate(&nt.stop) != hipSuccess)
                return;
            c->timers.push_back(nt);
        }
        t = &c->timers[c->timers_used++];
        t->name = name;
        (void)hipEventRecord(t->start, st);
    }
    ~StageScope()
    {
        if(t)
            (void)hipEventRecord(t->stop, st);
    }
};

// bow entry points used by api.hip
int bow_batch(mslam_hip_ctx* c, int add_to_db);
} // namespace mslam
