/* mslam_hip.h — C ABI of the MI355X-native ORB / Hamming-match / BoW front end.
 *
 * This is the drop-in boundary for modular-slam's feature hot path.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repo,
 * src/lib/modular_slam/...).  Plain C types only; no exceptions cross this boundary; every function
 * returns an int status (0 = MSLAM_HIP_OK) and mslam_hip_last_error() gives the text.
 *
 * Threading: a context is NOT thread-safe (the reference detector is not re-entrant either — it owns
 * its pyramid scratch, distributed_cv_feature.cpp:651).  Use one context per GPU / per caller thread.
 * Host-pointer entry points are synchronous (they return after the D2H copy); *_dev entry points
 * enqueue on the context's HIP stream and return without synchronising.
 */
#ifndef MSLAM_HIP_H_
#define MSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSLAM_HIP_ABI_VERSION 5

enum
{
    MSLAM_HIP_OK = 0,
    MSLAM_HIP_E_INVALID = 1,  /* bad argument / unsupported size */
    MSLAM_HIP_E_RUNTIME = 2,  /* HIP runtime failure (no device, launch failure, OOM...) */
    MSLAM_HIP_E_CAPACITY = 3, /* an output or scratch capacity was exceeded; no partial result is valid */
    MSLAM_HIP_E_NO_VOCABULARY = 4,
    MSLAM_HIP_E_FORMAT = 5,   /* vocabulary stream not understood */
    MSLAM_HIP_E_NO_MODEL = 6  /* RANSAC found no model (cv::solvePnPRansac returning false) */
};

typedef struct mslam_hip_ctx mslam_hip_ctx;

/* Detector parameters.  Defaults are the reference's hard-coded operating point
 * (distributed_cv_feature.cpp:1184-1186): orb_params("orb", 1.2f, 8, 20, 7), min node area 1000. */
typedef struct
{
    int32_t width, height;  /* frame size every call on this context uses (reference: 640x480); 0 x 0 = a
                             * context without detector (matcher / BoW only: no pyramid buffers are allocated)    */
    int32_t max_batch;      /* frames per batched device launch, >= 1                                      */
    int32_t n_levels;       /* pyramid levels (8)                                                          */
    float scale_factor;     /* pyramid scale (1.2f); level scales are the float32 chain of :411-420        */
    int32_t ini_fast_thr;   /* first FAST threshold (20)                                                   */
    int32_t min_fast_thr;   /* fallback threshold for cells that found nothing (7), :922-926               */
    uint32_t min_node_area; /* quadtree stop area in level-0 px^2 (1000), :1002,:1186                      */
    int32_t max_keypoints;  /* per-frame keypoint capacity of the output buffers                           */
    int32_t max_candidates; /* per-(frame,level) FAST candidate capacity feeding the quadtree              */
    int32_t device;         /* HIP device ordinal                                                          */
    void* stream;           /* hipStream_t to enqueue on; NULL = the context creates its own               */
    /* which of the reference's two IFeatureDetector implementations this context is a drop-in for: */
    int32_t detector;       /* MSLAM_HIP_DETECTOR_DISTRIBUTED (default) or MSLAM_HIP_DETECTOR_CV_ORB               */
    int32_t n_features;     /* CV_ORB: cv::ORB::create(nfeatures) (1000, orb_feature.cpp:25)                       */
    int32_t edge_threshold; /* CV_ORB: edgeThreshold (31, cv::ORB default); FAST threshold = ini_fast_thr (20)      */
} mslam_hip_params;

enum
{
    /* DistributedOrbOpenCvDetector (distributed_cv_feature.cpp:1181-1222): 64-px FAST cells + quadtree, util::cos/sin */
    MSLAM_HIP_DETECTOR_DISTRIBUTED = 0,
    /* OrbOpenCvDetector (orb_feature.cpp:25,33-65): toGrayScale + cv::ORB::create(n)->detectAndCompute — INTER_LINEAR_EXACT
     * pyramid, whole-level FAST, retainBest(2n) / Harris response / retainBest(n) per level, libm-style cos/sin.
     * Keypoint ORDER inside a level: what KeyPointsFilter::retainBest's std::nth_element + std::partition leave behind —
     * defined by the C++ library, not by the standard.  The reference is a GCC build, so the default reproduces libstdc++'s
     * introselect / partition step by step (mslam_hip_set_cv_keypoint_order); the raster order of FAST is the alternative.
     * cos / sin of the keypoint angle come from include/mslam_sincos.h, which equals the host library's
     * (float)cos((double)angle) / (float)sin((double)angle) — the expression orb.cpp evaluates in a GCC build — for every
     * float in [0, 6.5] (checked exhaustively against glibc 2.35).  n_levels, scale_factor, ini_fast_thr keep their meaning;
     * min_fast_thr / min_node_area are unused. */
    MSLAM_HIP_DETECTOR_CV_ORB = 1
};
/* CV_ORB detector: where the two retainBest calls of a level leave their survivors (the kept SET is the same either way).
 * LIBSTDCXX (default) = the order of a GCC build of the reference, hence its keypoint ids and DescriptorMatch indices;
 * RASTER = FAST's (y, x) order — a little faster (the selection is then a threshold and a compaction). */
enum
{
    MSLAM_HIP_CV_ORDER_LIBSTDCXX = 0,
    MSLAM_HIP_CV_ORDER_RASTER = 1
};
int mslam_hip_set_cv_keypoint_order(mslam_hip_ctx* ctx, int order);

void mslam_hip_default_params(mslam_hip_params* p);
int mslam_hip_abi_version(void);

int mslam_hip_create(const mslam_hip_params* p, mslam_hip_ctx** out);
void mslam_hip_destroy(mslam_hip_ctx* ctx);
/* ctx may be NULL: returns the message of the last failed mslam_hip_create on this thread. */
const char* mslam_hip_last_error(const mslam_hip_ctx* ctx);
/* Block until everything enqueued on the context's stream has finished; also surfaces capacity
 * overflows recorded by *_dev launches (returns MSLAM_HIP_E_CAPACITY once, then clears). */
int mslam_hip_sync(mslam_hip_ctx* ctx);

/* ---- IFeatureDetector<RgbFrame,uint8_t,32>::detect ------------------------------------------------
 * Replaces DistributedOrbOpenCvDetector::detect (distributed_cv_feature.cpp:1190-1222; interface
 * frontend/feature/feature_interface.hpp:50-56).  `bgr` is the RgbFrame::data buffer: interleaved
 * 3-channel u8, row-major, no padding (types/rgb_frame.hpp:12-16).  Outputs are SoA, caller
 * allocated, `max_out` entries each: xy = scale-corrected float coordinates (x0,y0,x1,y1,...) in
 * the reference's order (level 0..L-1, quadtree node-list order inside a level); desc = 32 bytes per
 * keypoint; octave/angle/response may be NULL.  The reference's Keypoint::id is the output index. */
int mslam_hip_detect(mslam_hip_ctx* ctx, const uint8_t* bgr, int width, int height, int max_out, float* xy,
                     uint8_t* desc, int32_t* octave, float* angle, float* response, int* n_out);

/* Device-resident batched form of the same operator: `d_bgr` holds n_frames (<= max_batch)
 * back-to-back frames in HBM.  Results stay in context-owned device buffers (mslam_hip_batch_view). */
int mslam_hip_detect_batch_dev(mslam_hip_ctx* ctx, const uint8_t* d_bgr, int n_frames);

typedef struct
{
    int32_t n_frames;     /* frames in the last detect batch                                     */
    int32_t capacity;     /* max_keypoints: per-frame stride (in keypoints) of the arrays below  */
    const float* xy;      /* [max_batch][capacity][2]                                            */
    const uint8_t* desc;  /* [max_batch][capacity][32]                                           */
    const int32_t* octave;
    const float* angle;
    const float* response;
    const int32_t* count; /* [max_batch] keypoints per frame                                     */
    /* results of mslam_hip_match_batch_dev, frame t matched against frame t-1: */
    const int32_t* match_from; /* [max_batch][capacity] index into frame t   (fromIndex)         */
    const int32_t* match_to;   /* [max_batch][capacity] index into frame t-1 (toIndex)           */
    const int32_t* match_count; /* [max_batch]; 0 for a frame without predecessor                */
} mslam_hip_batch_view;
int mslam_hip_get_batch_view(mslam_hip_ctx* ctx, mslam_hip_batch_view* view);

/* ---- IFeatureMatcher<uint8_t,32>::match ------------------------------------------------------------
 * Replaces OrbOpenCvMatcher::match(from, to) (orb_feature.cpp:84-117; interface
 * feature_interface.hpp:62-70): BFMatcher(HAMMING).knnMatch(query = to, train = from, k = 2) and the
 * ratio test (double)d0 < ratio*(double)d1 (:96-105, reference ratio 0.7).  Descriptors are packed
 * 32-byte rows.  Outputs (capacity n_to each) are ordered by query (= to) index (:110-114).
 * n_from < 2 is undefined behaviour in the reference (:101); here it yields zero matches. */
int mslam_hip_match(mslam_hip_ctx* ctx, const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to,
                    double ratio, int32_t* from_idx, int32_t* to_idx, int* n_out);
/* The knnMatch(k=2) result itself, per query row of `to`: train indices (-1 if absent) and integer
 * Hamming distances (INT32_MAX if absent), best first; ties rank the lower train index first. */
int mslam_hip_match_knn2(mslam_hip_ctx* ctx, const uint8_t* from_desc, int n_from, const uint8_t* to_desc,
                         int n_to, int32_t* idx0, int32_t* idx1, int32_t* dist0, int32_t* dist1);
/* Batched device form: for every frame t of the last detect batch, match(from = frame t,
 * to = frame t-1); frame 0 is matched against the last frame of the previous batch when
 * `chain_previous` is non-zero and one exists.  Results: mslam_hip_batch_view.match_*. */
int mslam_hip_match_batch_dev(mslam_hip_ctx* ctx, double ratio, int chain_previous);
/* The batched matcher runs on a stream of its own behind the detect batch it reads.  This makes the context's
 * stream wait (on the device, no host synchronisation) for every matcher launch enqueued so far, so that work the
 * caller enqueues on the context's stream afterwards — e.g. an asynchronous copy of the match results — sees them. */
int mslam_hip_join_matcher(mslam_hip_ctx* ctx);
/* Which kernel computes the 256-bit Hamming distances (results are identical, tests run both):
 * AUTO = matrix cores (bits as FP4 +-1, exact) up to 32736 train rows and xor/popcount beyond; POPCOUNT = the
 * xor/__popc form BASELINE.json's north_star names, always.  The initial value comes from the environment
 * variable MSLAM_HIP_MATCHER ("popcount") read at mslam_hip_create. */
enum
{
    MSLAM_HIP_MATCHER_AUTO = 0,
    MSLAM_HIP_MATCHER_POPCOUNT = 1
};
int mslam_hip_set_matcher(mslam_hip_ctx* ctx, int kind);
int mslam_hip_get_matcher(const mslam_hip_ctx* ctx);
/* which kernel the last matcher launch of this context took: 0 = none yet, 1 = matrix cores, 2 = xor/popcount (AUTO decides on
 * the CAPACITY of the train side: max_keypoints on the batched path, n_from on the host-pointer calls) */
int mslam_hip_last_match_kernel(const mslam_hip_ctx* ctx);

/* ---- IRelocalizer / ILoopDetector: DBoW3 bag of words ---------------------------------------------
 * Replaces what OrbRelocalizer is wired for (orb_relocalizer.cpp:26-50, relocalizer.hpp:11-20,
 * loop_detection.hpp:10-15): DBoW3::Vocabulary::transform and Database add/query with L1 scoring.
 * `blob` is a DBoW3 binary vocabulary stream (Vocabulary::toStream/fromStream,
 * conan_recipes/dbow3/dbow3.patch:2252-2355,2544-2651), plain or saved with compressed = true (QuickLZ 1.5 packets,
 * levels 1 and 3; the decoder is a restatement of the published format, see csrc/quicklz_decode.hip). */
enum /* DBoW3 WeightingType / ScoringType as stored in the vocabulary stream */
{
    MSLAM_BOW_TF_IDF = 0,
    MSLAM_BOW_TF = 1,
    MSLAM_BOW_IDF = 2,
    MSLAM_BOW_BINARY = 3
};
enum
{
    MSLAM_BOW_L1_NORM = 0,
    MSLAM_BOW_L2_NORM = 1,
    MSLAM_BOW_CHI_SQUARE = 2,
    MSLAM_BOW_KL = 3,
    MSLAM_BOW_BHATTACHARYYA = 4,
    MSLAM_BOW_DOT_PRODUCT = 5
};
/* Only L1_NORM scoring vocabularies are accepted (the ORB vocabularies DBoW3 ships are TF_IDF / L1_NORM).
 * The database (DBoW3::Database(voc, false, 0), orb_relocalizer.cpp:29) is an inverted file on the device: word ->
 * rows of (entry, value).  It is unbounded: storage grows by doubling (mslam_hip_bow_db_reserve pre-allocates);
 * mslam_hip_bow_db_query scores against EVERY entry ever added and not removed.  The batched device form
 * (mslam_hip_bow_batch_dev) scores every frame against the 64 entries that precede it and
 * adds its vectors to the same database. */
int mslam_hip_bow_load(mslam_hip_ctx* ctx, const void* blob, size_t size);
int mslam_hip_bow_info(mslam_hip_ctx* ctx, int* k, int* L, int* n_nodes, int* n_words, int* scoring,
                       int* weighting);
/* How a descriptor is assigned to a word, for every entry point below:
 * TREE = Vocabulary::transform's descent (dbow3.patch:1760-1860: at each level the child with the least Hamming
 * distance, first child on ties) — what DBoW3 does, the default;
 * FLAT = the exhaustive descriptor-vs-vocabulary search the descent approximates (BASELINE.json north_star's
 * "batched descriptor-vs-vocabulary Hamming kernel", SURVEY.md §8d bow_flat): the leaf with the least distance
 * over ALL words, lower word id on ties.  n x n_words distance evaluations: a stress mode, not a DBoW3 drop-in. */
enum
{
    MSLAM_BOW_ASSIGN_TREE = 0,
    MSLAM_BOW_ASSIGN_FLAT = 1
};
int mslam_hip_bow_set_assignment(mslam_hip_ctx* ctx, int mode);
/* Vocabulary::transform(feature, word_id, weight) for n descriptors (dbow3.patch:1760-1860). */
int mslam_hip_bow_words(mslam_hip_ctx* ctx, const uint8_t* desc, int n, uint32_t* word, double* weight);
/* Vocabulary::transform(features, BowVector) (dbow3.patch:1432-1530): ascending word ids, values
 * normalised as the vocabulary's scoring requires.  Capacity of words/values: n. */
int mslam_hip_bow_transform(mslam_hip_ctx* ctx, const uint8_t* desc, int n, uint32_t* words, double* values,
                            int* n_words);
/* DBoW3 L1Scoring::score of two BoW vectors, in [0,1]. */
int mslam_hip_bow_score(mslam_hip_ctx* ctx, const uint32_t* w1, const double* v1, int n1, const uint32_t* w2,
                        const double* v2, int n2, double* score);
/* Database: add(features) -> entry id (IRelocalizer::addKeyframe's feed, rgbd_feature_frontend.cpp:176)
 * and query(features) -> the best max_results entries by L1 score, best first, ties by lower entry
 * id (IRelocalizer::relocalize / ILoopDetector::detectLoop). */
int mslam_hip_bow_db_add(mslam_hip_ctx* ctx, const uint8_t* desc, int n, int* entry_id);
int mslam_hip_bow_db_query(mslam_hip_ctx* ctx, const uint8_t* desc, int n, int max_results, int32_t* entry_ids,
                           double* scores, int* n_results);
/* IRelocalizer::removeKeyframe: the entry is never reported again (its id is not reused). */
int mslam_hip_bow_db_remove(mslam_hip_ctx* ctx, int entry_id);
int mslam_hip_bow_db_clear(mslam_hip_ctx* ctx);
int mslam_hip_bow_db_reserve(mslam_hip_ctx* ctx, int max_entries);
int mslam_hip_bow_db_size(mslam_hip_ctx* ctx, int* n_entries);
/* Batched device form: transform every frame of the last detect batch into a BoW vector, score it
 * against the database (all entries), then add it as a new entry.  Per frame: best entry and score. */
int mslam_hip_bow_batch_dev(mslam_hip_ctx* ctx, int add_to_db);
typedef struct
{
    int32_t capacity;        /* per-frame stride of words/values                */
    const uint32_t* words;   /* [max_batch][capacity]                           */
    const double* values;    /* [max_batch][capacity]                           */
    const int32_t* n_words;  /* [max_batch]                                     */
    const int32_t* best_entry; /* [max_batch] (-1 if the database was empty)    */
    const double* best_score;  /* [max_batch]                                   */
} mslam_hip_bow_view;
int mslam_hip_get_bow_view(mslam_hip_ctx* ctx, mslam_hip_bow_view* view);

/* Cross-stream loop candidates (multi-camera / multi-GPU): score the BoW vector of
 * every frame t of the last mslam_hip_bow_batch_dev against vector [r][t] of n_sets foreign vector
 * sets (e.g. the all-gathered vectors of the other ranks).  Device arrays: d_words / d_values are
 * [n_sets][max_batch][capacity], d_n is [n_sets][max_batch], d_scores (out) is [max_batch][n_sets];
 * a pair without a common word scores exactly 0. */
int mslam_hip_bow_cross_score_dev(mslam_hip_ctx* ctx, const uint32_t* d_words, const double* d_values,
                                  const int32_t* d_n, int n_sets, int capacity, double* d_scores);

/* The exchange step itself, in the wire format of the all-gather (one "set" = one stream's batch):
 *   uint2 {u32 word, f32 value} vec[n_frames][k_max]  (ascending words, zero padded), then int32 count[n_frames]
 *   = n_frames * (2 * k_max + 1) dwords per set; k_max = 2048 gives the 16 KB per frame SURVEY.md §8e sizes.
 * pack: the BoW vectors of the last mslam_hip_bow_batch_dev into d_out (device), on the context's stream; a vector
 * with more than k_max words is reported by mslam_hip_sync as MSLAM_HIP_E_CAPACITY.
 * cross_score_packed: d_sets = n_sets gathered sets; d_scores[t][r] (f64, [n_frames][n_sets]) = L1 score of frame t
 * of set self_set against frame t of set r, on the f32 values as transmitted, summed in ascending word order.  It
 * reads nothing but d_sets and may be enqueued on any stream (`stream` = hipStream_t, NULL = the context's): e.g.
 * the communication stream right behind the collective, while the context's stream extracts the next batch. */
int mslam_hip_bow_pack_dev(mslam_hip_ctx* ctx, int k_max, uint32_t* d_out);
int mslam_hip_bow_cross_score_packed_dev(mslam_hip_ctx* ctx, const uint32_t* d_sets, int n_sets, int self_set,
                                         int n_frames, int k_max, double* d_scores, void* stream);

/* Host-only helper (no GPU, no context): decode `n_packets` consecutive QuickLZ packets, as DBoW3 writes them after the
 * nChunks word of a compressed vocabulary.  dst_size receives the decoded size (also when dst is too small). */
int mslam_hip_qlz_decompress(const void* src, size_t src_size, uint32_t n_packets, void* dst, size_t dst_capacity,
                             size_t* dst_size);

/* ---- RGB-D back-projection (the step after the matcher) ----------------------------
 * Replaces pointsFromRgbdKeypoints / reconstructPoint (rgbd_feature_frontend.cpp:101-138) with getDepth /
 * isDepthValid (types/depth_frame.hpp:20-30).  depth = DepthFrame::data (u16, row-major, width*height),
 * factor / focal / principal point = CameraParameters (sensors/camera_parameters.hpp:7-12; TUM: 1/5000, 525,
 * 525, 319.5, 239.5 — rgbd_file_provider.cpp:136-145).  xy = keypoint coordinates as returned by detect.
 * xyz[3i..3i+2] is the camera-frame point, valid[i] = 1 iff the depth is valid (std::optional engaged). */
int mslam_hip_backproject(mslam_hip_ctx* ctx, const uint16_t* depth, int width, int height, float factor, double fx,
                          double fy, double cx, double cy, const float* xy, int n, double* xyz, uint8_t* valid);
/* Batched device form: every keypoint of the last detect batch against d_depth = n_frames back-to-back
 * u16 depth frames in HBM.  Results: mslam_hip_points_view. */
int mslam_hip_backproject_batch_dev(mslam_hip_ctx* ctx, const uint16_t* d_depth, float factor, double fx, double fy,
                                    double cx, double cy);
typedef struct
{
    int32_t capacity;     /* per-frame stride (in keypoints) */
    const double* xyz;    /* [max_batch][capacity][3]        */
    const uint8_t* valid; /* [max_batch][capacity]           */
} mslam_hip_points_view;
int mslam_hip_get_points_view(mslam_hip_ctx* ctx, mslam_hip_points_view* view);

/* ---- packed results of a batch (one transfer instead of capacity-strided arrays) --------------------------------------
 * The batch views above are [max_batch][max_keypoints]-strided; copied back as they are, more than half of the bytes are
 * padding.  mslam_hip_pack_batch_dev writes, on the context's stream (after the matcher has been joined), a header, the
 * per-frame offset tables and then exactly count[t] keypoint records / match_count[t] match records per frame, back to
 * back, into `out`: device memory (follow with ONE copy of header.bytes) or page-locked, device-mapped host memory (the
 * kernel's stores are the transfer; read the header after synchronising).  Frame t's keypoints are records
 * kp_offset[t] .. kp_offset[t+1]-1 of every keypoint array, its matches records match_offset[t] .. match_offset[t+1]-1
 * (match indices are relative to the frame, as in the views).  A batch that mslam_hip_match_batch_dev has not run on is packed
 * with zero matches (never with an older batch's pairs).  If the results do not fit capacity_bytes, header.fits is 0,
 * header.bytes tells what was needed, no record is written (the two offset tables still are when they fit) and
 * mslam_hip_sync reports MSLAM_HIP_E_CAPACITY.
 * mslam_hip_packed_capacity: an upper bound for n_frames frames (every frame at max_keypoints). */
typedef struct
{
    int32_t n_frames, total_keypoints, total_matches, with_points;
    uint64_t off_kp_offset, off_match_offset;            /* int32[n_frames + 1] each                         */
    uint64_t off_xy, off_desc, off_octave, off_angle, off_response; /* f32[.][2], u8[.][32], i32, f32, f32   */
    uint64_t off_xyz, off_valid;                         /* f64[.][3], u8[.] (with_points)                   */
    uint64_t off_match_from, off_match_to;               /* i32[total_matches] each                          */
    uint64_t bytes;                                      /* bytes used (or needed, when fits == 0)           */
    int32_t fits, pad;
} mslam_hip_packed_header;
int mslam_hip_pack_batch_dev(mslam_hip_ctx* ctx, void* out, size_t capacity_bytes, int with_points);
size_t mslam_hip_packed_capacity(const mslam_hip_ctx* ctx, int n_frames, int with_points);

/* ---- IPnpAlgorithm::solvePnp (the consumer of the matches) ------------------------------------
 * Replaces OpenCvRansacPnp::solvePnp's cv::solvePnPRansac call (cv_ransac_pnp.cpp:56-57: useExtrinsicGuess = true, 100
 * iterations, 5 px, confidence 0.99, no distortion).  object_points = n x 3 f32 (landmark states cast to float, :22-31),
 * image_points = n x 2 f32 (:33-40), pin-hole intrinsics as in :52-53.  rvec / tvec (3 doubles each, Rodrigues vector
 * and translation of the world -> camera transform, OpenCV's convention) are the extrinsic guess on input when
 * use_extrinsic_guess is non-zero, and the result on output; inliers (n bytes, may be NULL) is the consensus mask of
 * the best hypothesis.  Returns MSLAM_HIP_E_NO_MODEL when no hypothesis reaches 4 inliers (solvePnPRansac == false).
 * Against cv::solvePnPRansac as OpenCV 4.8.1 runs it for this call (restated piece by piece in
 * oracle/mslam_cv_pnp_oracle.py from the library's published algorithm), of the four pieces
 *   1 sampler          DEVIATES: splitmix64 counter streams keyed by (`seed`, hypothesis) — hypotheses are independent of
 *                      each other, which is what lets them run in parallel — not cv::RNG((uint64)-1)'s one sequential
 *                      multiply-with-carry stream drawing 5-point subsets;
 *   2 minimal solver   DEVIATES: P3P on three points, the fourth picks the branch — not EPnP on five points;
 *   3 consensus loop   SAME: squared reprojection error <= 5^2 px, a hypothesis replaces the best one only with MORE inliers
 *                      (and at least 4), RANSACUpdateNumIters(confidence, outlier share, 5 model points) after every new
 *                      best one, hypotheses looked at in order (the kernel scores them in parallel rounds and walks each
 *                      round in order);
 *   4 final refit      SAME objective and set (reprojection error over the inliers of the best hypothesis, all in double),
 *                      other minimiser and start: damped Gauss-Newton to convergence from the caller's guess (or the best
 *                      hypothesis) — OpenCV runs <= 20 Levenberg-Marquardt steps from the LAST hypothesis its loop
 *                      evaluated (the callback writes every hypothesis into the guess buffers).
 * So the hypothesis sequence differs, the result the call site consumes (success, consensus set, refined pose,
 * cv_ransac_pnp.cpp:59-83) agrees wherever the consensus set is unambiguous: tests/test_pnp.py compares both entry
 * points with that oracle (masks equal, rvec / tvec within 1e-6 on noise-free scenes with 0 - 60 % outliers; within
 * 0.02 degrees / 2 mm with pixel noise, where borderline points may fall either side of 5 px).  The call site's confidence
 * (0.99, :57) ends the loop as in OpenCV's RANSACPointSetRegistrator: every new best hypothesis lowers the iteration
 * count to log(1 - confidence) / log(1 - w^5) (w = its inlier share; 5 = the model points cv::solvePnPRansac samples for this
 * call's flags, although this library's own minimal sample is 3 + 1 points), hypotheses beyond it are not looked at. */
int mslam_hip_pnp_ransac(mslam_hip_ctx* ctx, const float* object_points, const float* image_points, int n, double fx,
                         double fy, double cx, double cy, int use_extrinsic_guess, int iterations,
                         double reprojection_error, uint64_t seed, double* rvec, double* tvec, uint8_t* inliers,
                         int* n_inliers);

/* The RANSAC confidence of both PnP entry points (default 0.99 = the reference's call, cv_ransac_pnp.cpp:57); a value
 * outside (0, 1) switches the early exit off: all `iterations` hypotheses are scored. */
int mslam_hip_pnp_set_confidence(mslam_hip_ctx* ctx, double confidence);

/* Batched device form (the frame-to-frame tracking step of the RGB-D front end on device data): for every frame t >= 1 of
 * the last batch, the matches (frame t -> frame t-1) of mslam_hip_match_batch_dev whose train keypoint has a valid
 * back-projected point (mslam_hip_backproject_batch_dev) become the 3-D / 2-D correspondences
 *   object = xyz[t-1][to] (cast to float, rgbd_feature_frontend.cpp:232-254 / cv_ransac_pnp.cpp:22-31), image = xy[t][from],
 * in match order, and one RANSAC PnP per frame (one workgroup each, no extrinsic guess, seed + t as the sampling seed)
 * estimates the pose of camera t in the coordinates of camera t-1.  Frame 0 has no predecessor inside the batch
 * (status 0).  Results: mslam_hip_pnp_view. */
int mslam_hip_pnp_batch_dev(mslam_hip_ctx* ctx, double fx, double fy, double cx, double cy, int iterations,
                            double reprojection_error, uint64_t seed);
typedef struct
{
    int32_t capacity;          /* per-frame stride of the correspondence arrays (max_keypoints)                 */
    const double* pose;        /* [max_batch][16]: R row-major (9), t (3), inliers, best hypothesis, status, cost;
                                * status 1 = a model was found, 0 = none (fewer than 4 points / 4 inliers)      */
    const int32_t* n_points;   /* [max_batch] correspondences of the frame                                      */
    const float* object_points; /* [max_batch][capacity][3]                                                     */
    const float* image_points;  /* [max_batch][capacity][2]                                                     */
    const uint8_t* inliers;     /* [max_batch][capacity] consensus mask of the best hypothesis                  */
} mslam_hip_pnp_view;
int mslam_hip_get_pnp_view(mslam_hip_ctx* ctx, mslam_hip_pnp_view* view);

/* ---- test / debug access to intermediate stages (host copies; synchronises) -----------------------*/
enum
{
    MSLAM_HIP_DBG_PYRAMID = 0,    /* u8 [h_l][w_l] unblurred level                               */
    MSLAM_HIP_DBG_BLURRED = 1,    /* u8 [h_l][w_l] 7x7 sigma-2 blurred level                     */
    MSLAM_HIP_DBG_CANDIDATES = 2, /* float triples (x, y, response), border-relative, FAST order */
    MSLAM_HIP_DBG_SELECTED = 3    /* float triples after the quadtree, node-list order           */
};
int mslam_hip_level_geometry(mslam_hip_ctx* ctx, int* widths, int* heights, float* scales);
int mslam_hip_debug_read(mslam_hip_ctx* ctx, int what, int frame, int level, void* dst, size_t dst_bytes,
                         size_t* n_items);

/* Per-(frame, level) counts of the last detect batch (CANDIDATES or SELECTED): the first min(n_frames, frames of the
 * last batch) rows of out[n_frames][n_levels] are written, the rest is left untouched (ABI 3: the row count is an
 * argument; the call used to copy as many rows as the last batch had, whatever the caller had allocated). */
int mslam_hip_debug_counts(mslam_hip_ctx* ctx, int what, int32_t* out, int n_frames);

/* Synchronise the context's stream, then copy `bytes` from a device pointer (e.g. out of a view) to host memory. */
int mslam_hip_copy_to_host(mslam_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* Device timing with HIP events.  enable = 1: every stage of the last detect/match/bow batch, with
 * everything serialised on the context's stream (kernel-by-kernel analysis).  enable = 2: every stage
 * launch, timed in place on the stream it is launched on without changing the schedule; entries accumulate
 * over calls until they are read.  enable = 0: off.  names/ms hold up to cap entries. */
int mslam_hip_set_profiling(mslam_hip_ctx* ctx, int enable);
int mslam_hip_get_stage_times(mslam_hip_ctx* ctx, const char** names, float* ms, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* MSLAM_HIP_H_ */
