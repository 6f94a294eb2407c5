/* mslam_oracle.h — CPU ORACLE for the ORB / Hamming-match / BoW hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; the shipped path (modular-slam_amd/) never does.
 *
 * It is a plain-C, single-threaded restatement of the reference algorithm
 * (marcin-ochman/modular-slam @ 2025-08-24).  Each function cites the reference file:line it
 * follows.  The OpenCV 4.8.1 primitives the reference calls (cv::FAST, cv::resize, cv::GaussianBlur,
 * cv::fastAtan2, BFMatcher::knnMatch) and DBoW3 are NOT vendored in the reference, so those are
 * restated from their published algorithms and anchored on the reference call sites.
 *
 * PARITY STATUS: "parity unpinned" — the reference holds no golden vector, known-answer test or
 * fixture for this path and cannot be compiled here (OpenCV/Eigen/Boost/DBoW3
 * absent).  Goldens under tests/golden/ are restatement goldens.
 */
#ifndef MSLAM_ORACLE_H_
#define MSLAM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSO_MAX_LEVELS 16
#define MSO_PATCH_RADIUS 19 /* orb_patch_radius_, distributed_cv_feature.cpp:699 */

typedef struct
{
    int n_levels;          /* 8    distributed_cv_feature.cpp:1184 */
    float scale_factor;    /* 1.2f distributed_cv_feature.cpp:1184 */
    int ini_fast_thr;      /* 20   */
    int min_fast_thr;      /* 7    */
    unsigned min_size;     /* 1000 distributed_cv_feature.cpp:1186 (min node area, named max_num_keypts there) */
} mso_orb_params;

typedef struct
{
    float x, y;     /* level coordinates, relative to the level's 19-px border origin for candidates */
    float response; /* FAST score */
} mso_cand;

void mso_default_params(mso_orb_params* p);

/* frame.cpp:6-27 */
void mso_gray(const uint8_t* bgr, size_t n_px, uint8_t* gray);

/* distributed_cv_feature.cpp:411-420 (scale chain) and :836-837 (level sizes) */
void mso_level_geometry(int W, int H, const mso_orb_params* p, int* w, int* h, float* scale);

/* cv::resize(INTER_LINEAR) on CV_8UC1, call site distributed_cv_feature.cpp:839 */
void mso_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);

/* resize coefficient tables exactly as cv::resize builds them (for host-table parity tests) */
void mso_resize_tables(int ssize, int dsize, int32_t* ofs, int16_t* coef /* 2*dsize */);

/* cv::FAST(img, kps, threshold, nonmaxSuppression=true), TYPE_9_16; call sites :918,:924.
 * Returns the number of keypoints written (row-major order), at most cap. */
int mso_fast(const uint8_t* img, int step, int cols, int rows, int threshold, mso_cand* out, int cap);

/* compute_fast_keypoints for ONE level up to (not including) the quadtree:
 * distributed_cv_feature.cpp:858-952.  Coordinates are relative to (19,19). */
int mso_fast_level(const uint8_t* img, int cols, int rows, const mso_orb_params* p, mso_cand* out, int cap);

/* distribute_keypoints_via_tree + initialize_nodes + divide_node + assign_child_nodes +
 * find_keypoints_with_max_response: distributed_cv_feature.cpp:981-1155, :306-355.
 * min_x..max_y are the border rectangle; scale_factor is scale_factors_[level]. */
int mso_quadtree(const mso_cand* in, int n, int min_x, int max_x, int min_y, int max_y, float scale_factor,
                 unsigned min_size, mso_cand* out, int cap);

/* cv::fastAtan2 (degrees) */
float mso_fast_atan2(float y, float x);
/* util::cos / util::sin, distributed_cv_feature.cpp:456-503 */
float mso_util_cos(float v);
float mso_util_sin(float v);
/* orb_impl ctor u_max_ table, distributed_cv_feature.cpp:522-541 */
void mso_umax(int* umax /* 16 */);
/* ic_angle, distributed_cv_feature.cpp:543-570 */
float mso_ic_angle(const uint8_t* img, int step, int x, int y);

/* cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8U; call site :797-798 */
void mso_gaussian_kernel_fixed(int taps[7]);
void mso_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst);

/* compute_orb_descriptor, distributed_cv_feature.cpp:572-629 (scalar branch :603-607) */
void mso_orb_descriptor(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc);

/* whole DistributedOrbOpenCvDetector::detect (:1190-1222) minus its GUI side effects.
 * Outputs are SoA; xy are the scale-corrected float coordinates (:1166-1179).  Returns 0, or -1 if
 * more than max_out keypoints were produced (n_out then holds the full count). */
int mso_detect(const uint8_t* bgr, int W, int H, const mso_orb_params* p, int max_out, float* xy, uint8_t* desc,
               int32_t* octave, float* angle, float* response, int* n_out);

/* BFMatcher(NORM_HAMMING).knnMatch(query=to, train=from, k=2): orb_feature.cpp:96.
 * Per query q: idx0/idx1 = train indices (or -1), d0/d1 = distances (or INT32_MAX). */
void mso_match_knn2_raw(const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to, int32_t* idx0,
                        int32_t* idx1, int32_t* d0, int32_t* d1);
/* OrbOpenCvMatcher::match, orb_feature.cpp:84-117 (ratio test :99-105, output order :110-114).
 * n_from < 2 is UB in the reference; here it yields no matches. */
int mso_match(const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to, double ratio,
              int32_t* from_idx, int32_t* to_idx);

/* pointsFromRgbdKeypoints + reconstructPoint, rgbd_feature_frontend.cpp:101-138, with getDepth /
 * isDepthValid, types/depth_frame.hpp:20-30.  xy are the (float) keypoint coordinates, widened to double as
 * the detector adapter does (distributed_cv_feature.cpp:1207-1208). */
void mso_backproject(const uint16_t* depth, int width, int height, float factor, double fx, double fy, double cx,
                     double cy, const float* xy, int n, double* xyz, uint8_t* valid);

/* ---- the cv::ORB detector mode: OrbOpenCvDetector (orb_feature.cpp:25,33-65) ---------------------------------
 * = toGrayScale + cv::ORB::create(1000)->detectAndCompute.  Everything below the call site is OpenCV 4.8.1
 * (features2d/src/orb.cpp, imgproc resize INTER_LINEAR_EXACT, KeyPointsFilter), not in the reference tree:
 * restated from the published algorithm, PARITY UNPINNED.
 * Notes:
 *  - output ORDER: retainBest() uses std::nth_element + std::partition, so the order inside a level is whatever the C++
 *    library of the build does; the kept SET is defined by the standard (response >= the n-th largest, ties kept).  The
 *    reference is built with GCC: `order` = MSO_ORDER_LIBSTDCXX (default) reproduces libstdc++'s introselect / partition
 *    step by step (pinned against the real <algorithm> of this image); MSO_ORDER_RASTER keeps FAST's raster order (y, x);
 *  - cos/sin of the keypoint angle are the correctly rounded float values ((float)cos((double)a) from the host libm's
 *    double routines) instead of the host libm's cosf/sinf, whose last bit is implementation-defined; the product
 *    evaluates include/mslam_sincos.h, an independent implementation of the same correctly rounded values. */
typedef struct
{
    int n_features;     /* 1000  orb_feature.cpp:25            */
    float scale_factor; /* 1.2f  cv::ORB::create default       */
    int n_levels;       /* 8                                   */
    int edge_threshold; /* 31                                  */
    int fast_threshold; /* 20                                  */
    int order;          /* MSO_ORDER_LIBSTDCXX (default) / MSO_ORDER_RASTER: where retainBest leaves the survivors */
} mso_cvorb_params;
enum { MSO_ORDER_LIBSTDCXX = 0, MSO_ORDER_RASTER = 1 };
/* test hook: KeyPointsFilter::retainBest's std::nth_element + std::partition as libstdc++ runs them, on n responses;
 * order[] = payload indices of the survivors in their final places, returns their number */
int mso_std_retain_best_order(const float* response, int n, int n_points, int32_t* order);
int mso_std_heap_select_calls(void); /* how often introselect's depth-limit branch (heap_select) has run */
void mso_cvorb_default_params(mso_cvorb_params* p);
/* layerScale / layer sizes / per-level feature quota (orb.cpp getScale, detectAndCompute, computeKeyPoints) */
void mso_cvorb_geometry(int W, int H, const mso_cvorb_params* p, int* w, int* h, float* scale, int* quota);
/* cv::resize(..., INTER_LINEAR_EXACT) on CV_8UC1 (imgproc/src/resize.cpp resize_bitExact, ufixedpoint16) */
void mso_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
/* HarrisResponses (orb.cpp), blockSize 7, k 0.04, at integer (x, y) of one level image */
float mso_harris_response(const uint8_t* img, int step, int x, int y);
/* computeOrbDescriptors (orb.cpp), WTA_K 2, on the blurred level */
void mso_cvorb_descriptor(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc);
/* stages for the parity tests: keypoints of one level after FAST + border filter + retainBest(2n) [stage 0] or
 * after Harris + retainBest(n) [stage 1]; raster order; response = FAST score resp. Harris response */
int mso_cvorb_level_keypoints(const uint8_t* img, int w, int h, const mso_cvorb_params* p, int quota, int stage,
                              mso_cand* out, int cap);
/* whole OrbOpenCvDetector::detect; same output contract as mso_detect */
int mso_cvorb_detect(const uint8_t* bgr, int W, int H, const mso_cvorb_params* p, int max_out, float* xy, uint8_t* desc,
                     int32_t* octave, float* angle, float* response, int* n_out);

void mso_sincos_f32(float x, float* s, float* c);   /* include/mslam_sincos.h (the product's routine, host build: test hook only) */
void mso_libm_sincosf(float x, float* s, float* c); /* host libm sinf / cosf */

/* ---- DBoW3 (rmsalinas/DBow3 master, conan_recipes/dbow3/conanfile.py:9,18) ---------------- */
typedef struct mso_voc mso_voc;
enum { MSO_TF_IDF = 0, MSO_TF = 1, MSO_IDF = 2, MSO_BINARY = 3 };
enum { MSO_L1_NORM = 0, MSO_L2_NORM = 1, MSO_CHI_SQUARE = 2, MSO_KL = 3, MSO_BHATTACHARYYA = 4, MSO_DOT_PRODUCT = 5 };

/* Vocabulary::fromStream, dbow3.patch:2544-2651 (uncompressed streams only) */
mso_voc* mso_voc_load(const void* blob, size_t size);
void mso_voc_free(mso_voc* v);
int mso_voc_info(const mso_voc* v, int* k, int* L, int* n_nodes, int* n_words, int* scoring, int* weighting);
/* Vocabulary::transform(feature, word_id, weight), dbow3.patch:1760-1860 */
void mso_bow_words(const mso_voc* v, const uint8_t* desc, int n, uint32_t* word, double* weight);
/* exhaustive assignment over all words (SURVEY.md §8d bow_flat), lower word id on ties */
void mso_bow_words_flat(const mso_voc* v, const uint8_t* desc, int n, uint32_t* word, double* weight);
/* Vocabulary::transform(features, BowVector), dbow3.patch:1432-1530; ascending word order. */
int mso_bow_vector(const mso_voc* v, const uint8_t* desc, int n, uint32_t* words, double* values);
/* DBoW3 L1Scoring::score (published algorithm; not in the reference tree) */
double mso_bow_score_l1(const uint32_t* w1, const double* v1, int n1, const uint32_t* w2, const double* v2, int n2);

/* ---- the timed CPU leg (mslam_cpu_bench.c): n_threads pthreads, each one mso_detect + mso_match loop over its own block
 * of frames_per_thread frames of the stream (rgbd_feature_frontend.cpp:187,237: detect, then match against the previous
 * frame).  cvp non-NULL selects the cv::ORB detector mode.  out[5] = keypoints, matches, wall seconds, shortest and
 * longest thread loop.  Returns 0, -1 when a frame exceeded max_kp, -2 when the threads could not be started. */
int mso_bench_stream(const uint8_t* frames, int n_unique, int W, int H, const mso_orb_params* p,
                     const mso_cvorb_params* cvp, int n_threads, int frames_per_thread, int max_kp, double out[5]);

#ifdef __cplusplus
}
#endif
#endif /* MSLAM_ORACLE_H_ */
