// k_bow.hip — DBoW3 bag-of-words: vocabulary tree descent, BoW vectors, L1 scoring, database.
// (first slice: entry points exist and fail loudly until a vocabulary is loaded)
#include "context.hpp"

namespace mslam
{
struct BowState
{
    int dummy;
};
void bow_destroy(BowState* b) { delete b; }
int bow_batch(mslam_hip_ctx* c, int) { c->err = "no vocabulary loaded"; return MSLAM_HIP_E_NO_VOCABULARY; }
} // namespace mslam

static int novoc(mslam_hip_ctx* c)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    c->err = "no vocabulary loaded";
    return MSLAM_HIP_E_NO_VOCABULARY;
}

extern "C" {
int mslam_hip_bow_load(mslam_hip_ctx* c, const void*, size_t) { if(!c) return MSLAM_HIP_E_INVALID; c->err = "bow_load: not built yet"; return MSLAM_HIP_E_FORMAT; }
int mslam_hip_bow_info(mslam_hip_ctx* c, int*, int*, int*, int*, int*, int*) { return novoc(c); }
int mslam_hip_bow_words(mslam_hip_ctx* c, const uint8_t*, int, uint32_t*, double*) { return novoc(c); }
int mslam_hip_bow_transform(mslam_hip_ctx* c, const uint8_t*, int, uint32_t*, double*, int*) { return novoc(c); }
int mslam_hip_bow_score(mslam_hip_ctx* c, const uint32_t*, const double*, int, const uint32_t*, const double*, int, double*) { return novoc(c); }
int mslam_hip_bow_db_add(mslam_hip_ctx* c, const uint8_t*, int, int*) { return novoc(c); }
int mslam_hip_bow_db_query(mslam_hip_ctx* c, const uint8_t*, int, int, int32_t*, double*, int*) { return novoc(c); }
int mslam_hip_bow_db_clear(mslam_hip_ctx* c) { return novoc(c); }
int mslam_hip_bow_batch_dev(mslam_hip_ctx* c, int) { return novoc(c); }
int mslam_hip_get_bow_view(mslam_hip_ctx* c, mslam_hip_bow_view*) { return novoc(c); }
}
