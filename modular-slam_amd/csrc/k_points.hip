// k_points.hip — RGB-D back-projection of keypoints (the step right after the matcher).
//
// Replaces pointsFromRgbdKeypoints + reconstructPoint (reference rgbd_feature_frontend.cpp:101-138) with
// getDepth / isDepthValid (types/depth_frame.hpp:20-30):
//   imgPoint = coordinates.cast<int>()                      (truncation of the double coordinates)
//   depth    = (float)data[w*y + x] * factor                (float multiply; TUM factor = 1/5000, rgbd_file_provider.cpp:136-145)
//   valid    = depth > FLT_EPSILON
//   X = (x - cx) * z * (1/fx),  Y = (y - cy) * z * (1/fy),  Z = z      (double, left to right; z = (double)depth)
// One lane per keypoint; pure element-wise f64 arithmetic (products only, so nothing can contract).
#include "context.hpp"

#include <cfloat>

namespace mslam
{

struct Camera
{
    double cx, cy, inv_fx, inv_fy;
    float factor;
};

__global__ __launch_bounds__(256) void k_backproject(const uint16_t* __restrict__ depth, long long depth_stride, int w, int h,
                                                     Camera cam, const float* __restrict__ xy, long long xy_stride,
                                                     const int32_t* __restrict__ counts, int n_fixed, int cap,
                                                     double* __restrict__ xyz, uint8_t* __restrict__ valid)
{
    const int frame = blockIdx.y;
    const int n = min(counts ? counts[frame] : n_fixed, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i >= n)
        return;
    const float* p = xy + frame * xy_stride + 2 * (size_t)i;
    const double x = (double)p[0], y = (double)p[1]; // the adapter widens the float coordinates (:1207-1208)
    const int ix = (int)x, iy = (int)y;
    float d = 0.f;
    if(ix >= 0 && ix < w && iy >= 0 && iy < h)
        d = __fmul_rn((float)depth[frame * depth_stride + (size_t)w * iy + ix], cam.factor);
    const bool ok = d > FLT_EPSILON;
    const size_t o = (size_t)frame * cap + i;
    const double z = (double)d;
    xyz[3 * o] = ok ? (x - cam.cx) * z * cam.inv_fx : 0.0;
    xyz[3 * o + 1] = ok ? (y - cam.cy) * z * cam.inv_fy : 0.0;
    xyz[3 * o + 2] = ok ? z : 0.0;
    valid[o] = ok ? 1 : 0;
}

} // namespace mslam

using namespace mslam;

#define PHIPCHK(c, call)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if(e_ != hipSuccess)                                                                                           \
        {                                                                                                              \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                              \
            return MSLAM_HIP_E_RUNTIME;                                                                                \
        }                                                                                                              \
    } while(0)

static int pfail(mslam_hip_ctx* c, const char* m)
{
    c->err = m;
    return MSLAM_HIP_E_INVALID;
}

static int ensure_points(mslam_hip_ctx* c)
{
    if(c->d_xyz)
        return MSLAM_HIP_OK;
    const size_t n = (size_t)c->p.max_batch * c->p.max_keypoints;
    PHIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_xyz), n * 3 * sizeof(double)));
    PHIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->d_valid), n));
    return MSLAM_HIP_OK;
}

extern "C" {

int mslam_hip_backproject_batch_dev(mslam_hip_ctx* c, const uint16_t* d_depth, float factor, double fx, double fy,
                                    double cx, double cy)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!d_depth || !(fx != 0.0) || !(fy != 0.0))
        return pfail(c, "backproject_batch_dev: bad argument");
    if(c->n_last < 1)
        return pfail(c, "backproject_batch_dev: no detect batch");
    PHIPCHK(c, hipSetDevice(c->p.device));
    int rc = ensure_points(c);
    if(rc)
        return rc;
    const size_t K = (size_t)c->p.max_keypoints;
    const Camera cam{cx, cy, 1.0 / fx, 1.0 / fy, factor}; // invFocal = 1.0 / focal (:126)
    StageScope t(c, "backproject");
    dim3 grid((c->p.max_keypoints + 255) / 256, c->n_last);
    hipLaunchKernelGGL(k_backproject, grid, dim3(256), 0, c->stream, d_depth, (long long)c->p.width * c->p.height,
                       c->p.width, c->p.height, cam, c->d_xy + K * 2, (long long)K * 2, c->d_count + 1, 0,
                       c->p.max_keypoints, c->d_xyz, c->d_valid);
    PHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

int mslam_hip_get_points_view(mslam_hip_ctx* c, mslam_hip_points_view* v)
{
    if(!c || !v)
        return MSLAM_HIP_E_INVALID;
    int rc = ensure_points(c);
    if(rc)
        return rc;
    v->capacity = c->p.max_keypoints;
    v->xyz = c->d_xyz;
    v->valid = c->d_valid;
    return MSLAM_HIP_OK;
}

int mslam_hip_backproject(mslam_hip_ctx* c, const uint16_t* depth, int width, int height, float factor, double fx,
                          double fy, double cx, double cy, const float* xy, int n, double* xyz, uint8_t* valid)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!depth || width <= 0 || height <= 0 || n < 0 || (n > 0 && (!xy || !xyz || !valid)) || !(fx != 0.0) || !(fy != 0.0))
        return pfail(c, "backproject: bad argument");
    if(n == 0)
        return MSLAM_HIP_OK;
    PHIPCHK(c, hipSetDevice(c->p.device));
    uint16_t* d_depth = nullptr;
    float* d_xy = nullptr;
    double* d_xyz = nullptr;
    uint8_t* d_valid = nullptr;
    const size_t npx = (size_t)width * height;
    hipStream_t s = c->stream;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_depth), npx * 2);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_xy), (size_t)n * 8);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_xyz), (size_t)n * 24);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_valid), (size_t)n);
    if(e == hipSuccess) e = hipMemcpyAsync(d_depth, depth, npx * 2, hipMemcpyHostToDevice, s);
    if(e == hipSuccess) e = hipMemcpyAsync(d_xy, xy, (size_t)n * 8, hipMemcpyHostToDevice, s);
    if(e == hipSuccess)
    {
        const Camera cam{cx, cy, 1.0 / fx, 1.0 / fy, factor};
        hipLaunchKernelGGL(k_backproject, dim3((n + 255) / 256, 1), dim3(256), 0, s, d_depth, 0ll, width, height, cam, d_xy,
                           0ll, nullptr, n, n, d_xyz, d_valid);
        e = hipGetLastError();
    }
    if(e == hipSuccess) e = hipMemcpyAsync(xyz, d_xyz, (size_t)n * 24, hipMemcpyDeviceToHost, s);
    if(e == hipSuccess) e = hipMemcpyAsync(valid, d_valid, (size_t)n, hipMemcpyDeviceToHost, s);
    if(e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_depth);
    (void)hipFree(d_xy);
    (void)hipFree(d_xyz);
    (void)hipFree(d_valid);
    if(e != hipSuccess)
    {
        c->err = std::string("backproject: ") + hipGetErrorString(e);
        return MSLAM_HIP_E_RUNTIME;
    }
    return MSLAM_HIP_OK;
}

} // extern "C"
