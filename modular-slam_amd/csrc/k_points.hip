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


// ---- packed results of a batch (mslam_hip_pack_batch_dev) ---------------------------------------------------------------
// The batch views are capacity-strided ([max_batch][max_keypoints]...): copied back as they are, more than half of what
// crosses PCIe is padding.  k_pack_plan turns the per-frame counts into offsets (one workgroup, a 1024-wide scan) and writes
// the header; k_pack_copy then moves exactly count[t] keypoint records and match_count[t] match records per frame behind it.
// `out` may be device memory (one copy of header.bytes follows) or page-locked, device-mapped host memory (the kernel's
// stores ARE the transfer).
__global__ __launch_bounds__(1024) void k_pack_plan(const int32_t* __restrict__ count, const int32_t* __restrict__ mcount, int n_frames,
                                                    int cap, int with_points, unsigned long long capacity_bytes, uint8_t* __restrict__ out,
                                                    uint32_t* __restrict__ flags)
{
    __shared__ uint32_t wsum[2][16];
    __shared__ uint32_t carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    mslam_hip_packed_header* h = reinterpret_cast<mslam_hip_packed_header*>(out);
    const unsigned long long hdr = (sizeof(mslam_hip_packed_header) + 15) & ~15ull;
    const unsigned long long tab = (((unsigned long long)(n_frames + 1) * 4) + 15) & ~15ull;
    const bool room_for_tables = capacity_bytes >= hdr + 2 * tab;
    int32_t* kp_off = reinterpret_cast<int32_t*>(out + hdr);
    int32_t* m_off = reinterpret_cast<int32_t*>(out + hdr + tab);
    if(tid < 2)
        carry[tid] = 0;
    __syncthreads();
    for(int base = 0; base < n_frames; base += 1024)
    {
        const int t = base + tid;
        uint32_t v[2] = {t < n_frames ? (uint32_t)min(max(count[t], 0), cap) : 0u, (t < n_frames && mcount) ? (uint32_t)min(max(mcount[t], 0), cap) : 0u}; // (mcount == nullptr: no matches of this batch)
        uint32_t inc[2];
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            uint32_t x = v[k];
            for(int o = 1; o < 64; o <<= 1)
            {
                const uint32_t y = (uint32_t)__shfl_up((int)x, o);
                x += lane >= o ? y : 0u;
            }
            inc[k] = x;
            if(lane == 63)
                wsum[k][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for(int k = 0; k < 2; ++k)
        {
            uint32_t pre = carry[k];
            for(int w = 0; w < wave; ++w)
                pre += wsum[k][w];
            if(t < n_frames && room_for_tables)
                (k == 0 ? kp_off : m_off)[t] = (int32_t)(pre + inc[k] - v[k]);
        }
        __syncthreads();
        if(tid < 2)
        {
            uint32_t tot = carry[tid];
            for(int w = 0; w < 16; ++w)
                tot += wsum[tid][w];
            carry[tid] = tot;
        }
        __syncthreads();
    }
    if(tid == 0)
    {
        const unsigned long long nk = carry[0], nm = carry[1];
        auto al = [](unsigned long long x) { return (x + 15) & ~15ull; };
        unsigned long long o = hdr + 2 * tab;
        mslam_hip_packed_header hh;
        hh.n_frames = n_frames, hh.total_keypoints = (int32_t)nk, hh.total_matches = (int32_t)nm, hh.with_points = with_points;
        hh.off_kp_offset = hdr, hh.off_match_offset = hdr + tab;
        hh.off_xy = o, o = al(o + nk * 8);
        hh.off_desc = o, o = al(o + nk * 32);
        hh.off_octave = o, o = al(o + nk * 4);
        hh.off_angle = o, o = al(o + nk * 4);
        hh.off_response = o, o = al(o + nk * 4);
        hh.off_xyz = o, o = al(o + (with_points ? nk * 24 : 0));
        hh.off_valid = o, o = al(o + (with_points ? nk : 0));
        hh.off_match_from = o, o = al(o + nm * 4);
        hh.off_match_to = o, o = al(o + nm * 4);
        hh.bytes = o;
        hh.fits = o <= capacity_bytes ? 1 : 0;
        hh.pad = 0;
        if(capacity_bytes >= sizeof(hh))
            *h = hh;
        if(room_for_tables)
            kp_off[n_frames] = (int32_t)nk, m_off[n_frames] = (int32_t)nm;
        if(!hh.fits)
            atomicOr(flags, kFlagPackOverflow);
    }
}

__global__ __launch_bounds__(256) void k_pack_copy(const float* __restrict__ xy, const uint8_t* __restrict__ desc, const int32_t* __restrict__ octave,
                                                   const float* __restrict__ angle, const float* __restrict__ response,
                                                   const double* __restrict__ xyz, const uint8_t* __restrict__ valid,
                                                   const int32_t* __restrict__ mfrom, const int32_t* __restrict__ mto, int cap,
                                                   uint8_t* __restrict__ out)
{
    const mslam_hip_packed_header* h = reinterpret_cast<const mslam_hip_packed_header*>(out);
    if(!h->fits)
        return;
    const int t = blockIdx.x, tid = blockIdx.y * 256 + threadIdx.x, nt = gridDim.y * 256;
    const int32_t* kp_off = reinterpret_cast<const int32_t*>(out + h->off_kp_offset);
    const int32_t* m_off = reinterpret_cast<const int32_t*>(out + h->off_match_offset);
    const size_t k0 = (size_t)kp_off[t], m0 = (size_t)m_off[t];
    const int n = kp_off[t + 1] - kp_off[t], m = m_off[t + 1] - m_off[t];
    const size_t f = (size_t)t * cap;
    uint2* o_xy = reinterpret_cast<uint2*>(out + h->off_xy) + k0;
    uint2* o_desc = reinterpret_cast<uint2*>(out + h->off_desc) + 4 * k0; // (8-byte pieces: a frame's first record is 32-byte aligned only)
    uint32_t* o_oct = reinterpret_cast<uint32_t*>(out + h->off_octave) + k0;
    uint32_t* o_ang = reinterpret_cast<uint32_t*>(out + h->off_angle) + k0;
    uint32_t* o_resp = reinterpret_cast<uint32_t*>(out + h->off_response) + k0;
    for(int i = tid; i < 4 * n; i += nt)
        o_desc[i] = reinterpret_cast<const uint2*>(desc)[4 * f + i];
    for(int i = tid; i < n; i += nt)
    {
        o_xy[i] = reinterpret_cast<const uint2*>(xy)[f + i];
        o_oct[i] = reinterpret_cast<const uint32_t*>(octave)[f + i];
        o_ang[i] = reinterpret_cast<const uint32_t*>(angle)[f + i];
        o_resp[i] = reinterpret_cast<const uint32_t*>(response)[f + i];
    }
    if(h->with_points)
    {
        double* o_xyz = reinterpret_cast<double*>(out + h->off_xyz) + 3 * k0;
        uint8_t* o_val = out + h->off_valid + k0;
        for(int i = tid; i < 3 * n; i += nt)
            o_xyz[i] = xyz[3 * f + i];
        for(int i = tid; i < n; i += nt)
            o_val[i] = valid[f + i];
    }
    uint32_t* o_mf = reinterpret_cast<uint32_t*>(out + h->off_match_from) + m0;
    uint32_t* o_mt = reinterpret_cast<uint32_t*>(out + h->off_match_to) + m0;
    for(int i = tid; i < m; i += nt)
    {
        o_mf[i] = reinterpret_cast<const uint32_t*>(mfrom)[f + i];
        o_mt[i] = reinterpret_cast<const uint32_t*>(mto)[f + i];
    }
}

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
    c->points_seq = c->detect_seq;
    return MSLAM_HIP_OK;
}

int mslam_hip_pack_batch_dev(mslam_hip_ctx* c, void* out, size_t capacity_bytes, int with_points)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!out || capacity_bytes < sizeof(mslam_hip_packed_header))
        return pfail(c, "pack_batch_dev: no output buffer (at least the header must fit)");
    if(c->n_last < 1)
        return pfail(c, "pack_batch_dev: no detect batch");
    if(with_points && (!c->d_xyz || c->points_seq != c->detect_seq))
        return pfail(c, "pack_batch_dev: with_points, but the last detect batch has not been back-projected");
    PHIPCHK(c, hipSetDevice(c->p.device));
    int rc = mslam_hip_join_matcher(c); // the matches come from the matcher's own stream
    if(rc)
        return rc;
    const size_t K = (size_t)c->p.max_keypoints;
    // the matcher's outputs belong to the batch match_batch_dev last ran on: when that is not this detect batch (match_seq), the
    // slot still holds an older batch's pairs, and the batch is packed with zero matches instead of those
    const int32_t* mcount = c->match_seq == c->detect_seq ? c->d_mcount : nullptr;
    hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(1024), 0, c->stream, c->d_count + 1, mcount, c->n_last, c->p.max_keypoints,
                       with_points ? 1 : 0, (unsigned long long)capacity_bytes, static_cast<uint8_t*>(out), c->d_flags);
    hipLaunchKernelGGL(k_pack_copy, dim3(c->n_last, 4), dim3(256), 0, c->stream, c->d_xy + K * 2, c->d_desc + K * 32, c->d_octave + K,
                       c->d_angle + K, c->d_response + K, c->d_xyz, c->d_valid, c->d_mfrom, c->d_mto, c->p.max_keypoints,
                       static_cast<uint8_t*>(out));
    PHIPCHK(c, hipGetLastError());
    return MSLAM_HIP_OK;
}

size_t mslam_hip_packed_capacity(const mslam_hip_ctx* c, int n_frames, int with_points)
{
    if(!c || n_frames < 0)
        return 0;
    const size_t K = (size_t)c->p.max_keypoints, n = (size_t)n_frames;
    return 256 + 2 * (((n + 1) * 4 + 15) & ~(size_t)15) + n * K * (8 + 32 + 4 + 4 + 4 + (with_points ? 25 : 0) + 8) + 16 * 16;
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
