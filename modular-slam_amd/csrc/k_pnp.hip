// k_pnp.hip — RANSAC PnP: the consumer of the matches.
//
// Replaces OpenCvRansacPnp::solvePnp (reference cv_ransac_pnp.cpp:14-85), i.e.
//   cv::solvePnPRansac(objectPoints f32, imagePoints f32, K, noArray(), rvec, tvec, useExtrinsicGuess = true,
//                      iterationsCount = 100, reprojectionError = 5.0, confidence = 0.99, inliers)   (:56-57)
// Contract kept: pin-hole camera without distortion, at most `iterations` hypotheses from random minimal samples, a
// point is an inlier iff its squared reprojection error is <= reprojectionError^2, the pose returned is refined on the
// consensus set of the best hypothesis (most inliers, first on ties) starting from the caller's guess when
// useExtrinsicGuess is set, and the inlier mask is that of the best hypothesis.
// What differs from OpenCV's internals (not in the reference tree; parity with it is UNPINNED — DESIGN.md): the minimal
// solver is P3P (Grunert's quartic, 3 points + 1 to disambiguate) instead of EPnP on 5 points, samples come from a
// counter-based generator (splitmix64) instead of cv::RNG, and the refinement is a damped Gauss-Newton on (rotation,
// translation) instead of
// cvFindExtrinsicCameraParams2's LM on (rvec, tvec).  The confidence of the call site (0.99, :57) is honoured the way
// RANSACPointSetRegistrator::run does it: hypotheses are looked at in their order, every new best one lowers the
// number of iterations to log(1 - confidence) / log(1 - w^m) (w = its inlier share, m = 5: the model points cv::solvePnPRansac samples for this call, kPnpModelPoints below; OpenCV's
// RANSACUpdateNumIters, cvRound and all), and the loop ends there — the winner is the best of the hypotheses
// 0 .. niters-1, exactly what the sequential loop returns.  tests/ pin it against ground-truth poses and an
// independent numpy oracle.
//
// One workgroup solves one problem: hypotheses are generated one per thread, scored against every point by the
// whole workgroup, and the refinement's normal equations are reduced in a fixed order (bit-reproducible).
// All arithmetic is f64: + - * / sqrt and include/mslam_sincos.h.
#include "context.hpp"
#include "../../include/mslam_sincos.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace mslam
{

struct PnpArgs
{
    const float* obj; // [n][3]
    const float* img; // [n][2]
    int n;
    double fx, fy, cx, cy;
    double R0[9], t0[3]; // the caller's guess (use_guess)
    int use_guess;
    int iterations;
    double confidence; // the RANSAC loop ends once enough hypotheses were looked at for this confidence (>= 1 or <= 0: never early)
    double thr2; // reprojectionError^2
    unsigned long long seed;
    double* hyp;     // [iterations][12] R (row major), t
    int32_t* counts; // [iterations]
    uint8_t* mask;   // [n]
    double* out;     // R[9], t[3], n_inliers, best hypothesis, status
};

__device__ __forceinline__ unsigned long long splitmix(unsigned long long& x)
{
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct V3
{
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 mulR(const double* R, V3 p)
{
    return v3(R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z, R[6] * p.x + R[7] * p.y + R[8] * p.z);
}

// orthonormal frame of a triangle: e1 along (b - a), e3 normal, e2 = e3 x e1; false when degenerate
__device__ bool tri_frame(V3 a, V3 b, V3 c, V3& e1, V3& e2, V3& e3)
{
    const V3 ab = b - a, ac = c - a;
    const double l1 = sqrt(dot(ab, ab));
    if(!(l1 > 1e-12))
        return false;
    e1 = (1.0 / l1) * ab;
    const V3 nrm = cross(e1, ac);
    const double l3 = sqrt(dot(nrm, nrm));
    if(!(l3 > 1e-12))
        return false;
    e3 = (1.0 / l3) * nrm;
    e2 = cross(e3, e1);
    return true;
}

// real roots of x^4 + a x^3 + b x^2 + c x + d (Ferrari; the resolvent root by bisection, roots polished by Newton)
__device__ int quartic_roots(double a, double b, double c, double d, double* roots)
{
    const double a2 = a * a;
    const double p = b - 0.375 * a2;
    const double q = c - 0.5 * a * b + 0.125 * a2 * a;
    const double r = d - 0.25 * a * c + 0.0625 * a2 * b - (3.0 / 256.0) * a2 * a2;
    int n = 0;
    double y[4];
    const double tiny = 1e-14 * (1.0 + fabs(p) * sqrt(fabs(p)) + fabs(q));
    if(fabs(q) <= tiny)
    {
        // biquadratic: y^2 = (-p +- sqrt(p^2 - 4 r)) / 2
        const double disc = p * p - 4.0 * r;
        if(disc >= 0.0)
        {
            const double sq = sqrt(disc);
            const double z0 = 0.5 * (-p + sq), z1 = 0.5 * (-p - sq);
            if(z0 >= 0.0)
                y[n++] = sqrt(z0), y[n++] = -sqrt(z0);
            if(z1 >= 0.0)
                y[n++] = sqrt(z1), y[n++] = -sqrt(z1);
        }
    }
    else
    {
        // a positive root m of m^3 + p m^2 + (p^2/4 - r) m - q^2/8: g(0) < 0, g(M) > 0
        const double c1 = 0.25 * p * p - r, c0 = -0.125 * q * q;
        double lo = 0.0, hi = 1.0 + fmax(fabs(p), fmax(fabs(c1), fabs(c0)));
        for(int it = 0; it < 100; ++it)
        {
            const double m = 0.5 * (lo + hi);
            const double g = ((m + p) * m + c1) * m + c0;
            if(g > 0.0)
                hi = m;
            else
                lo = m;
        }
        const double m = 0.5 * (lo + hi);
        const double s = sqrt(2.0 * m);
        const double h = q / (2.0 * s);
        const double k0 = 0.5 * p + m;
        // y^2 - s y + (k0 + h) = 0 and y^2 + s y + (k0 - h) = 0
        const double d1 = s * s - 4.0 * (k0 + h), d2 = s * s - 4.0 * (k0 - h);
        if(d1 >= 0.0)
        {
            const double sq = sqrt(d1);
            y[n++] = 0.5 * (s + sq), y[n++] = 0.5 * (s - sq);
        }
        if(d2 >= 0.0)
        {
            const double sq = sqrt(d2);
            y[n++] = 0.5 * (-s + sq), y[n++] = 0.5 * (-s - sq);
        }
    }
    for(int i = 0; i < n; ++i)
    {
        double x = y[i] - 0.25 * a;
        for(int it = 0; it < 3; ++it)
        {
            const double f = (((x + a) * x + b) * x + c) * x + d;
            const double fp = ((4.0 * x + 3.0 * a) * x + 2.0 * b) * x + c;
            if(fp != 0.0)
                x -= f / fp;
        }
        roots[i] = x;
    }
    return n;
}

struct Cam
{
    double fx, fy, cx, cy;
};

__device__ __forceinline__ bool project(const Cam& k, const double* R, const double* t, V3 P, double& u, double& v)
{
    const V3 X = mulR(R, P) + v3(t[0], t[1], t[2]);
    if(!(X.z > 1e-9))
        return false;
    u = k.fx * X.x / X.z + k.cx;
    v = k.fy * X.y / X.z + k.cy;
    return true;
}

// inlier test of point P / pixel (pu, pv) under (R, t): reprojection error <= thr, written without the two divisions
// (for X.z > 0:  (fx X.x / X.z + cx - pu)^2 + (fy X.y / X.z + cy - pv)^2 <= thr^2
//            <=> (fx X.x + (cx - pu) X.z)^2 + (fy X.y + (cy - pv) X.z)^2 <= thr^2 X.z^2);
// the scoring loop evaluates it 100 x n times per problem and an f64 division is ~25 instructions.  The CPU checker of
// the tests evaluates the same form.
__device__ __forceinline__ bool is_inlier(const Cam& k, const double* R, const double* t, V3 P, double pu, double pv, double thr2)
{
    const V3 X = mulR(R, P) + v3(t[0], t[1], t[2]);
    if(!(X.z > 1e-9))
        return false;
    const double eu = k.fx * X.x + (k.cx - pu) * X.z, ev = k.fy * X.y + (k.cy - pv) * X.z;
    return eu * eu + ev * ev <= thr2 * (X.z * X.z);
}

// P3P on points 0..2 of the sample (Grunert's quartic), the 4th point picks among the solutions
__device__ bool p3p_hypothesis(const Cam& k, const V3* P, const double* uv, double* R, double* t)
{
    V3 f[3];
    for(int i = 0; i < 3; ++i)
    {
        const V3 d = v3((uv[2 * i] - k.cx) / k.fx, (uv[2 * i + 1] - k.cy) / k.fy, 1.0);
        f[i] = (1.0 / sqrt(dot(d, d))) * d;
    }
    const V3 d23 = P[1] - P[2], d13 = P[0] - P[2], d12 = P[0] - P[1];
    const double a2 = dot(d23, d23), b2 = dot(d13, d13), c2 = dot(d12, d12);
    if(!(a2 > 1e-18 && b2 > 1e-18 && c2 > 1e-18))
        return false;
    const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
    const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2;
    const double A4 = (k1 - 1.0) * (k1 - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (k1 * (1.0 - k1) * cb - (1.0 - k2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (k1 * k1 - 1.0 + 2.0 * k1 * k1 * cb * cb + 2.0 * ((b2 - c2) / b2) * ca * ca - 4.0 * k2 * ca * cb * cg +
                             2.0 * ((b2 - a2) / b2) * cg * cg);
    const double A1 = 4.0 * (-k1 * (1.0 + k1) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - k2) * ca * cg);
    const double A0 = (1.0 + k1) * (1.0 + k1) - 4.0 * a2 / b2 * cg * cg;
    if(!(fabs(A4) > 1e-14))
        return false;
    double roots[4];
    const int nr = quartic_roots(A3 / A4, A2 / A4, A1 / A4, A0 / A4, roots);
    V3 e1, e2, e3;
    if(!tri_frame(P[0], P[1], P[2], e1, e2, e3))
        return false;
    double best = 1e300;
    bool found = false;
    for(int i = 0; i < nr; ++i)
    {
        const double v = roots[i];
        const double den = 2.0 * (cg - v * ca);
        if(!(v > 0.0) || !(fabs(den) > 1e-14))
            continue;
        const double u = ((k1 - 1.0) * v * v - 2.0 * k1 * cb * v + 1.0 + k1) / den;
        const double s1d = 1.0 + v * v - 2.0 * v * cb;
        if(!(u > 0.0) || !(s1d > 1e-14))
            continue;
        const double s1 = sqrt(b2 / s1d);
        const V3 X0 = s1 * f[0], X1 = (u * s1) * f[1], X2 = (v * s1) * f[2];
        V3 g1, g2, g3;
        if(!tri_frame(X0, X1, X2, g1, g2, g3))
            continue;
        // R maps the world triangle frame onto the camera one: R = [g1 g2 g3] [e1 e2 e3]^T
        double Rc[9];
        Rc[0] = g1.x * e1.x + g2.x * e2.x + g3.x * e3.x, Rc[1] = g1.x * e1.y + g2.x * e2.y + g3.x * e3.y, Rc[2] = g1.x * e1.z + g2.x * e2.z + g3.x * e3.z;
        Rc[3] = g1.y * e1.x + g2.y * e2.x + g3.y * e3.x, Rc[4] = g1.y * e1.y + g2.y * e2.y + g3.y * e3.y, Rc[5] = g1.y * e1.z + g2.y * e2.z + g3.y * e3.z;
        Rc[6] = g1.z * e1.x + g2.z * e2.x + g3.z * e3.x, Rc[7] = g1.z * e1.y + g2.z * e2.y + g3.z * e3.y, Rc[8] = g1.z * e1.z + g2.z * e2.z + g3.z * e3.z;
        const V3 tt = X0 - mulR(Rc, P[0]);
        const double tc[3] = {tt.x, tt.y, tt.z};
        double pu, pv;
        if(!project(k, Rc, tc, P[3], pu, pv))
            continue;
        const double e = (pu - uv[6]) * (pu - uv[6]) + (pv - uv[7]) * (pv - uv[7]);
        if(e < best)
        {
            best = e;
            found = true;
            for(int j = 0; j < 9; ++j)
                R[j] = Rc[j];
            t[0] = tc[0], t[1] = tc[1], t[2] = tc[2];
        }
    }
    return found;
}

// OpenCV's RANSACUpdateNumIters(p, ep, modelPoints, maxIters) (calib3d/src/ptsetreg.cpp): the number of samples after which
// an all-inlier sample has been drawn with probability p when a share ep of the points are outliers
__device__ __forceinline__ int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    if(!(p > 0.0) || !(p < 1.0))
        return max_iters; // no early exit asked for
    ep = fmax(ep, 0.0);
    ep = fmin(ep, 1.0);
    double num = fmax(1.0 - p, 2.2250738585072014e-308);
    double w = 1.0 - ep, wm = 1.0;
    for(int i = 0; i < model_points; ++i)
        wm *= w; // pow(1 - ep, modelPoints) for a small integer exponent, in a fixed order
    double denom = 1.0 - wm;
    if(denom < 2.2250738585072014e-308)
        return 0;
    num = log(num);
    denom = log(denom);
    if(denom >= 0 || -num >= max_iters * (-denom))
        return max_iters;
    return (int)rint(num / denom); // cvRound
}

// modelPoints of the iteration-count formula: cv::solvePnPRansac called with the default flags (SOLVEPNP_ITERATIVE,
// cv_ransac_pnp.cpp:56-57) samples 5 points per hypothesis when there are more than 4 (its minimal solver is then EPnP), so its
// loop ends after log(1 - p) / log(1 - w^5) samples.  This library's hypotheses come from 3 + 1 points (P3P), but the stopping
// rule follows the call site: with 4 the loop would stop earlier than the reference's (17 instead of 25 hypotheses at 30 %
// outliers).
constexpr int kPnpModelPoints = 5;
constexpr int kPnpThreads = 256;

// fixed-order workgroup sum of `cnt` doubles per thread (acc[cnt]); the totals land in red[0..cnt).  Butterfly inside
// each wave (cross-lane moves, no memory), then the four wave totals are added in wave order: one barrier pair instead
// of a nine-barrier LDS tree, and 1.2 KB of LDS instead of 57 KB (which held the batched kernel to 2 workgroups per CU).
constexpr int kPnpRed = 32 + 28 * (kPnpThreads / 64); // doubles of LDS: totals [32] + per-wave partials [28][waves]
constexpr int kPnpLdsPts = 2048;                      // points staged in LDS for the scoring loop (40 KB)
constexpr int kPnpLdsHyp = 256;                       // hypotheses (12 doubles + a count) kept in LDS up to this many
template <int CNT>
__device__ __forceinline__ void wg_sum(double* acc, double* red, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int W = kPnpThreads / 64;
    // CNT independent butterflies, fully unrolled: the cross-lane moves of different sums overlap
#pragma unroll
    for(int o = 32; o > 0; o >>= 1)
    {
#pragma unroll
        for(int j = 0; j < CNT; ++j)
            acc[j] += __shfl_xor(acc[j], o);
    }
    if(lane == 0)
    {
#pragma unroll
        for(int j = 0; j < CNT; ++j)
            red[32 + j * W + wave] = acc[j];
    }
    __syncthreads();
    if(tid < CNT)
    {
        double t = red[32 + tid * W];
#pragma unroll
        for(int w = 1; w < W; ++w)
            t += red[32 + tid * W + w];
        red[tid] = t;
    }
    __syncthreads();
}

// one problem, one workgroup (shared by the single-problem kernel and the batched one)
__device__ __forceinline__ void pnp_problem(const PnpArgs& a, double* red /* LDS [kPnpRed] doubles + [kPnpLdsPts][5] floats */)
{
    __shared__ int s_best, s_cnt, s_niters;
    __shared__ double sR[9], st[3], sNew[12];
    __shared__ double s_lambda, s_cost;
    __shared__ int s_stop;
    const int tid = threadIdx.x;
    const Cam cam{a.fx, a.fy, a.cx, a.cy};
    const int n = a.n;
    // hypotheses and their inlier counts: LDS when they fit (every wave re-reads them; a global round trip per hypothesis
    // was a third of the scoring time), else the caller's global arrays
    double* l_hyp = red + kPnpRed;
    int32_t* l_cnt = reinterpret_cast<int32_t*>(l_hyp + kPnpLdsHyp * 12);
    float* pts = reinterpret_cast<float*>(l_cnt + kPnpLdsHyp);
    uint8_t* l_mask = reinterpret_cast<uint8_t*>(pts + 5 * kPnpLdsPts);
    double* hyp = a.iterations <= kPnpLdsHyp ? l_hyp : a.hyp;
    int32_t* counts = a.iterations <= kPnpLdsHyp ? l_cnt : a.counts;

    // ---- 1. hypotheses: thread h draws 4 distinct points and solves P3P
    for(int h = tid; h < a.iterations; h += kPnpThreads)
    {
        unsigned long long x = a.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(h + 1);
        int idx[4];
        bool ok = n >= 4;
        for(int k = 0; k < 4 && ok; ++k)
        {
            int tries = 0;
            for(;;)
            {
                idx[k] = (int)(splitmix(x) % (unsigned long long)n);
                bool dup = false;
                for(int j = 0; j < k; ++j)
                    dup = dup || idx[j] == idx[k];
                if(!dup)
                    break;
                if(++tries > 64)
                {
                    ok = false;
                    break;
                }
            }
        }
        double R[9], t[3];
        if(ok)
        {
            V3 P[4];
            double uv[8];
            for(int k = 0; k < 4; ++k)
            {
                P[k] = v3((double)a.obj[3 * idx[k]], (double)a.obj[3 * idx[k] + 1], (double)a.obj[3 * idx[k] + 2]);
                uv[2 * k] = (double)a.img[2 * idx[k]];
                uv[2 * k + 1] = (double)a.img[2 * idx[k] + 1];
            }
            ok = p3p_hypothesis(cam, P, uv, R, t);
        }
        double* hp = hyp + (size_t)h * 12;
        for(int j = 0; j < 9; ++j)
            hp[j] = ok ? R[j] : 0.0;
        for(int j = 0; j < 3; ++j)
            hp[9 + j] = ok ? t[j] : 0.0;
        counts[h] = ok ? 0 : -1;
    }
    __syncthreads();

    // ---- 2. score every hypothesis against every point: wave w takes hypotheses w, w+4, ...
    // (the first kPnpLdsPts points as 5 floats each in LDS: every wave reads every point once per hypothesis)
    const int n_lds = min(n, kPnpLdsPts);
    for(int i = tid; i < n_lds; i += kPnpThreads)
    {
        pts[5 * i] = a.obj[3 * i], pts[5 * i + 1] = a.obj[3 * i + 1], pts[5 * i + 2] = a.obj[3 * i + 2];
        pts[5 * i + 3] = a.img[2 * i], pts[5 * i + 4] = a.img[2 * i + 1];
    }
    __syncthreads();
    // Rounds of one hypothesis per wave; after every round thread 0 walks the round's hypotheses in order, exactly as the
    // sequential RANSAC loop would (a new best hypothesis lowers s_niters; a hypothesis at or beyond s_niters is never
    // looked at), so a clean scene ends after a round or two instead of `iterations` hypotheses.
    if(tid == 0)
        s_best = -1, s_cnt = -1, s_niters = a.iterations;
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int W = kPnpThreads / 64;
        for(int h0 = 0; h0 < s_niters; h0 += W)
        {
            const int h = h0 + wave;
            if(h < a.iterations && counts[h] >= 0)
            {
            // the hypothesis in registers (wave-uniform), the points from LDS when they fit
            double hp[12];
            for(int j = 0; j < 12; ++j)
                hp[j] = hyp[(size_t)h * 12 + j];
            int c = 0;
            // four points per lane and step: with one wave per SIMD (the P3P code takes 174 VGPRs) only independent work
            // inside the wave hides the f64 and LDS latencies.  Two loops so that the LDS one is pure ds_read code.
            if(n <= n_lds)
            {
                auto one = [&](int i) -> int {
                    if(i >= n)
                        return 0;
                    const float* po = pts + 5 * i;
                    return is_inlier(cam, hp, hp + 9, v3((double)po[0], (double)po[1], (double)po[2]), (double)po[3],
                                     (double)po[4], a.thr2) ? 1 : 0;
                };
                for(int i = lane; i < n; i += 256)
                    c += one(i) + one(i + 64) + one(i + 128) + one(i + 192);
            }
            else
            {
                for(int i = lane; i < n; i += 64)
                    c += is_inlier(cam, hp, hp + 9, v3((double)a.obj[3 * i], (double)a.obj[3 * i + 1], (double)a.obj[3 * i + 2]),
                                   (double)a.img[2 * i], (double)a.img[2 * i + 1], a.thr2) ? 1 : 0;
            }
            for(int o = 32; o > 0; o >>= 1)
                c += __shfl_xor(c, o);
            if(lane == 0)
                counts[h] = c;
            }
            __syncthreads();
            if(tid == 0)
            {
                int best = s_best, bc = s_cnt, niters = s_niters;
                for(int k = 0; k < W && h0 + k < niters; ++k)
                {
                    const int hh = h0 + k;
                    // RANSACPointSetRegistrator::run: goodCount > max(maxGoodCount, modelPoints - 1)
                    if(counts[hh] > max(bc, 3))
                    {
                        bc = counts[hh], best = hh;
                        niters = ransac_update_iters(a.confidence, (double)(n - bc) / (double)n, kPnpModelPoints, niters);
                    }
                }
                s_best = best, s_cnt = bc, s_niters = niters;
            }
            __syncthreads();
        }
    }
    const int best = s_best;
    if(best < 0 || s_cnt < 4)
    {
        if(tid == 0)
        {
            for(int j = 0; j < 12; ++j)
                a.out[j] = 0.0;
            a.out[12] = 0.0, a.out[13] = -1.0, a.out[14] = 0.0; // status 0: no model
        }
        for(int i = tid; i < n; i += kPnpThreads)
            a.mask[i] = 0;
        return;
    }
    // ---- 3. consensus set of the best hypothesis
    {
        const double* hp = hyp + (size_t)best * 12;
        for(int i = tid; i < n; i += kPnpThreads)
        {
            const V3 P = v3((double)a.obj[3 * i], (double)a.obj[3 * i + 1], (double)a.obj[3 * i + 2]);
            const bool in = is_inlier(cam, hp, hp + 9, P, (double)a.img[2 * i], (double)a.img[2 * i + 1], a.thr2);
            a.mask[i] = in ? 1 : 0;
            if(i < n_lds)
                l_mask[i] = in ? 1 : 0;
        }
        if(tid < 9)
            sR[tid] = a.use_guess ? a.R0[tid] : hp[tid];
        if(tid < 3)
            st[tid] = a.use_guess ? a.t0[tid] : hp[9 + tid];
        if(tid == 0)
            s_lambda = 1e-3, s_stop = 0;
    }
    __syncthreads();

    // ---- 4. damped Gauss-Newton on the consensus set: X = exp(w) R P + t + dt
    // point i of the consensus set (false: not in it); from LDS when the problem fits (workgroup-uniform choice)
    const bool in_lds = n <= n_lds;
    auto fetch = [&](int i, V3& P, double& pu, double& pv) -> bool {
        if(in_lds)
        {
            if(!l_mask[i])
                return false;
            const float* po = pts + 5 * i;
            P = v3((double)po[0], (double)po[1], (double)po[2]);
            pu = (double)po[3], pv = (double)po[4];
            return true;
        }
        if(!a.mask[i])
            return false;
        P = v3((double)a.obj[3 * i], (double)a.obj[3 * i + 1], (double)a.obj[3 * i + 2]);
        pu = (double)a.img[2 * i], pv = (double)a.img[2 * i + 1];
        return true;
    };
    auto cost_of = [&](const double* R, const double* t) -> double {
        double c = 0.0;
        for(int i = tid; i < n; i += kPnpThreads)
        {
            V3 P;
            double pu, pv, u, v;
            if(!fetch(i, P, pu, pv))
                continue;
            if(project(cam, R, t, P, u, v))
            {
                const double du = u - pu, dv = v - pv;
                c += du * du + dv * dv;
            }
            else
                c += 1e12; // behind the camera
        }
        wg_sum<1>(&c, red, tid);
        const double total = red[0];
        __syncthreads();
        return total;
    };
    double cost = cost_of(sR, st);
    for(int it = 0; it < 30; ++it)
    {
        double acc[27]; // upper triangle of J^T J (21) and J^T r (6)
        for(int j = 0; j < 27; ++j)
            acc[j] = 0.0;
        for(int i = tid; i < n; i += kPnpThreads)
            {
                V3 P;
                double pu, pv;
                if(!fetch(i, P, pu, pv))
                    continue;
                const V3 Y = mulR(sR, P); // rotated, not translated
                const V3 X = Y + v3(st[0], st[1], st[2]);
                if(!(X.z > 1e-9))
                    continue;
                const double iz = 1.0 / X.z;
                const double ru = cam.fx * X.x * iz + cam.cx - pu;
                const double rv = cam.fy * X.y * iz + cam.cy - pv;
                // d(u, v)/dX, dX/dw = -[Y]x, dX/dt = I
                const double ux = cam.fx * iz, uz = -cam.fx * X.x * iz * iz, vy = cam.fy * iz, vz = -cam.fy * X.y * iz * iz;
                const double Ju[6] = {uz * Y.y, ux * Y.z - uz * Y.x, -ux * Y.y, ux, 0.0, uz};
                const double Jv[6] = {-vy * Y.z + vz * Y.y, -vz * Y.x, vy * Y.x, 0.0, vy, vz};
                int k = 0;
                for(int r = 0; r < 6; ++r)
                    for(int c2 = r; c2 < 6; ++c2)
                        acc[k++] += Ju[r] * Ju[c2] + Jv[r] * Jv[c2];
                for(int r = 0; r < 6; ++r)
                    acc[21 + r] += Ju[r] * ru + Jv[r] * rv;
            }
        wg_sum<27>(acc, red, tid);
        if(tid == 0)
        {
            // 6 x 6 damped normal equations, solved in registers (every loop below has constant bounds and is unrolled; with
            // early exits the arrays lived in scratch memory and this single-thread section took most of an iteration)
            double H[6][6], g[6];
            {
                int k = 0;
#pragma unroll
                for(int r = 0; r < 6; ++r)
#pragma unroll
                    for(int c2 = r; c2 < 6; ++c2)
                    {
                        H[r][c2] = red[k];
                        H[c2][r] = red[k];
                        ++k;
                    }
            }
#pragma unroll
            for(int r = 0; r < 6; ++r)
                g[r] = red[21 + r];
#pragma unroll
            for(int r = 0; r < 6; ++r)
                H[r][r] += s_lambda * H[r][r] + 1e-12;
            // Cholesky H = L L^T, solve H d = -g
            double L[6][6];
            bool ok = true;
#pragma unroll
            for(int r = 0; r < 6; ++r)
#pragma unroll
                for(int c2 = 0; c2 <= r; ++c2)
                {
                    double sum = H[r][c2];
#pragma unroll
                    for(int j = 0; j < c2; ++j)
                        sum -= L[r][j] * L[c2][j];
                    if(r == c2)
                    {
                        ok = ok && sum > 0.0;
                        L[r][r] = sqrt(ok ? sum : 1.0);
                    }
                    else
                        L[r][c2] = sum / L[c2][c2];
                }
            double d[6] = {0, 0, 0, 0, 0, 0};
            {
                double y[6];
#pragma unroll
                for(int r = 0; r < 6; ++r)
                {
                    double sum = -g[r];
#pragma unroll
                    for(int j = 0; j < r; ++j)
                        sum -= L[r][j] * y[j];
                    y[r] = sum / L[r][r];
                }
#pragma unroll
                for(int r = 5; r >= 0; --r)
                {
                    double sum = y[r];
#pragma unroll
                    for(int j = r + 1; j < 6; ++j)
                        sum -= L[j][r] * d[j];
                    d[r] = sum / L[r][r];
                }
                if(!ok)
                {
#pragma unroll
                    for(int r = 0; r < 6; ++r)
                        d[r] = 0.0;
                }
            }
            // R' = exp(w) R (Rodrigues), t' = t + dt
            const double th2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const double th = sqrt(th2);
            double A, B; // sin(th)/th, (1 - cos(th))/th^2
            if(th < 1e-4)
                A = 1.0 - th2 / 6.0, B = 0.5 - th2 / 24.0;
            else
            {
                double sn, cs;
                mslam_sincos_f64(th, &sn, &cs);
                A = sn / th, B = (1.0 - cs) / th2;
            }
            const double wx = d[0], wy = d[1], wz = d[2];
            const double E[9] = {1.0 - B * (wy * wy + wz * wz), -A * wz + B * wx * wy, A * wy + B * wx * wz,
                                 A * wz + B * wx * wy, 1.0 - B * (wx * wx + wz * wz), -A * wx + B * wy * wz,
                                 -A * wy + B * wx * wz, A * wx + B * wy * wz, 1.0 - B * (wx * wx + wy * wy)};
            for(int r = 0; r < 3; ++r)
                for(int c2 = 0; c2 < 3; ++c2)
                    sNew[r * 3 + c2] = E[r * 3] * sR[c2] + E[r * 3 + 1] * sR[3 + c2] + E[r * 3 + 2] * sR[6 + c2];
            sNew[9] = st[0] + d[3], sNew[10] = st[1] + d[4], sNew[11] = st[2] + d[5];
            s_cost = sqrt(th2 + d[3] * d[3] + d[4] * d[4] + d[5] * d[5]); // step length
        }
        __syncthreads();
        const double step = s_cost;
        const double trial = cost_of(sNew, sNew + 9);
        if(tid == 0)
        {
            if(trial < cost)
            {
                for(int j = 0; j < 9; ++j)
                    sR[j] = sNew[j];
                st[0] = sNew[9], st[1] = sNew[10], st[2] = sNew[11];
                s_lambda = fmax(s_lambda * 0.1, 1e-12);
                s_stop = (step < 1e-12 || cost - trial <= 1e-14 * cost) ? 1 : 0;
            }
            else
            {
                s_lambda *= 10.0;
                // a rejected step shorter than 1e-9 means the optimum is reached to rounding: without this test the loop
                // spends its remaining iterations raising lambda to 1e12
                s_stop = (s_lambda > 1e12 || step < 1e-9) ? 1 : 0;
            }
        }
        if(trial < cost)
            cost = trial;
        __syncthreads();
        if(s_stop)
            break;
    }
    if(tid == 0)
    {
        for(int j = 0; j < 9; ++j)
            a.out[j] = sR[j];
        a.out[9] = st[0], a.out[10] = st[1], a.out[11] = st[2];
        a.out[12] = (double)s_cnt;
        a.out[13] = (double)best;
        a.out[14] = 1.0;
        a.out[15] = cost;
    }
}

__global__ __launch_bounds__(kPnpThreads) void k_pnp_ransac(PnpArgs a)
{
    extern __shared__ double red[]; // [kPnpRed]
    pnp_problem(a, red);
}

// ---- batched form: correspondences of every frame from the device-resident match / back-projection results ---------
struct PnpBatchArgs
{
    PnpArgs proto;        // camera, iterations, threshold, seed; pointers = frame 0's slices
    const int32_t* n;     // [B] correspondences per frame
    int cap;              // per-frame stride (keypoints)
};

// frame t: matches (from = keypoint of t, to = keypoint of t-1) with a valid 3-D point at `to`, in match order
__global__ __launch_bounds__(256) void k_pnp_gather(const int32_t* __restrict__ mfrom, const int32_t* __restrict__ mto,
                                                    const int32_t* __restrict__ mcount, const float* __restrict__ xy,
                                                    const double* __restrict__ xyz, const uint8_t* __restrict__ valid, int cap,
                                                    float* __restrict__ obj, float* __restrict__ img, int32_t* __restrict__ n_out)
{
    __shared__ uint32_t wsum[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = t >= 1 ? min(mcount[t], cap) : 0;
    const size_t ft = (size_t)t * cap, fp = (size_t)(t - 1) * cap;
    uint32_t running = 0;
    for(int base = 0; base < m; base += 256)
    {
        const int i = base + tid;
        int from = 0, to = 0;
        bool ok = false;
        if(i < m)
        {
            from = mfrom[ft + i], to = mto[ft + i];
            ok = (unsigned)from < (unsigned)cap && (unsigned)to < (unsigned)cap && valid[fp + to] != 0;
        }
        const unsigned long long b = __ballot(ok);
        if(lane == 0)
            wsum[wave] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for(int k = 0; k < 4; ++k)
        {
            pre += k < wave ? wsum[k] : 0;
            tot += wsum[k];
        }
        if(ok)
        {
            const size_t o = ft + running + pre + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            const double* P = xyz + (fp + to) * 3;
            obj[o * 3] = (float)P[0], obj[o * 3 + 1] = (float)P[1], obj[o * 3 + 2] = (float)P[2];
            img[o * 2] = xy[(ft + from) * 2], img[o * 2 + 1] = xy[(ft + from) * 2 + 1];
        }
        running += tot;
        __syncthreads();
    }
    if(tid == 0)
        n_out[t] = (int32_t)running;
}

__global__ __launch_bounds__(kPnpThreads) void k_pnp_ransac_batch(PnpBatchArgs b)
{
    extern __shared__ double red[]; // [kPnpRed]
    const int t = blockIdx.x;
    PnpArgs a = b.proto;
    a.n = b.n[t];
    a.obj += (size_t)t * b.cap * 3;
    a.img += (size_t)t * b.cap * 2;
    a.mask += (size_t)t * b.cap;
    a.hyp += (size_t)t * a.iterations * 12;
    a.counts += (size_t)t * a.iterations;
    a.out += (size_t)t * 16;
    a.seed += (unsigned long long)t;
    if(a.n < 4)
    {
        if(threadIdx.x < 16)
            a.out[threadIdx.x] = threadIdx.x == 13 ? -1.0 : 0.0; // status 0: no model
        return;
    }
    pnp_problem(a, red);
}

} // namespace mslam

using namespace mslam;

static void rodrigues_to_R(const double r[3], double R[9])
{
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if(th < 1e-12)
    {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for(int i = 0; i < 9; ++i)
            R[i] = I[i];
        return;
    }
    const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th, c = std::cos(th), s = std::sin(th), v = 1 - c;
    R[0] = c + kx * kx * v, R[1] = kx * ky * v - kz * s, R[2] = kx * kz * v + ky * s;
    R[3] = ky * kx * v + kz * s, R[4] = c + ky * ky * v, R[5] = ky * kz * v - kx * s;
    R[6] = kz * kx * v - ky * s, R[7] = kz * ky * v + kx * s, R[8] = c + kz * kz * v;
}

static void R_to_rodrigues(const double R[9], double r[3])
{
    const double tr = R[0] + R[4] + R[8];
    double c = 0.5 * (tr - 1.0);
    c = c > 1 ? 1 : c < -1 ? -1 : c;
    const double th = std::acos(c);
    const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];
    const double s2 = std::sqrt(ax * ax + ay * ay + az * az); // 2 sin(th)
    if(s2 > 1e-9)
    {
        r[0] = th * ax / s2, r[1] = th * ay / s2, r[2] = th * az / s2;
        return;
    }
    if(c > 0)
    {
        r[0] = 0.5 * ax, r[1] = 0.5 * ay, r[2] = 0.5 * az; // small angle
        return;
    }
    // angle near pi: axis from the diagonal
    double k[3] = {std::sqrt(std::fmax(0.0, 0.5 * (R[0] + 1))), std::sqrt(std::fmax(0.0, 0.5 * (R[4] + 1))),
                   std::sqrt(std::fmax(0.0, 0.5 * (R[8] + 1)))};
    if(R[1] + R[3] < 0)
        k[1] = -k[1];
    if(R[2] + R[6] < 0)
        k[2] = -k[2];
    r[0] = th * k[0], r[1] = th * k[1], r[2] = th * k[2];
}

// both kernels take more than 64 KB of dynamic LDS: the attribute belongs to the (function, device) pair, so it is set once
// per CONTEXT (after hipSetDevice), not once per process
static hipError_t pnp_lds_attributes(mslam_hip_ctx* c, size_t lds)
{
    if(c->pnp_attr_set)
        return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_ransac), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if(e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pnp_ransac_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    c->pnp_attr_set = e == hipSuccess;
    return e;
}

extern "C" int mslam_hip_pnp_ransac(mslam_hip_ctx* c, const float* object_points, const float* image_points, int n, double fx,
                                    double fy, double cx, double cy, int use_extrinsic_guess, int iterations,
                                    double reprojection_error, uint64_t seed, double* rvec, double* tvec, uint8_t* inliers,
                                    int* n_inliers)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    auto fail = [&](int code, const char* msg) {
        c->err = msg;
        return code;
    };
    if(!object_points || !image_points || !rvec || !tvec || n < 4 || iterations < 1 || iterations > 4096 ||
       !(reprojection_error > 0) || !(fx != 0.0) || !(fy != 0.0))
        return fail(MSLAM_HIP_E_INVALID, "pnp_ransac: bad argument (at least 4 points, 1..4096 iterations)");
    if(n_inliers)
        *n_inliers = 0;
    hipError_t e = hipSetDevice(c->p.device);
#define PCHK(call)                                                                                                     \
    if(e == hipSuccess)                                                                                                \
    e = (call)
    // scratch of the single-problem call lives in the context and only grows (this is the per-frame tracking path of the
    // plugin: six hipMalloc + six hipFree, each a device synchronisation, per call would cost more than the solve)
    if(n > c->pnp1_n_cap || iterations > c->pnp1_it_cap)
    {
        PCHK(hipStreamSynchronize(c->stream));
        void* old[] = {c->d_pnp1_obj, c->d_pnp1_img, c->d_pnp1_hyp, c->d_pnp1_out, c->d_pnp1_counts, c->d_pnp1_mask};
        for(void* b : old)
            if(b)
                (void)hipFree(b);
        c->d_pnp1_obj = c->d_pnp1_img = nullptr;
        c->d_pnp1_hyp = c->d_pnp1_out = nullptr;
        c->d_pnp1_counts = nullptr;
        c->d_pnp1_mask = nullptr;
        c->pnp1_n_cap = c->pnp1_it_cap = 0;
        const int n_cap = std::max(n, 1024), it_cap = std::max(iterations, 128);
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_obj), (size_t)n_cap * 12));
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_img), (size_t)n_cap * 8));
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_hyp), (size_t)it_cap * 12 * 8));
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_out), 16 * 8));
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_counts), (size_t)it_cap * 4));
        PCHK(hipMalloc(reinterpret_cast<void**>(&c->d_pnp1_mask), (size_t)n_cap));
        if(e == hipSuccess)
            c->pnp1_n_cap = n_cap, c->pnp1_it_cap = it_cap;
    }
    float *d_obj = c->d_pnp1_obj, *d_img = c->d_pnp1_img;
    double *d_hyp = c->d_pnp1_hyp, *d_out = c->d_pnp1_out;
    int32_t* d_counts = c->d_pnp1_counts;
    uint8_t* d_mask = c->d_pnp1_mask;
    PCHK(hipMemcpyAsync(d_obj, object_points, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
    PCHK(hipMemcpyAsync(d_img, image_points, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    PnpArgs a{};
    a.obj = d_obj, a.img = d_img, a.n = n;
    a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy;
    a.use_guess = use_extrinsic_guess ? 1 : 0;
    rodrigues_to_R(rvec, a.R0);
    a.t0[0] = tvec[0], a.t0[1] = tvec[1], a.t0[2] = tvec[2];
    a.iterations = iterations;
    a.thr2 = reprojection_error * reprojection_error;
    a.confidence = c->pnp_confidence;
    a.seed = seed;
    a.hyp = d_hyp, a.counts = d_counts, a.mask = d_mask, a.out = d_out;
    const size_t lds = (size_t)kPnpRed * 8 + (size_t)kPnpLdsHyp * (96 + 4) + (size_t)kPnpLdsPts * 21;
    PCHK(pnp_lds_attributes(c, lds));
    if(e == hipSuccess)
        hipLaunchKernelGGL(k_pnp_ransac, dim3(1), dim3(kPnpThreads), lds, c->stream, a);
    PCHK(hipGetLastError());
    double out[16] = {0};
    std::vector<uint8_t> mask((size_t)n);
    PCHK(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    PCHK(hipMemcpyAsync(mask.data(), d_mask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PCHK(hipStreamSynchronize(c->stream));
#undef PCHK
    if(e != hipSuccess)
    {
        c->err = std::string("pnp_ransac: ") + hipGetErrorString(e);
        return MSLAM_HIP_E_RUNTIME;
    }
    if(out[14] != 1.0)
        return fail(MSLAM_HIP_E_NO_MODEL, "pnp_ransac: no hypothesis reached 4 inliers");
    R_to_rodrigues(out, rvec);
    tvec[0] = out[9], tvec[1] = out[10], tvec[2] = out[11];
    if(inliers)
        std::memcpy(inliers, mask.data(), (size_t)n);
    if(n_inliers)
        *n_inliers = (int)out[12];
    return MSLAM_HIP_OK;
}


static int pnp_batch_buffers(mslam_hip_ctx* c, int iterations)
{
    const size_t B = (size_t)c->p.max_batch, K = (size_t)c->p.max_keypoints;
    if(c->d_pnp_obj && c->pnp_iterations >= iterations)
        return MSLAM_HIP_OK;
    void* old[] = {c->d_pnp_obj, c->d_pnp_img, c->d_pnp_n, c->d_pnp_counts, c->d_pnp_hyp, c->d_pnp_out, c->d_pnp_mask};
    for(void* p : old)
        if(p)
            (void)hipFree(p);
    c->d_pnp_obj = c->d_pnp_img = nullptr, c->d_pnp_n = c->d_pnp_counts = nullptr, c->d_pnp_hyp = c->d_pnp_out = nullptr;
    c->d_pnp_mask = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_obj), B * K * 12);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_img), B * K * 8);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_n), B * 4);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_counts), B * (size_t)iterations * 4);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_hyp), B * (size_t)iterations * 12 * 8);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_out), B * 16 * 8);
    if(e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_pnp_mask), B * K);
    if(e == hipSuccess) e = hipMemset(c->d_pnp_n, 0, B * 4);
    if(e == hipSuccess) e = hipMemset(c->d_pnp_out, 0, B * 16 * 8);
    if(e != hipSuccess)
    {
        c->err = std::string("pnp_batch_dev: ") + hipGetErrorString(e);
        return MSLAM_HIP_E_RUNTIME;
    }
    c->pnp_iterations = iterations;
    return MSLAM_HIP_OK;
}

extern "C" int mslam_hip_pnp_batch_dev(mslam_hip_ctx* c, double fx, double fy, double cx, double cy, int iterations,
                                       double reprojection_error, uint64_t seed)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    auto fail = [&](int code, const char* msg) {
        c->err = msg;
        return code;
    };
    if(iterations < 1 || iterations > 4096 || !(reprojection_error > 0) || !(fx != 0.0) || !(fy != 0.0))
        return fail(MSLAM_HIP_E_INVALID, "pnp_batch_dev: bad argument (1..4096 iterations)");
    if(c->n_last < 1 || !c->d_xyz)
        return fail(MSLAM_HIP_E_INVALID, "pnp_batch_dev: needs a detect batch, its matches and its back-projection");
    if(hipSetDevice(c->p.device) != hipSuccess)
        return fail(MSLAM_HIP_E_RUNTIME, "pnp_batch_dev: hipSetDevice");
    int rc = pnp_batch_buffers(c, iterations);
    if(rc)
        return rc;
    rc = mslam_hip_join_matcher(c); // the matches come from the matcher's stream
    if(rc)
        return rc;
    const int K = c->p.max_keypoints, n = c->n_last;
    {
        StageScope t(c, "pnp_gather");
        hipLaunchKernelGGL(k_pnp_gather, dim3(n), dim3(256), 0, c->stream, c->d_mfrom, c->d_mto, c->d_mcount,
                           c->d_xy + (size_t)K * 2, c->d_xyz, c->d_valid, K, c->d_pnp_obj, c->d_pnp_img, c->d_pnp_n);
    }
    PnpBatchArgs b{};
    PnpArgs& a = b.proto;
    a.obj = c->d_pnp_obj, a.img = c->d_pnp_img, a.n = 0;
    a.fx = fx, a.fy = fy, a.cx = cx, a.cy = cy;
    a.use_guess = 0;
    a.iterations = iterations;
    a.thr2 = reprojection_error * reprojection_error;
    a.confidence = c->pnp_confidence;
    a.seed = seed;
    a.hyp = c->d_pnp_hyp, a.counts = c->d_pnp_counts, a.mask = c->d_pnp_mask, a.out = c->d_pnp_out;
    b.n = c->d_pnp_n;
    b.cap = K;
    const size_t lds = (size_t)kPnpRed * 8 + (size_t)kPnpLdsHyp * (96 + 4) + (size_t)kPnpLdsPts * 21;
    if(pnp_lds_attributes(c, lds) != hipSuccess)
        return fail(MSLAM_HIP_E_RUNTIME, "pnp_batch_dev: hipFuncSetAttribute");
    {
        StageScope t(c, "pnp_ransac");
        hipLaunchKernelGGL(k_pnp_ransac_batch, dim3(n), dim3(kPnpThreads), lds, c->stream, b);
    }
    if(hipGetLastError() != hipSuccess)
        return fail(MSLAM_HIP_E_RUNTIME, "pnp_batch_dev: launch failed");
    return MSLAM_HIP_OK;
}

extern "C" int mslam_hip_get_pnp_view(mslam_hip_ctx* c, mslam_hip_pnp_view* v)
{
    if(!c || !v)
        return MSLAM_HIP_E_INVALID;
    if(!c->d_pnp_obj)
    {
        c->err = "get_pnp_view: mslam_hip_pnp_batch_dev has not run";
        return MSLAM_HIP_E_INVALID;
    }
    v->capacity = c->p.max_keypoints;
    v->pose = c->d_pnp_out;
    v->n_points = c->d_pnp_n;
    v->object_points = c->d_pnp_obj;
    v->image_points = c->d_pnp_img;
    v->inliers = c->d_pnp_mask;
    return MSLAM_HIP_OK;
}

extern "C" int mslam_hip_pnp_set_confidence(mslam_hip_ctx* c, double confidence)
{
    if(!c)
        return MSLAM_HIP_E_INVALID;
    if(!(confidence == confidence))
        return MSLAM_HIP_E_INVALID; // NaN
    c->pnp_confidence = confidence;
    return MSLAM_HIP_OK;
}
