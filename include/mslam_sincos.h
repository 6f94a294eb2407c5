/* mslam_sincos.h — the cos/sin used by the cv::ORB detector mode (orb_feature.cpp:25,40; OpenCV orb.cpp
 * computeOrbDescriptors: `angle *= (float)(CV_PI/180.f); float a = (float)cos(angle), b = (float)sin(angle);`).
 *
 * WORKING HYPOTHESIS (unpinned: no OpenCV build exists in this image; tests/test_opencv_pin.py skips as PARITY UNPINNED until
 * an opencv_dump golden file is installed): inside namespace cv the unqualified cos(angle) of a float resolves, in a GCC /
 * libstdc++ build, to the C library's DOUBLE function (the float overloads of <cmath> live in namespace std; cv's using-list
 * does not take them) — the (float) casts in the source point the same way — i.e. the reference computes
 * (float)cos((double)angle).  The argument is fragile: libstdc++'s <math.h> wrapper does `using std::cos;`, so if any header of
 * the orb.cpp translation unit includes <math.h> instead of <cmath>, cos(float) is the FLOAT overload = cosf, which differs in
 * the last bit for 0.14 % of the arguments (below).  Until the pin exists this stays a DOCUMENTED POSSIBLE DEVIATION of the
 * cv::ORB mode; oracle/opencv_check/compare.py tolerates it with a bound and counts it.  A GPU cannot call the host library, so this
 * routine evaluates cos / sin of the float angle in IEEE double arithmetic (Cody-Waite reduction to [-pi/4, pi/4], Taylor
 * polynomials to r^16 / r^15: absolute error < 1e-15 for |x| <= 8) and rounds the result to float.  Every operation is a
 * single rounded double add / multiply / floor in a fixed order, so the host build (gcc -ffp-contract=off) and the device
 * build (hipcc -ffp-contract=off) produce identical bits — and the host build equals glibc 2.35's (float)cos((double)x) /
 * (float)sin((double)x) for EVERY float in [0, 6.5] (1 087 373 313 arguments, oracle/sincos_check/sincos_exhaustive.c,
 * profiles/r05_g_sincos_exhaustive.txt; tests/test_oracle_cv_orb.py runs every 97th).  The same library's float routines
 * cosf / sinf — what a build that resolved to the float overloads would call — differ from it in the last bit for 0.14 % of
 * the arguments.
 */
#ifndef MSLAM_SINCOS_H_
#define MSLAM_SINCOS_H_

#if defined(__HIPCC__)
#define MSLAM_HD __host__ __device__ static inline
#else
#include <math.h>
#define MSLAM_HD static inline
#endif

MSLAM_HD void mslam_sincos_f64(double x, double* s_out, double* c_out);

MSLAM_HD void mslam_sincos_f32(float xf, float* s_out, float* c_out)
{
    double s, c;
    mslam_sincos_f64((double)xf, &s, &c);
    *s_out = (float)s;
    *c_out = (float)c;
}

/* the double-precision evaluation itself (|x| <= 8; absolute error < 1e-15); also used by the PnP refinement */
MSLAM_HD void mslam_sincos_f64(double x, double* s_out, double* c_out)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632679489655800e+00; /* pi/2 rounded to double            */
    const double pio2_lo = 6.12323399573676603587e-17; /* pi/2 - pio2_hi                    */
    const double kd = floor(x * two_over_pi + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * pio2_hi) - kd * pio2_lo;
    const double z = r * r;
    /* sin r = r (1 - z/3! + z^2/5! - ... - z^7/15!),  cos r = 1 - z/2! + z^2/4! - ... + z^8/16! */
    double ps = -1.0 / 1307674368000.0;
    ps = ps * z + 1.0 / 6227020800.0;
    ps = ps * z - 1.0 / 39916800.0;
    ps = ps * z + 1.0 / 362880.0;
    ps = ps * z - 1.0 / 5040.0;
    ps = ps * z + 1.0 / 120.0;
    ps = ps * z - 1.0 / 6.0;
    ps = ps * z + 1.0;
    const double s = ps * r;
    double pc = 1.0 / 20922789888000.0;
    pc = pc * z - 1.0 / 87178291200.0;
    pc = pc * z + 1.0 / 479001600.0;
    pc = pc * z - 1.0 / 3628800.0;
    pc = pc * z + 1.0 / 40320.0;
    pc = pc * z - 1.0 / 720.0;
    pc = pc * z + 1.0 / 24.0;
    pc = pc * z - 0.5;
    const double c = pc * z + 1.0;
    double sr, cr;
    switch(k & 3)
    {
    case 0: sr = s, cr = c; break;
    case 1: sr = c, cr = -s; break;
    case 2: sr = -s, cr = -c; break;
    default: sr = -c, cr = s; break;
    }
    *s_out = sr;
    *c_out = cr;
}

#endif /* MSLAM_SINCOS_H_ */
