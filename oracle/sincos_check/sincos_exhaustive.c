/* sincos_exhaustive.c — the cv::ORB mode's cos / sin against the REAL C library of this image, over the whole input domain.
 *
 * OpenCV's computeOrbDescriptors (features2d/src/orb.cpp, reached from the reference's orb_feature.cpp:25,40) does
 *     angle *= (float)(CV_PI/180.f);  float a = (float)cos(angle), b = (float)sin(angle);
 * inside namespace cv.  With GCC's libstdc++ the float overloads of cos / sin live in namespace std only (<cmath>), cv's own
 * using-list (cvstd.hpp) takes sqrt / exp / pow / log from std but not cos / sin, so the unqualified call resolves to the C
 * library's double function — which is also what the explicit (float) casts say: a = (float)cos((double)angle).
 * include/mslam_sincos.h (what the GPU evaluates, bit-identical on host and device: single rounded double operations in a
 * fixed order) is compared here with exactly that expression for EVERY float in [0, 6.5] (angles are degrees in [0, 360)
 * times (float)(pi/180): at most 6.2832), and the C library's float routines cosf / sinf are counted beside it (what a build
 * that resolved to the float overloads would call).
 * Test infrastructure: tests/test_oracle_cv_orb.py builds and runs it (strided by default, every float with
 * MSLAM_EXHAUSTIVE=1).  usage: sincos_exhaustive [stride]; prints one line, exit code 1 on a mismatch. */
#include "../../include/mslam_sincos.h"
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char** argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    const uint32_t lo = bits(0.0f), hi = bits(6.5f);
    uint64_t n = 0, bad = 0, libm_f = 0;
    for(uint64_t u = lo; u <= hi; u += stride ? stride : 1)
    {
        const uint32_t v = (uint32_t)u;
        float x, s, c;
        memcpy(&x, &v, 4);
        mslam_sincos_f32(x, &s, &c);
        const float rs = (float)sin((double)x), rc = (float)cos((double)x);
        if(bits(s) != bits(rs) || bits(c) != bits(rc))
        {
            if(bad < 5)
                printf("mismatch at %a: sin %a vs %a, cos %a vs %a\n", x, s, rs, c, rc);
            ++bad;
        }
        if(bits(sinf(x)) != bits(rs) || bits(cosf(x)) != bits(rc))
            ++libm_f;
        ++n;
    }
    printf("floats %llu mismatches %llu libm_float_routines_differ %llu\n", (unsigned long long)n, (unsigned long long)bad,
           (unsigned long long)libm_f);
    return bad ? 1 : 0;
}
