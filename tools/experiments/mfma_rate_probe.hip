// mfma_rate_probe.hip — how many cycles does a SIMD's matrix core take per v_mfma_f32_32x32x64_f8f6f4 with FP4 operands?
// (round 6: the matcher's MFMA-only loop measured 43-46 cycles per MFMA against the 32 the 10 PFLOP/s paper peak implies)
// W waves per SIMD each issue ITER x 16 MFMAs on four independent accumulators (dependent distance 4), nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <bool SCALED>
__global__ __launch_bounds__(256) void k_rate(const v8i* in, float* out, long long* cyc, int iters)
{
    const v8i a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    v16f c0, c1, c2, c3;
    for(int i = 0; i < 16; ++i)
        c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for(int it = 0; it < iters; ++it)
    {
#pragma unroll
        for(int s = 0; s < 4; ++s)
        {
            if(SCALED)
            {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
            }
            else
            {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for(int i = 0; i < 16; ++i)
        s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if(threadIdx.x == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
}

int main()
{
    v8i* in;
    float* out;
    long long* cyc;
    hipMalloc(&in, 128 * sizeof(v8i));
    hipMemset(in, 0x22, 128 * sizeof(v8i));
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    for(int scaled = 0; scaled < 2; ++scaled)
        for(int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2)
        {
            const int grid = 256 * wgs_per_cu; // one 4-wave workgroup = one wave per SIMD
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for(int rep = 0; rep < 3; ++rep)
            {
                hipEventRecord(e0);
                if(scaled)
                    hipLaunchKernelGGL(k_rate<true>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                else
                    hipLaunchKernelGGL(k_rate<false>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double mfma_per_simd = (double)iters * 16 * wgs_per_cu;
            printf("%s, %d wave(s) per SIMD: %.3f ms, %.1f ns per MFMA per SIMD = %.1f cycles at 2.1 GHz (%.1f at 2.4); wave 0 saw %.1f counter ticks per own MFMA; %.2f PFLOP/s\n",
                   scaled ? "scaled  " : "unscaled", wgs_per_cu, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.1,
                   ms * 1e6 / mfma_per_simd * 2.4, (double)c / (iters * 16), mfma_per_simd * 1024 * 131072.0 / (ms * 1e-3) / 1e15);
        }
    return 0;
}
