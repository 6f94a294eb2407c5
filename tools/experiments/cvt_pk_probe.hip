// cvt_pk_probe — how does v_cvt_pk_u8_f32 round?  (candidate for gray: min + cvt + pack in one instruction.)
//   hipcc -O3 --offload-arch=gfx950 cvt_pk_probe.hip -o /tmp/cvt_pk_probe && /tmp/cvt_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n)
{
    const int i = threadIdx.x;
    if(i < n)
    {
        unsigned r;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(r) : "v"(in[i]), "v"(0xAABBCCDDu));
        out[i] = r;
    }
}
int main()
{
    const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.0f, 1.5f, 2.5f, 2.51f, 3.49f, 3.5f, 3.99f, 254.5f, 254.99f, 255.0f, 255.5f, 256.f, 300.f, -0.5f, -3.f, 127.5f, 128.5f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; unsigned ho[64];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    for(int i = 0; i < n; ++i) printf("%8.3f -> byte1 = %3u   (word %08x)\n", h[i], (ho[i] >> 8) & 255u, ho[i]);
    return 0;
}
