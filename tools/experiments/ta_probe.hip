// ta_probe — how fast can a wave fetch a 39-row x 44-byte patch from an L2-resident u8 image, depending on how the
// lanes are laid over it?  (the k_describe question).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 ta_probe.hip -o /tmp/ta_probe && /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int W = 640, H = 480, PITCH = 640, NF = 64;
constexpr int PATCHES_PER_WAVE = 64;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int V>
__global__ __launch_bounds__(256) void probe(const uint8_t* __restrict__ img, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave_id = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t seed = 12345u + wave_id * 7919u;
    uint32_t acc = 0;
    const uint8_t* frame = img + (size_t)(wave_id % NF) * PITCH * H;
    for(int p = 0; p < PATCHES_PER_WAVE; ++p)
    {
        const int x0 = (int)(rnd(seed) % (W - 64)), y0 = (int)(rnd(seed) % (H - 40));
        const uint8_t* base = frame + y0 * PITCH + x0;
        if(V == 1) // dword per lane, 11 lanes per row, base 4-aligned
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~3);
#pragma unroll
            for(int q = 0; q < 7; ++q)
            {
                const int t = lane + 64 * q;
                if(t < 429)
                {
                    const int r = t / 11, c = t - r * 11;
                    acc += *reinterpret_cast<const uint32_t*>(b + r * PITCH + 4 * c);
                }
            }
        }
        if(V == 2) // dword per lane, 16 lanes per row (64 B from a 16-aligned start)
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~15);
#pragma unroll
            for(int q = 0; q < 10; ++q)
            {
                const int t = lane + 64 * q;
                if(t < 39 * 16)
                    acc += *reinterpret_cast<const uint32_t*>(b + (t >> 4) * PITCH + 4 * (t & 15));
            }
        }
        if(V == 3) // dwordx2 per lane, 6 lanes per row (48 B), 8-aligned
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~7);
#pragma unroll
            for(int q = 0; q < 4; ++q)
            {
                const int t = lane + 64 * q;
                if(t < 39 * 6)
                {
                    const int r = t / 6, c = t - r * 6;
                    const uint2 v = *reinterpret_cast<const uint2*>(b + r * PITCH + 8 * c);
                    acc += v.x + v.y;
                }
            }
        }
        if(V == 4) // dwordx4 per lane, 4 lanes per row (64 B), 16-aligned
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~15);
#pragma unroll
            for(int q = 0; q < 3; ++q)
            {
                const int t = lane + 64 * q;
                if(t < 39 * 4)
                {
                    const uint4 v = *reinterpret_cast<const uint4*>(b + (t >> 2) * PITCH + 16 * (t & 3));
                    acc += v.x + v.y + v.z + v.w;
                }
            }
        }
        if(V == 5) // lane = row: 11 dword loads down the row (each instruction: 39 rows x 4 B)
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~3);
            if(lane < 39)
            {
#pragma unroll
                for(int c = 0; c < 11; ++c)
                    acc += *reinterpret_cast<const uint32_t*>(b + lane * PITCH + 4 * c);
            }
        }
        if(V == 6) // lane = row: 3 x dwordx4 per lane (16-aligned)
        {
            const uint8_t* b = frame + y0 * PITCH + (x0 & ~15);
            if(lane < 39)
            {
#pragma unroll
                for(int c = 0; c < 4; ++c)
                {
                    const uint4 v = *reinterpret_cast<const uint4*>(b + lane * PITCH + 16 * c);
                    acc += v.x + v.y + v.z + v.w;
                }
            }
        }
        if(V == 7) // byte gathers straight from memory: 8 random bytes per lane inside the patch (no staging at all)
        {
#pragma unroll
            for(int q = 0; q < 8; ++q)
            {
                const uint32_t rr = rnd(seed) + lane * 2654435761u;
                acc += base[((rr >> 4) % 39) * PITCH + ((rr >> 12) % 39)];
            }
        }
        if(V == 8) // unaligned dword per lane, 8 lanes per row x 31 rows (the phase-A pattern)
        {
#pragma unroll
            for(int q = 0; q < 4; ++q)
            {
                const int t = lane + 64 * q;
                if(t < 248)
                {
                    uint32_t v;
                    __builtin_memcpy(&v, base + (t >> 3) * PITCH + 4 * (t & 7), 4);
                    acc += v;
                }
            }
        }
        (void)base;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int V>
static void run(const uint8_t* img, uint32_t* out, const char* what)
{
    const int blocks = 256 * 16;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, img, out);
    hipEventRecord(a, 0);
    for(int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, img, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double patches = 5.0 * blocks * 4 * PATCHES_PER_WAVE;
    std::printf("V%d %-70s %7.3f ms  %6.1f ns per 1000 patches  (%.2f G patches/s)\n", V, what, ms / 5, ms * 1e6 / patches * 1000 / 1e0 / 1e3,
                patches / (ms * 1e-3) / 1e9);
}

int main()
{
    uint8_t* img;
    uint32_t* out;
    hipMalloc(&img, (size_t)NF * PITCH * H + 4096);
    hipMalloc(&out, 256 * 16 * 256 * 4);
    hipMemset(img, 7, (size_t)NF * PITCH * H + 4096);
    run<1>(img, out, "dword/lane, 11 lanes per row (k_describe phase C today)");
    run<2>(img, out, "dword/lane, 16 lanes per row (64 B, 16-aligned)");
    run<3>(img, out, "dwordx2/lane, 6 lanes per row");
    run<4>(img, out, "dwordx4/lane, 4 lanes per row, 16-aligned");
    run<5>(img, out, "lane = row, 11 dword loads");
    run<6>(img, out, "lane = row, 4 dwordx4 loads");
    run<7>(img, out, "8 random byte gathers per lane, no staging");
    run<8>(img, out, "unaligned dword/lane, 8 lanes x 31 rows (phase A today)");
    return 0;
}
