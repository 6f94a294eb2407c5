// Can the CPU write a frame straight into device memory (large BAR), and how fast?  Fine-grained device memory
// (hipExtMallocWithFlags / hipDeviceMallocFinegrained) is host-accessible when the whole VRAM sits behind the PCIe BAR.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/bar_write_probe.hip -o /tmp/bar_write_probe && /tmp/bar_write_probe
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

__global__ void k_sum(const uint32_t* p, size_t n, unsigned long long* out)
{
    unsigned long long s = 0;
    for(size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        s += p[i];
    atomicAdd(out, s);
}

int main()
{
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    printf("device %s, isLargeBar %d\n", prop.name, prop.isLargeBar);
    const size_t bytes = 640 * 480 * 3;
    std::vector<uint8_t> src(bytes);
    for(size_t i = 0; i < bytes; ++i) src[i] = (uint8_t)(i * 2654435761u >> 24);
    unsigned long long expect = 0;
    for(size_t i = 0; i < bytes / 4; ++i) { uint32_t v; memcpy(&v, &src[4 * i], 4); expect += v; }
    unsigned long long* d_out;
    hipMalloc(&d_out, 8);
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    struct Kind { const char* name; int which; } kinds[] = {{"fine-grained device memory", 0}, {"plain hipMalloc", 1}, {"hipMallocManaged", 2}};
    for(const Kind& k : kinds)
    {
        void* p = nullptr;
        hipError_t e = k.which == 0 ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : k.which == 1 ? hipMalloc(&p, bytes) : hipMallocManaged(&p, bytes);
        if(e != hipSuccess) { printf("%s: allocation failed (%s)\n", k.name, hipGetErrorString(e)); continue; }
        if(sigsetjmp(g_jmp, 1)) { printf("%s: NOT host-accessible (fault)\n", k.name); continue; }
        memcpy(p, src.data(), bytes);
        double best = 1e30;
        for(int rep = 0; rep < 50; ++rep)
        {
            const double t0 = now_us();
            memcpy(p, src.data(), bytes);
            __builtin_ia32_sfence();
            const double t1 = now_us();
            if(t1 - t0 < best) best = t1 - t0;
        }
        hipMemset(d_out, 0, 8);
        hipLaunchKernelGGL(k_sum, dim3(64), dim3(256), 0, 0, (const uint32_t*)p, bytes / 4, d_out);
        unsigned long long got = 0;
        hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost);
        // a word the CPU writes, seen by a kernel that is already running?  (coherence of the BAR path: write, then launch)
        printf("%s: host-accessible, CPU memcpy of %zu bytes best %.1f us (%.1f GB/s), kernel sees %s data\n", k.name, bytes, best,
               bytes / best * 1e-3, got == expect ? "the right" : "WRONG");
        // CPU read-back speed (uncached reads over the BAR are slow: just to know)
        const double t0 = now_us();
        volatile uint32_t sink = 0;
        for(size_t i = 0; i < 4096; i += 4) sink += ((volatile uint32_t*)p)[i / 4];
        printf("    CPU read of 4 KB: %.1f us\n", now_us() - t0);
    }
    return 0;
}
