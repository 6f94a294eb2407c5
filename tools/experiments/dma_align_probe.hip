// dma_align_probe — does global_load_lds_dwordx4 (LDS-DMA) accept a global source address that is not 16-byte (not even
// 4-byte) aligned?  (the k_fast_cells question: a tile staged from column x0 - 1 instead of x0 - 3 would put the tested
// pixels on dword boundaries of the tile and save three v_alignbyte per step.)  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 dma_align_probe.hip -o /tmp/dma_align_probe && /tmp/dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(64) void probe(const uint8_t* __restrict__ src, int shift, uint8_t* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * 16];
    const int lane = threadIdx.x;
    for(int i = lane; i < 64 * 16 / 4; i += 64)
        reinterpret_cast<uint32_t*>(tile)[i] = 0xEEEEEEEEu;
    __syncthreads();
    if constexpr(BYTES == 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 16 * lane),
                                         (__attribute__((address_space(3))) void*)&tile[0], 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 4 * lane),
                                         (__attribute__((address_space(3))) void*)&tile[0], 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    for(int i = lane; i < 64 * BYTES; i += 64)
        out[i] = tile[i];
}

template <int BYTES>
static int run(const uint8_t* d_src, uint8_t* d_out, const std::vector<uint8_t>& h_src)
{
    int bad_shifts = 0;
    for(int shift = 0; shift < 20; ++shift)
    {
        hipMemset(d_out, 0, 64 * 16);
        hipLaunchKernelGGL(probe<BYTES>, dim3(1), dim3(64), 0, 0, d_src, shift, d_out);
        if(hipDeviceSynchronize() != hipSuccess)
        {
            printf("bytes %2d shift %2d: launch failed\n", BYTES, shift);
            return -1;
        }
        std::vector<uint8_t> h(64 * BYTES);
        hipMemcpy(h.data(), d_out, h.size(), hipMemcpyDeviceToHost);
        int bad = 0;
        for(int i = 0; i < 64 * BYTES; ++i)
            bad += h[i] != h_src[shift + i];
        printf("bytes %2d shift %2d: %s (%d of %d bytes differ; first bytes %02x %02x %02x %02x, expected %02x %02x %02x %02x)\n", BYTES,
               shift, bad ? "MISMATCH" : "ok", bad, 64 * BYTES, h[0], h[1], h[2], h[3], h_src[shift], h_src[shift + 1], h_src[shift + 2],
               h_src[shift + 3]);
        bad_shifts += bad != 0;
    }
    return bad_shifts;
}

int main()
{
    std::vector<uint8_t> h_src(4096);
    for(size_t i = 0; i < h_src.size(); ++i)
        h_src[i] = (uint8_t)((i * 37 + (i >> 8) * 11 + 5) & 0xFF);
    uint8_t *d_src, *d_out;
    hipMalloc(&d_src, h_src.size());
    hipMalloc(&d_out, 64 * 16);
    hipMemcpy(d_src, h_src.data(), h_src.size(), hipMemcpyHostToDevice);
    const int b16 = run<16>(d_src, d_out, h_src);
    const int b4 = run<4>(d_src, d_out, h_src);
    printf("SUMMARY dwordx4: %d shifts wrong; dword: %d shifts wrong\n", b16, b4);
    return 0;
}
