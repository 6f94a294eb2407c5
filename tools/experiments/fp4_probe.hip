// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands holding +-1 and a 2^k block scale
// computes exact integer dot products scaled by 2^k.  Build: hipcc --offload-arch=gfx950 fp4_probe.hip -o fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// bits[row][2 dwords] = 64 bits per row; lane (r, h) expands dword h of row r to 32 fp4 values (+1 = 0x2, -1 = 0xA)
__device__ uint32_t expand8(uint32_t byte)
{
    uint32_t o = 0;
    for(int i = 0; i < 8; ++i)
        o |= (((byte >> i) & 1u) ? 0x2u : 0xAu) << (4 * i);
    return o;
}

__global__ void probe(const uint32_t* abits, const uint32_t* bbits, float* out, int sa, int sb)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const uint32_t da = abits[r * 2 + h], db = bbits[r * 2 + h];
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    for(int k = 0; k < 4; ++k)
    {
        a[k] = (int)expand8((da >> (8 * k)) & 255u);
        b[k] = (int)expand8((db >> (8 * k)) & 255u);
    }
    v16f c;
    for(int i = 0; i < 16; ++i)
        c[i] = 1000.0f + i;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, sa, 0, sb);
    for(int i = 0; i < 16; ++i)
    {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = c[i]; // out[A row][B col]
    }
}

int main()
{
    uint32_t ha[64], hb[64];
    srand(7);
    for(int i = 0; i < 64; ++i)
    {
        ha[i] = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
        hb[i] = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 11);
    }
    uint32_t *da, *db;
    float* dout;
    hipMalloc(&da, 256);
    hipMalloc(&db, 256);
    hipMalloc(&dout, 4096);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    for(int sb = 127; sb <= 127 + 14; sb += 14)
    {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout, 127, sb);
        float ho[1024];
        hipMemcpy(ho, dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for(int i = 0; i < 32; ++i)
            for(int j = 0; j < 32; ++j)
            {
                int ham = __builtin_popcount(ha[2 * i] ^ hb[2 * j]) + __builtin_popcount(ha[2 * i + 1] ^ hb[2 * j + 1]);
                int dot = 64 - 2 * ham;
                // which accumulator register held (i, j)?
                int reg = -1;
                for(int q = 0; q < 16; ++q)
                    for(int hh = 0; hh < 2; ++hh)
                        if((q & 3) + 8 * (q >> 2) + 4 * hh == i)
                            reg = q;
                float want = 1000.0f + reg + (float)dot * (float)(1 << (sb - 127));
                if(ho[i * 32 + j] != want)
                {
                    if(bad < 5)
                        printf("mismatch (%d,%d): got %f want %f\n", i, j, ho[i * 32 + j], want);
                    ++bad;
                }
            }
        printf("scale_b=%d: %d mismatches\n", sb, bad);
    }
    return 0;
}
