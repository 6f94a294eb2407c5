// valu_probe — issue cost (cycles per wave64 instruction per SIMD) of the vector instructions the extract
// kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD.  Settles whether an integer / packed / dot
// instruction occupies a SIMD for 2 or 4 cycles on gfx950 (MI355X_MICROARCH.md quotes 2 for v_fma_f32 with
// several waves resident, 4 for one wave alone).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x)                                                                             \
    do                                                                                       \
    {                                                                                        \
        hipError_t e_ = (x);                                                                 \
        if(e_ != hipSuccess)                                                                 \
        {                                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));        \
            return 1;                                                                        \
        }                                                                                    \
    } while(0)

constexpr int ITER = 2048; // loop trips; 16 instructions per trip

// 16 independent instructions per trip on 16 different destination registers.  BODY(d, a, b) is the asm text.
#define PROBE_KERNEL(NAME, ASM3)                                                                               \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, uint64_t* clk)                    \
    {                                                                                                          \
        uint32_t r[16];                                                                                        \
        _Pragma("unroll") for(int i = 0; i < 16; ++i) r[i] = seed * (i + 3) + threadIdx.x;                      \
        uint32_t a = seed ^ threadIdx.x, b = seed + 77u * threadIdx.x;                                          \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                       \
        for(int it = 0; it < ITER; ++it)                                                                        \
        {                                                                                                      \
            _Pragma("unroll") for(int i = 0; i < 16; ++i) asm volatile(ASM3 : "+v"(r[i]) : "v"(a), "v"(b));     \
        }                                                                                                      \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                       \
        uint32_t s = 0;                                                                                        \
        _Pragma("unroll") for(int i = 0; i < 16; ++i) s ^= r[i];                                               \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                               \
        if(threadIdx.x == 0)                                                                                   \
            clk[blockIdx.x] = t1 - t0;                                                                         \
    }

// 64-bit destination forms (packed f32): 8 independent register pairs
#define PROBE_KERNEL64(NAME, ASM3)                                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, uint64_t* clk)                    \
    {                                                                                                          \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                  \
        f2 r[16];                                                                                              \
        _Pragma("unroll") for(int i = 0; i < 16; ++i) r[i] = f2{(float)(seed * (i + 3)), (float)threadIdx.x};  \
        f2 a = {1.0f, 0.5f}, b = {0.25f, (float)threadIdx.x};                                                  \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                       \
        for(int it = 0; it < ITER; ++it)                                                                        \
        {                                                                                                      \
            _Pragma("unroll") for(int i = 0; i < 16; ++i) asm volatile(ASM3 : "+v"(r[i]) : "v"(a), "v"(b));     \
        }                                                                                                      \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                       \
        float s = 0;                                                                                           \
        _Pragma("unroll") for(int i = 0; i < 16; ++i) s += r[i].x + r[i].y;                                    \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);                                              \
        if(threadIdx.x == 0)                                                                                   \
            clk[blockIdx.x] = t1 - t0;                                                                         \
    }

PROBE_KERNEL(p_add_u32, "v_add_u32 %0, %1, %0")
PROBE_KERNEL(p_xor, "v_xor_b32 %0, %1, %0")
PROBE_KERNEL(p_and_or, "v_and_or_b32 %0, %0, %1, %2")
PROBE_KERNEL(p_lshl_or, "v_lshl_or_b32 %0, %0, 3, %2")
PROBE_KERNEL(p_lshrrev, "v_lshrrev_b32 %0, 3, %0")
PROBE_KERNEL(p_bfe, "v_bfe_u32 %0, %0, 3, 8")
PROBE_KERNEL(p_perm, "v_perm_b32 %0, %0, %1, %2")
PROBE_KERNEL(p_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
PROBE_KERNEL(p_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
PROBE_KERNEL(p_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
PROBE_KERNEL(p_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
PROBE_KERNEL(p_mul_u24, "v_mul_u32_u24 %0, %1, %0")
PROBE_KERNEL(p_mul_lo, "v_mul_lo_u32 %0, %1, %0")
PROBE_KERNEL(p_min3_u32, "v_min3_u32 %0, %0, %1, %2")
PROBE_KERNEL(p_max_u32, "v_max_u32 %0, %1, %0")
PROBE_KERNEL(p_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
PROBE_KERNEL(p_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
PROBE_KERNEL(p_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
PROBE_KERNEL(p_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
PROBE_KERNEL(p_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
PROBE_KERNEL(p_pk_min3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
PROBE_KERNEL(p_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
PROBE_KERNEL(p_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
PROBE_KERNEL(p_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
PROBE_KERNEL(p_fma_f32, "v_fma_f32 %0, %0, %1, %2")
PROBE_KERNEL(p_add_f32, "v_add_f32 %0, %1, %0")
PROBE_KERNEL(p_mul_f32, "v_mul_f32 %0, %1, %0")
PROBE_KERNEL(p_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
PROBE_KERNEL(p_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
PROBE_KERNEL(p_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
PROBE_KERNEL(p_sad_u8, "v_sad_u8 %0, %1, %2, %0")
PROBE_KERNEL(p_msad_u8, "v_msad_u8 %0, %1, %2, %0")
PROBE_KERNEL(p_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE_KERNEL(p_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE_KERNEL(p_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE_KERNEL(p_mov_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0")
PROBE_KERNEL(p_add_sdwa, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
PROBE_KERNEL(p_mad_u16, "v_mad_u16 %0, %0, %1, %2")
PROBE_KERNEL(p_mad_i32_i16, "v_mad_i32_i16 %0, %1, %2, %0")
PROBE_KERNEL(p_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0")
PROBE_KERNEL(p_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %2, vcc")
PROBE_KERNEL64(p_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
PROBE_KERNEL64(p_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
PROBE_KERNEL64(p_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")


PROBE_KERNEL(q_and, "v_and_b32 %0, %1, %0")
PROBE_KERNEL(q_or, "v_or_b32 %0, %1, %0")
PROBE_KERNEL(q_sub_u32, "v_sub_u32 %0, %1, %0")
PROBE_KERNEL(q_subrev_u32, "v_subrev_u32 %0, %1, %0")
PROBE_KERNEL(q_lshlrev, "v_lshlrev_b32 %0, 3, %0")
PROBE_KERNEL(q_ashrrev, "v_ashrrev_i32 %0, 3, %0")
PROBE_KERNEL(q_mov, "v_mov_b32 %0, %1")
PROBE_KERNEL(q_min_u32, "v_min_u32 %0, %1, %0")
PROBE_KERNEL(q_min_i32, "v_min_i32 %0, %1, %0")
PROBE_KERNEL(q_max_i32, "v_max_i32 %0, %1, %0")
PROBE_KERNEL(q_max_f32, "v_max_f32 %0, %1, %0")
PROBE_KERNEL(q_min_f32, "v_min_f32 %0, %1, %0")
PROBE_KERNEL(q_sub_f32, "v_sub_f32 %0, %1, %0")
PROBE_KERNEL(q_fmac_f32, "v_fmac_f32 %0, %1, %2")
PROBE_KERNEL(q_mac_legacy, "v_mul_legacy_f32 %0, %1, %0")
PROBE_KERNEL(q_add3, "v_add3_u32 %0, %0, %1, %2")
PROBE_KERNEL(q_lshl_add, "v_lshl_add_u32 %0, %0, 3, %2")
PROBE_KERNEL(q_add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
PROBE_KERNEL(q_or3, "v_or3_b32 %0, %0, %1, %2")
PROBE_KERNEL(q_xad, "v_xad_u32 %0, %0, %1, %2")
PROBE_KERNEL(q_bfi, "v_bfi_b32 %0, %1, %2, %0")
PROBE_KERNEL(q_alignbit, "v_alignbit_b32 %0, %0, %1, 8")
PROBE_KERNEL(q_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
PROBE_KERNEL(q_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
PROBE_KERNEL(q_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
PROBE_KERNEL(q_med3_f32, "v_med3_f32 %0, %0, %1, %2")
PROBE_KERNEL(q_max3_f32, "v_max3_f32 %0, %0, %1, %2")
PROBE_KERNEL(q_min3_i32, "v_min3_i32 %0, %0, %1, %2")
PROBE_KERNEL(q_add_u16, "v_add_u16 %0, %1, %0")
PROBE_KERNEL(q_sub_u16, "v_sub_u16 %0, %1, %0")
PROBE_KERNEL(q_max_u16, "v_max_u16 %0, %1, %0")
PROBE_KERNEL(q_min_u16, "v_min_u16 %0, %1, %0")
PROBE_KERNEL(q_max_i16, "v_max_i16 %0, %1, %0")
PROBE_KERNEL(q_mul_lo_u16, "v_mul_lo_u16 %0, %1, %0")
PROBE_KERNEL(q_lshlrev_b16, "v_lshlrev_b16 %0, 3, %0")
PROBE_KERNEL(q_add_f16, "v_add_f16 %0, %1, %0")
PROBE_KERNEL(q_max_f16, "v_max_f16 %0, %1, %0")
PROBE_KERNEL(q_fma_f16, "v_fma_f16 %0, %0, %1, %2")
PROBE_KERNEL(q_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
PROBE_KERNEL(q_pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 3, %0")
PROBE_KERNEL(q_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
PROBE_KERNEL(q_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
PROBE_KERNEL(q_add_co, "v_add_co_u32 %0, vcc, %1, %0")
PROBE_KERNEL(q_cmp_u32, "v_cmp_lt_u32 vcc, %1, %0")
PROBE_KERNEL(q_cmp_f32, "v_cmp_lt_f32 vcc, %1, %0")
PROBE_KERNEL(q_cmpx, "v_cmp_lt_u32 s[10:11], %1, %0")
PROBE_KERNEL(q_cndmask_ok, "v_cndmask_b32 %0, %0, %1, vcc")
PROBE_KERNEL(q_and_sdwa, "v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
PROBE_KERNEL(q_sub_u16_sdwa, "v_sub_u16_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0")
PROBE_KERNEL(q_readlane, "v_readlane_b32 s10, %0, 3")
PROBE_KERNEL(q_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
PROBE_KERNEL(q_sad_u16, "v_sad_u16 %0, %1, %2, %0")
PROBE_KERNEL(q_dot4_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
PROBE_KERNEL(q_dot8_u4, "v_dot8_u32_u4 %0, %1, %2, %0")
PROBE_KERNEL(q_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
PROBE_KERNEL(q_lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
PROBE_KERNEL(q_pk_add_i16_clamp, "v_pk_add_i16 %0, %0, %1 clamp")
PROBE_KERNEL(q_pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
PROBE_KERNEL(q_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
PROBE_KERNEL(q_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %0")
PROBE_KERNEL(q_cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %1, %0")

typedef void (*kern_t)(uint32_t*, uint32_t, uint64_t*);
struct Entry
{
    const char* name;
    kern_t fn;
    int insts_per_slot; // instructions in one asm slot
};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* d_out;
    uint64_t* d_clk;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(uint32_t)));
    CHECK(hipMalloc(&d_clk, (size_t)cus * 8 * sizeof(uint64_t)));
#define E(n) {#n, n, 1}
    std::vector<Entry> es = {E(q_and), E(q_or), E(q_sub_u32), E(q_subrev_u32), E(q_lshlrev), E(q_ashrrev), E(q_mov), E(q_min_u32), E(q_min_i32), E(q_max_i32), E(q_max_f32), E(q_min_f32), E(q_sub_f32), E(q_fmac_f32), E(q_mac_legacy), E(q_add3), E(q_lshl_add), E(q_add_lshl), E(q_or3), E(q_xad), E(q_bfi), E(q_alignbit), E(q_cvt_f32_u32), E(q_cvt_f32_i32), E(q_cvt_i32_f32), E(q_med3_f32), E(q_max3_f32), E(q_min3_i32), E(q_add_u16), E(q_sub_u16), E(q_max_u16), E(q_min_u16), E(q_max_i16), E(q_mul_lo_u16), E(q_lshlrev_b16), E(q_add_f16), E(q_max_f16), E(q_fma_f16), E(q_pk_sub_i16), E(q_pk_lshrrev_b16), E(q_pk_max_f16), E(q_pk_min_u16), E(q_add_co), E(q_cmp_u32), E(q_cmp_f32), E(q_cmpx), E(q_cndmask_ok), E(q_and_sdwa), E(q_sub_u16_sdwa), E(q_readlane), E(q_mbcnt), E(q_sad_u16), E(q_dot4_i8), E(q_dot8_u4), E(q_dot2_i16), E(q_lerp_u8), E(q_pk_add_i16_clamp), E(q_pk_sub_u16_clamp), E(q_mul_hi_u32), E(q_cvt_pkrtz), E(q_cvt_pk_u16_u32)};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-20s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD: in-kernel s_memtime | wall at 2.4 GHz)\n", "instr",
           "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for(const Entry& e : es)
    {
        printf("%-20s", e.name);
        for(int wps : {1, 2, 4, 8})
        {
            // 256-thread blocks = 4 waves = one per SIMD; wps blocks per CU
            const int blocks = cus * wps;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, d_clk); // warm
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, d_clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> clk(blocks);
            CHECK(hipMemcpy(clk.data(), d_clk, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
            double mean = 0;
            for(uint64_t c : clk)
                mean += (double)c;
            mean /= blocks;
            const double n_inst = (double)ITER * 16 * e.insts_per_slot;
            // s_memtime counts at a fixed 100 MHz on gfx9; report both it (x24 -> 2.4 GHz cycles) and wall
            const double cyc_mem = mean / n_inst / wps;
            const double cyc_wall = ms * 1e-3 * 2.4e9 / n_inst / wps;
            printf("  %4.2f|%4.2f", cyc_mem, cyc_wall);
        }
        printf("\n");
    }
    return 0;
}
