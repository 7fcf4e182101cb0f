// Issue cost of the VALU instructions k_trace's node step is made of, per wave64 instruction, at 8 waves per SIMD (tools/calib_valu.hip
// does the same for v_fma_f32 alone).  Each kernel runs a long chain over 8 independent registers of ONE instruction kind.
//   hipcc --offload-arch=gfx950 -O3 -o calib_ops tools/calib_ops.hip && ./calib_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

static const int kIters = 1 << 13;

#define KERNEL(NAME, BODY)                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned a, unsigned b)  \
    {                                                                                   \
        unsigned r[8]; unsigned long long m = a | ((unsigned long long)b << 32); (void)m;                                                                  \
        for (int k = 0; k < 8; ++k) r[k] = threadIdx.x * 2654435761u + k;               \
        for (int i = 0; i < kIters; ++i) {                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                               \
            {                                                                           \
                _Pragma("unroll") for (int k = 0; k < 8; ++k) { BODY; }                 \
            }                                                                           \
        }                                                                               \
        unsigned s = 0;                                                                 \
        for (int k = 0; k < 8; ++k) s ^= r[k];                                          \
        if (s == 0x12345678u) out[0] = s;                                               \
    }

KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_cvt_ubyte, asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[k])))
KERNEL(k_cvt_u32, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[k])))
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 7, 7" : "+v"(r[k])))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(a)))
KERNEL(k_min_u32, asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_cmp, asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(r[k]), "v"(a) : "vcc"))
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(r[k]) : "v"(a)))

KERNEL(k_cndmask_s, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "s"(m)))
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[k]) : "v"(a) : "vcc"))
KERNEL(k_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_min3_f32, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_sub_f32, asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_mad_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b)))
KERNEL(k_cmp_u32_s, asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(r[k]), "v"(a)))
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k])))
KERNEL(k_ldexp, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a)))

// round 4 (packet traversal): what moving a wave-uniform value costs
KERNEL(k_readlane, { unsigned t; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(r[k])); asm volatile("" :: "s"(t)); })
KERNEL(k_readlane_use, { unsigned t; asm volatile("v_readlane_b32 %0, %1, 5\n s_nop 3\n v_fma_f32 %1, %0, %2, %1" : "=&s"(t), "+v"(r[k]) : "v"(a)); })
KERNEL(k_fma_sgpr, asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r[k]) : "s"(a), "v"(b)))
KERNEL(k_cvt_ubyte_sgpr, asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r[k]) : "s"(a)))
__global__ __launch_bounds__(256) void k_lds_bcast(unsigned *out, unsigned a, unsigned b)
{
    __shared__ float4 tab[64];
    tab[threadIdx.x & 63] = make_float4((float)threadIdx.x, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned idx = a & 7u; // wave-uniform address
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(idx * 16u), "n"((j & 31) * 16));
            acc.x += v.x;
        }
    }
    if (acc.x == 12345.0f) out[0] = 1;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *out = nullptr;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    struct { const char *name; void (*fn)(unsigned *, unsigned, unsigned); } ks[] = {
        {"v_fma_f32", k_fma}, {"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_cvt_f32_u32", k_cvt_u32}, {"v_bfe_u32", k_bfe}, {"v_cndmask_b32", k_cndmask},
        {"v_min_u32", k_min_u32}, {"v_max3_f32", k_max3}, {"v_and_or_b32", k_and_or}, {"v_add_u32", k_add_u32}, {"v_mul_f32", k_mul_f32},
        {"v_cmp_le_f32", k_cmp}, {"v_lshl_or_b32", k_lshl_or}, {"v_cndmask_b32_e64 (sgpr mask)", k_cndmask_s}, {"v_cmp_lt_u32 + v_cndmask_b32 (pair)", k_cmp_cnd},
        {"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32}, {"v_min3_f32", k_min3_f32}, {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_lshl_add_u32", k_lshl_add},
        {"v_sub_f32", k_sub_f32}, {"v_perm_b32", k_perm}, {"v_mad_u32_u24", k_mad_u24}, {"v_cmp_lt_u32_e64 (sgpr dst)", k_cmp_u32_s}, {"v_rcp_f32", k_rcp},
        {"v_ldexp_f32", k_ldexp}, {"v_readlane_b32 (sgpr dst)", k_readlane}, {"v_readlane_b32 + s_nop 3 + v_fma_f32 using it", k_readlane_use}, {"v_fma_f32 (one sgpr operand)", k_fma_sgpr}, {"v_cvt_f32_ubyte1 (sgpr source)", k_cvt_ubyte_sgpr}, {"ds_read_b128 uniform address + wait (64 per iteration = 1 op here)", k_lds_bcast}};
    // reference clock from the FMA kernel: 2 cycles per wave64 FMA with many waves resident (profiles/r2_calib_valu.json)
    printf("{\"cus\": %d, \"ops\": [", cus);
    double fmaNs = 0;
    for (size_t k = 0; k < sizeof(ks) / sizeof(ks[0]); ++k) {
        const int grid = cus * 8;
        hipLaunchKernelGGL(ks[k].fn, dim3(grid), dim3(256), 0, 0, out, 0x3f800001u, 0x33000000u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[k].fn, dim3(grid), dim3(256), 0, 0, out, 0x3f800001u, 0x33000000u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double instrPerSimd = 8.0 * (double)kIters * 64.0; // 8 waves per SIMD x instructions per wave
        const double nsPerInstr = best * 1e6 / instrPerSimd;
        if (k == 0) fmaNs = nsPerInstr;
        printf("%s{\"op\": \"%s\", \"ms\": %.4f, \"ns_per_wave_instr_per_simd\": %.4f, \"relative_to_fma\": %.2f}", k ? ", " : "", ks[k].name, best, nsPerInstr,
               nsPerInstr / fmaNs);
    }
    printf("]}\n");
    return 0;
}
