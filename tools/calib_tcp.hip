// What does a DEPENDENT chain of scattered 64-byte record fetches cost on one MI355X CU, and does it depend on how the lanes of
// a wave split the record between them?  (The access pattern of k_trace: every node step of every ray is one such fetch.)
//   A  "lane":  every lane walks its own chain; a step = three global_load_dwordx4 to one 64-B record (48 B used)  [= nodeStep4]
//   A1 "lane1": the same with ONE dwordx4 per step (16 B of the record)                                          [fewer tag lookups per record?]
//   A2 "lane2": two dwordx4 per step (32-B node)
//   B  "quad":  four lanes share a chain; a step = one dwordx4 per lane, the quad reads the 64 consecutive bytes  [cooperative 4-wide]
//   C  "oct":   eight lanes share a chain, one dwordx4 per lane, 128 consecutive bytes                            [cooperative 8-wide]
// The next record index depends on the loaded data (the table holds zeros, the compiler cannot know), like a child reference.
// Table sizes: 16 KiB (L1-resident), 2 MiB (L2), 64 MiB (Infinity Cache), 2 GiB (HBM).  `waves` resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o calib_tcp tools/calib_tcp.hip && ./calib_tcp      (prints one JSON object)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

static const int kSteps = 2048;

__device__ __forceinline__ unsigned nextIndex(unsigned idx, unsigned data, unsigned mask) { return (idx * 2654435761u + 0x9E3779B9u + data) & mask; }

template <int LOADS>
__global__ __launch_bounds__(256) void k_lane(const uint4 *__restrict__ table, unsigned mask, unsigned *out)
{
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 747796405u & mask;
    unsigned acc = 0;
    for (int s = 0; s < kSteps; ++s) {
        const uint4 *p = table + (size_t)idx * 4;
        uint4 a = p[0];
        unsigned d = a.x ^ a.w;
        if (LOADS >= 2) {
            uint4 b = p[1];
            d ^= b.y;
        }
        if (LOADS >= 3) {
            uint4 c = p[2];
            d ^= c.z;
        }
        acc += d;
        idx = nextIndex(idx, d, mask);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// W: dwords per load (1, 2, 4); LOADS loads per step at consecutive addresses; ACTIVE: 64 or 32 lanes of each wave take part
template <int W, int LOADS, int ACTIVE>
__global__ __launch_bounds__(256) void k_width(const unsigned *__restrict__ table, unsigned mask, unsigned *out)
{
    if ((threadIdx.x & 63u) >= (unsigned)ACTIVE) return;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 747796405u & mask;
    unsigned acc = 0;
    for (int s = 0; s < kSteps; ++s) {
        const unsigned *p = table + (size_t)idx * 16;
        unsigned d = 0;
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            if (W == 4) {
                uint4 a = *reinterpret_cast<const uint4 *>(p + 4 * l);
                d ^= a.x ^ a.w;
            } else if (W == 2) {
                uint2 a = *reinterpret_cast<const uint2 *>(p + 2 * l);
                d ^= a.x ^ a.y;
            } else {
                d ^= p[l];
            }
        }
        acc += d;
        idx = nextIndex(idx, d, mask);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int GROUP> // lanes per chain: 4 (64-B record) or 8 (128-B record)
__global__ __launch_bounds__(256) void k_coop(const uint4 *__restrict__ table, unsigned mask, unsigned *out)
{
    const unsigned lane = threadIdx.x & 63u, sub = lane & (GROUP - 1);
    unsigned idx = ((blockIdx.x * 256u + threadIdx.x) / GROUP) * 747796405u & mask; // the same for the lanes of a group
    unsigned acc = 0;
    for (int s = 0; s < kSteps; ++s) {
        const uint4 *p = table + (size_t)idx * (GROUP == 8 ? 8 : 4) + sub;
        uint4 a = *p;
        unsigned d = a.x ^ a.w;
        // combine over the group (every lane needs the result to form the next index): xor-butterfly inside the quad / octet
        d ^= __shfl_xor((int)d, 1);
        d ^= __shfl_xor((int)d, 2);
        if (GROUP == 8) d ^= __shfl_xor((int)d, 4);
        acc += d;
        idx = nextIndex(idx, d, mask);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxBytes = (size_t)2 << 30;
    uint4 *table = nullptr;
    unsigned *out = nullptr;
    CHECK(hipMalloc(&table, maxBytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 0, maxBytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t sizes[4] = {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)2 << 30};
    const char *names[4] = {"16KiB_L1", "2MiB_L2", "64MiB_MALL", "2GiB_HBM"};
    const int wavesPerSimd[2] = {2, 6};
    printf("{\"cus\": %d, \"steps\": %d, \"runs\": [", cus, kSteps);
    bool first = true;
    for (int si = 0; si < 4; ++si) {
        for (int wi = 0; wi < 2; ++wi) {
            const int grid = cus * wavesPerSimd[wi];
            for (int variant = 0; variant < 5; ++variant) {
                const unsigned recBytes = variant == 4 ? 128u : 64u;
                const unsigned mask = (unsigned)(sizes[si] / recBytes) - 1u;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    switch (variant) {
                    case 0: hipLaunchKernelGGL(k_lane<3>, dim3(grid), dim3(256), 0, 0, table, mask, out); break;
                    case 1: hipLaunchKernelGGL(k_lane<1>, dim3(grid), dim3(256), 0, 0, table, mask, out); break;
                    case 2: hipLaunchKernelGGL(k_lane<2>, dim3(grid), dim3(256), 0, 0, table, mask, out); break;
                    case 3: hipLaunchKernelGGL(k_coop<4>, dim3(grid), dim3(256), 0, 0, table, mask, out); break;
                    default: hipLaunchKernelGGL(k_coop<8>, dim3(grid), dim3(256), 0, 0, table, mask, out); break;
                    }
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                const char *vn[5] = {"lane3", "lane1", "lane2", "quad", "oct"};
                const double chainsPerWave = variant == 3 ? 16.0 : (variant == 4 ? 8.0 : 64.0);
                const double chainSteps = (double)grid * 4.0 * chainsPerWave * kSteps; // record fetches
                printf("%s{\"table\": \"%s\", \"waves_per_simd\": %d, \"variant\": \"%s\", \"ms\": %.4f, \"record_fetches_per_s\": %.4e, "
                       "\"cycles_per_wave_step_at_2p4GHz\": %.1f}",
                       first ? "" : ", ", names[si], wavesPerSimd[wi], vn[variant], best, chainSteps / (best * 1e-3),
                       best * 1e-3 * 2.4e9 / kSteps);
                first = false;
            }
        }
    }
    // load width and lane activity, L2-resident table, 6 waves per SIMD
    {
        const unsigned mask = (unsigned)(((size_t)2 << 20) / 64) - 1u;
        const int grid = cus * 6;
        const char *vn[8] = {"x4x1", "x2x1", "x1x1", "x2x2", "x1x4", "x4x3_half_lanes", "x4x2_plus_x2", "x4x1_half_lanes"};
        for (int v = 0; v < 8; ++v) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                const unsigned *t = reinterpret_cast<const unsigned *>(table);
                switch (v) {
                case 0: hipLaunchKernelGGL((k_width<4, 1, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 1: hipLaunchKernelGGL((k_width<2, 1, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 2: hipLaunchKernelGGL((k_width<1, 1, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 3: hipLaunchKernelGGL((k_width<2, 2, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 4: hipLaunchKernelGGL((k_width<1, 4, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 5: hipLaunchKernelGGL((k_width<4, 3, 32>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                case 6: hipLaunchKernelGGL((k_width<2, 5, 64>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                default: hipLaunchKernelGGL((k_width<4, 1, 32>), dim3(grid), dim3(256), 0, 0, t, mask, out); break;
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double lanes = (v == 5 || v == 7) ? 32.0 : 64.0;
            const double chainSteps = (double)grid * 4.0 * lanes * kSteps;
            printf(", {\"table\": \"2MiB_L2\", \"waves_per_simd\": 6, \"variant\": \"%s\", \"ms\": %.4f, \"record_fetches_per_s\": %.4e, "
                   "\"cycles_per_wave_step_at_2p4GHz\": %.1f}", vn[v], best, chainSteps / (best * 1e-3), best * 1e-3 * 2.4e9 / kSteps);
        }
    }
    printf("]}\n");
    return 0;
}
