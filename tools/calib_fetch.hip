// Calibration of rocprofv3's FETCH_SIZE for THIS code's access pattern (MI355X_MICROARCH.md, HBM section: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  k_trace reads 64-byte BVH nodes,
// one node per lane, as four global_load_dwordx4 at scattered 64-B-aligned addresses.  Here every lane does exactly that on a
// table far larger than L2 + Infinity Cache, each 64-B record read exactly once (odd-multiplier bijection), so the byte count
// is known:  k_gather64 reads nRecords x 64 B;  k_stream reads the same table as a coalesced 16 B/lane stream (the pattern the
// guide calibrated: FETCH_SIZE shows half of it).
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/calib_fetch.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o calib --output-format csv -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void k_gather64(const float4 *__restrict__ table, unsigned nRecords /* power of two */, float *out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRecords) return;
    const unsigned rec = (i * 2654435761u) & (nRecords - 1u); // bijection on [0, nRecords)
    const float4 *p = table + (size_t)rec * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s; // keeps the loads alive, never true
}

__global__ __launch_bounds__(256) void k_gather64_half(const float4 *__restrict__ table, unsigned nRecords, float *out)
{
    // same, but only every second 64-B record (one per 128-B line): tells 64-B from 128-B fabric requests apart
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nRecords / 2) return;
    const unsigned rec = ((i * 2654435761u) & (nRecords / 2 - 1u)) * 2u;
    const float4 *p = table + (size_t)rec * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ table, size_t n4, float *out)
{
    float s = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = table[i];
        s += v.x + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    const unsigned nRecords = 1u << 25; // 32 Mi records x 64 B = 2 GiB  (>> 256 MiB Infinity Cache)
    const size_t bytes = (size_t)nRecords * 64;
    float4 *table = nullptr;
    float *out = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMemset(out, 0, 64));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        k_gather64<<<nRecords / 256, 256>>>(table, nRecords, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather64      reads %.1f MiB in %.3f ms  (%.2f TB/s)\n", bytes / 1048576.0, ms, bytes / ms / 1e9);
        CHECK(hipEventRecord(e0));
        k_gather64_half<<<nRecords / 512, 256>>>(table, nRecords, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gather64_half reads %.1f MiB in %.3f ms  (%.2f TB/s)\n", bytes / 2 / 1048576.0, ms, bytes / 2 / ms / 1e9);
        CHECK(hipEventRecord(e0));
        k_stream<<<256 * 16, 256>>>(table, bytes / 16, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_stream        reads %.1f MiB in %.3f ms  (%.2f TB/s)\n", bytes / 1048576.0, ms, bytes / ms / 1e9);
    }
    CHECK(hipFree(table));
    CHECK(hipFree(out));
    return 0;
}
