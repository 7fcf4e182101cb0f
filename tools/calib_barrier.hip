// What does a device-wide barrier inside a persistent kernel cost on one MI355X, against the boundary between two dependent kernels of
// one stream?  (DESIGN.md §2 "Why there is still no fused multi-stage tail kernel" prices a fused trace | sort | hit | advance loop with
// these two numbers.)  One 256-thread workgroup per CU; every round each workgroup writes a line, the barrier follows, then it reads the
// line its neighbour on ANOTHER XCD wrote (so the barrier needs agent-scope release / acquire, as a pipeline stage's queues would).
// Every spin is bounded: a barrier that does not complete sets a flag and all workgroups leave.
//   hipcc --offload-arch=gfx950 -O3 -o calib_barrier tools/calib_barrier.hip && ./calib_barrier     (prints one JSON object)
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

static const unsigned kSpinLimit = 1u << 20; // x s_sleep: far beyond any barrier, far below a watchdog

__device__ bool gridBarrier(unsigned *bar, unsigned target, unsigned *timedOut, bool fences)
{
    __syncthreads();
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // the other XCDs' L2s must see this workgroup's stores
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > kSpinLimit || __hip_atomic_load(timedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(timedOut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok != 0;
}

__global__ __launch_bounds__(256) void k_rounds(unsigned *bar, unsigned *timedOut, unsigned *lines, unsigned *sink, int rounds, int fences)
{
    const unsigned n = gridDim.x, b = blockIdx.x;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        lines[(size_t)b * 256 + threadIdx.x] = (unsigned)r * 2654435761u + b + threadIdx.x;
        if (!gridBarrier(bar, (unsigned)(r + 1) * n, timedOut, fences != 0)) return;
        acc += lines[(size_t)((b + 1) % n) * 256 + threadIdx.x]; // workgroup b + 1 runs on the next XCD
        if (!gridBarrier(bar + 32, (unsigned)(r + 1) * n, timedOut, fences != 0)) return; // (before the line is overwritten)
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_one(unsigned *lines, unsigned *sink, int r)
{
    const unsigned n = gridDim.x, b = blockIdx.x;
    const unsigned v = lines[(size_t)((b + 1) % n) * 256 + threadIdx.x];
    lines[(size_t)b * 256 + threadIdx.x + (size_t)n * 256 * ((r & 1) ? 1 : 0)] = v + (unsigned)r;
    if (v == 0x12345u) sink[0] = v;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *bar, *timedOut, *lines, *sink;
    CHECK(hipMalloc(&bar, 512));
    CHECK(hipMalloc(&timedOut, 64));
    CHECK(hipMalloc(&lines, (size_t)cus * 256 * 4 * 2));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(lines, 0, (size_t)cus * 256 * 4 * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int rounds = 2000;
    printf("{\"cus\": %d, \"rounds\": %d", cus, rounds);
    for (int fences = 1; fences >= 0; --fences) {
        float best = 1e30f;
        unsigned bad = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipMemset(bar, 0, 512));
            CHECK(hipMemset(timedOut, 0, 64));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_rounds, dim3(cus), dim3(256), 0, 0, bar, timedOut, lines, sink, rounds, fences);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned t = 0;
            CHECK(hipMemcpy(&t, timedOut, 4, hipMemcpyDeviceToHost));
            bad |= t;
            if (rep > 0 && ms < best) best = ms;
        }
        // two barriers per round
        printf(", \"%s\": {\"us_per_barrier\": %.3f, \"timed_out\": %u}", fences ? "barrier_release_acquire" : "barrier_atomics_only", best * 1000.0f / (2.0f * rounds), bad);
    }
    {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_one, dim3(cus), dim3(256), 0, 0, lines, sink, r);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf(", \"dependent_kernels\": {\"us_per_kernel\": %.3f}", best * 1000.0f / rounds);
    }
    printf("}\n");
    return 0;
}
