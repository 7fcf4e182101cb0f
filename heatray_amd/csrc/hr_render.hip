// hr_render.hip — the per-pass wavefront pipeline: primary-ray generation, persistent-threads BVH
// traversal (closest hit + occlusion in ONE kernel), SoA shading with wave-level queue compaction,
// accumulation.
//
// Replaces rlRenderFrame() (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386) and the RLSL
// programs it runs (Resources/shaders/perspective.rlsl, physicallyBased.rlsl, glass.rlsl, *Light.rlsl,
// accumulator.rlsl).
//
// A pass needs depth+2 dependent stages (trace -> shade -> trace -> ...), and late stages hold few, long
// rays, so running one pass at a time leaves the chip idle in every stage's tail.  The host therefore
// keeps up to `depth+2` passes in flight, each at a different stage, and every "macro step" launches
//
//   raygen (the pass injected this step)  ->  k_trace (all in-flight passes: closest-hit rays of the
//   current stage + occlusion rays emitted by the previous stage)  ->  k_shade (all in-flight passes)
//   ->  k_resolve (passes that finished)
//
// so each launch carries about one whole pass worth of rays of every depth.  Each in-flight pass sums its
// sample into its own pass buffer (plain read-modify-write: a pixel has at most one live path and one
// live occlusion ray per pass, and the stages are stream-ordered), and k_resolve adds finished samples to
// the accumulation buffer in pass order — reproducible bit for bit.
#include "hr_kernels.h"
#include "hr_display.h"
#include "hr_shade.h"
#include "hr_trace.h"

namespace hr {

#ifndef HR_NODE_STEPS
#define HR_NODE_STEPS 6 // inner-node steps per round of the trace loop
#endif
static const int kBlock = 256;
static const int kWavesPerBlock = kBlock / 64;
#ifndef HR_TRACE_BLOCK
#define HR_TRACE_BLOCK 256 // threads per workgroup of k_trace (an exited workgroup frees its CU slot only as a whole)
#endif
static const int kTraceBlock = HR_TRACE_BLOCK;
static const int kTraceWaves = kTraceBlock / 64;

HRD uint32_t laneId() { return threadIdx.x & 63u; }

// Wave-level compaction: lanes with `want` get consecutive slots from *counter (one atomic per wave).
HRD uint32_t waveReserve(bool want, uint32_t *counter)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0;
    const uint32_t lane = laneId();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Block-level compaction: one global atomic per workgroup.  Every thread of the block must call it.
// `scratch` is 2 + (blockDim/64) words of LDS.
HRD uint32_t blockReserve(bool want, uint32_t *counter, uint32_t *scratch)
{
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const unsigned long long mask = __ballot(want);
    if (lane == 0) scratch[2 + wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < nWaves; ++w) {
            const uint32_t c = scratch[2 + w];
            scratch[2 + w] = tot; // exclusive prefix
            tot += c;
        }
        scratch[0] = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    const uint32_t slot = scratch[0] + scratch[2 + wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads(); // scratch may be reused by the next call
    return slot;
}

HRD uint32_t waveSum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- queue guards.  Queue capacities are upper bounds the host derives (hr_core.hip::macroStep); should one ever be wrong, the append
// that does not fit is DROPPED and the reader sees the counter clamped, and the first such event is reported to pinned host memory
// (kind of queue, step, table entry, count): the next hr_flush / hr_readback / hr_synchronize fails with HR_ERR_DEVICE naming it — instead
// of a write past the end of an arena (round 4 met one as a memory fault while the packet kernel's partial count served as a bound).
enum OverflowKind : uint32_t { OVF_CAMERA = 1, OVF_CLOSEST_IN = 2, OVF_OCCLUSION_IN = 3, OVF_CLOSEST_OUT = 4, OVF_OCCLUSION_OUT = 5, OVF_HIT_LIST = 6 };
__device__ __attribute__((noinline)) void queueOverflow(const StepTable *tbl, uint32_t kind, uint32_t seg, uint32_t count)
{
    uint32_t *h = tbl->hostOverflow;
    if (!h || __hip_atomic_load(&h[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return; // (the FIRST report stays: what follows from it — a clamped reader downstream — would only hide it)
    __hip_atomic_store(&h[1], (uint32_t)tbl->seqValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[2], seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[3], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h[0], kind, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
HRD uint32_t closestCap(const SegDev &sg) { return sg.qinCap < sg.hitCap ? sg.qinCap : sg.hitCap; } // (hits and the hit list are as long as the step's bound)
HRD uint32_t closestCount(const SegDev &sg) // rays of the closest-hit queue that are really there
{
    const uint32_t n = *sg.qCountIn, cap = closestCap(sg);
    return n < cap ? n : cap;
}
HRD uint32_t occlusionCount(const SegDev &sg)
{
    const uint32_t n = *sg.sCountIn;
    return n < sg.sInCap ? n : sg.sInCap;
}

HRD uint32_t packMeta(const Ray &r)
{
    return (uint32_t)(r.sequenceID & 0xFF) | ((uint32_t)(r.depth & 0xFFFF) << 8) | ((uint32_t)r.missKind << 24) | ((uint32_t)r.missIdx << 27);
}

HRD void storeRay(const RayQueue &q, uint32_t slot, const Ray &r, uint32_t pixel, uint32_t srcPrim)
{
    G(q.A)[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.maxT);
    G(q.B)[slot] = make_float4(r.d.x, r.d.y, r.d.z, r.extraT);
    G(q.C)[slot] = make_float4(r.weight.x, r.weight.y, r.weight.z, __uint_as_float(pixel));
    G(q.D)[slot] = make_int4((int)packMeta(r), r.sequenceIndexOffset, (int)srcPrim, (int)packCone(r.coneW, r.coneG));
}

// pixel of thread `gid` in this context's tile shard: tiles in round-robin order, 8x8-pixel blocks inside
// a tile so that one wave covers a compact screen patch
HRD bool ownedPixel(const FrameDev &fr, uint32_t gid, int &x, int &y)
{
    const uint32_t perTile = (uint32_t)(fr.tile * fr.tile);
    const uint32_t tileSlot = gid / perTile, within = gid % perTile;
    if (tileSlot >= (uint32_t)fr.nOwnedTiles) return false;
    const int tileId = fr.rank + (int)tileSlot * fr.world;
    const int tx = tileId % fr.tilesX, ty = tileId / fr.tilesX;
    const int blk = (int)(within >> 6), l = (int)(within & 63u), bpr = fr.tile >> 3;
    x = tx * fr.tile + (blk % bpr) * 8 + (l & 7);
    y = ty * fr.tile + (blk / bpr) * 8 + (l >> 3);
    return x < fr.W && y < fr.H;
}

// ------------------------------------------------------------------------------------------ raygen
// One pixel of one pass (perspective.rlsl:39-93), shared by k_raygen and k_raygen_packets: the camera ray, the pass sample's zero
// (perspective.rlsl:60 accumulate(vec4(0,0,0,1)) for sampled pixels) and the root cull.
// A camera ray that misses the box of the whole tree is finished here: its traversal would be ONE node step at the root that
// pushes nothing, then a hit record, then the miss shader in k_shade_sort — a queue slot (64 B), a record and three kernels' worth
// of loads for a ray whose fate is already known (with the benchmark's camera — SURVEY 8d: distance 3 x the scene's radius, 50 mm lens
// — that is three camera rays in four).  Its defaultPrimitive's shader runs right here (the sample was just set to zero: same single
// addition as later), and it still counts as a closest-hit ray: it WAS traced, by the test below.  A decision no traversal can
// contradict: rootMissed() is the slab test on the frame box of the root's 64-byte node (planes q = 0 and 255), which contains the padded
// box of every triangle of the scene; a ray that misses it can hit nothing (a hit point lies inside its triangle's half-padded box, hr_trace.h),
// whichever copy of the tree — the packet kernel's 64-byte nodes, k_trace's 32-byte ones — it would have walked.
struct CameraLane {
    Ray r;
    uint32_t pixel;
    bool active;  // the pixel is sampled this pass
    bool enqueue; // ... and its ray has to be traced
    uint32_t nAcc;
};
HRD void cameraLane(const SceneDev &S, const hr_pass_params &pp, const SegDev &seg, const FrameDev &fr, bool inFrame, int x, int y, CameraLane &c)
{
    c.pixel = (uint32_t)(y * fr.W + x);
    c.r.valid = false;
    c.active = inFrame;
    if (c.active) c.active = generatePrimary(S, pp, fr.W, fr.H, x, y, c.r);
    if (inFrame) G(reinterpret_cast<float4 *>(seg.passbuf))[c.pixel] = make_float4(0.0f, 0.0f, 0.0f, c.active ? 1.0f : 0.0f);
    if (inFrame && seg.passbufB) { // HR_ESTIMATOR_ALL_LIGHTS: three more partial sums behind the first
        const size_t framePixels = (size_t)(seg.passbufB - seg.passbuf) >> 2;
        for (int j = 1; j <= 3; ++j) G(reinterpret_cast<float4 *>(seg.passbuf))[c.pixel + j * framePixels] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    c.enqueue = c.active;
    c.nAcc = 0;
    if (c.active && S.nTris > 0 && S.rootLeafCount == 0 && rootMissed(S.nodes, c.r.o, c.r.d, S.rayEps, c.r.maxT)) {
        c.enqueue = false;
        if (c.r.missKind == MISS_ENV) {
            ShaderT<0> sh(S, pp, G(seg.passbuf) + (size_t)c.pixel * 4);
            sh.performAccumulate(sh.environmentRadiance(c.r.d, c.r.weight));
            c.nAcc = sh.nAccum;
        }
    }
}

static const int kRaygenBlock = 1024; // one queue-slot reservation (global atomic) per 1024 pixels
__global__ __launch_bounds__(kRaygenBlock) void k_raygen(const SceneDev *__restrict__ Sp, const StepTable *__restrict__ tbl, SegList segs, FrameDev fr,
                                                         Stats *stats)
{
    __shared__ uint32_t scratch[2 + kRaygenBlock / 64];
    const SceneDev &S = *Sp;
    stats += blockIdx.x & (kStatSlots - 1);
    const SegDev &seg = tbl->seg[segs.seg[blockIdx.y]]; // blockIdx.y: which of the passes injected this step
    int x = 0, y = 0;
    const bool inFrame = ownedPixel(fr, blockIdx.x * kRaygenBlock + threadIdx.x, x, y);
    CameraLane c;
    cameraLane(S, seg.pp, seg, fr, inFrame, x, y, c);
    const Ray &r = c.r;
    const uint32_t pixel = c.pixel;
    const bool active = c.active, enqueue = c.enqueue;
    uint32_t nAcc = c.nAcc;
    const uint32_t slot = blockReserve(enqueue, seg.qCountIn, scratch);
    if (enqueue) {
        if (slot < closestCap(seg))
            storeRay(seg.qin, slot, r, pixel, 0xFFFFFFFFu);
        else
            queueOverflow(tbl, OVF_CAMERA, (uint32_t)segs.seg[blockIdx.y], slot + 1u);
    }
    const uint32_t n = waveSum(active ? 1u : 0u), nCulled = waveSum((active && !enqueue) ? 1u : 0u);
    nAcc = waveSum(nAcc);
    if (laneId() == 0) {
        if (n) atomicAdd(&stats->paths, (unsigned long long)n);
        if (nCulled) atomicAdd(&stats->raysClosest, (unsigned long long)nCulled);
        if (nAcc) atomicAdd(&stats->accumulates, (unsigned long long)nAcc);
    }
}

// one workgroup per injected pass: its Counters block (a few hundred words) back to zero
__global__ __launch_bounds__(256) void k_zero_counters(CounterList list)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(list.ctr[blockIdx.x]);
    for (uint32_t i = threadIdx.x; i < sizeof(Counters) / 4; i += 256) w[i] = 0u;
}
// The step table comes to the device by a kernel that reads its pinned host entry (hr_core.hip: Group::hTables), 16 bytes per thread
__global__ __launch_bounds__(256) void k_fetch_table(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t n16)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
void launchFetchTable(hipStream_t stream, const void *hostMapped, void *dst, size_t bytes)
{
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    hipLaunchKernelGGL(k_fetch_table, dim3((n16 + 255u) / 256u), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(hostMapped), reinterpret_cast<uint4 *>(dst), n16);
}
void launchZeroCounters(const LaunchCfg &cfg, const CounterList &list)
{
    if (list.n > 0) hipLaunchKernelGGL(k_zero_counters, dim3(list.n), dim3(256), 0, cfg.stream, list);
}

// ------------------------------------------------------------------------------------------ resolve
// The finished passes' samples are added one after the other, in pass order (float addition order is part of the contract)
__global__ __launch_bounds__(kBlock) void k_resolve(FrameDev fr, PassBufList bufs)
{
    int x = 0, y = 0;
    if (!ownedPixel(fr, blockIdx.x * kBlock + threadIdx.x, x, y)) return;
    const uint32_t pixel = (uint32_t)(y * fr.W + x);
    float4 a = reinterpret_cast<float4 *>(fr.fb)[pixel];
    for (int k = 0; k < bufs.n; ++k) {
        float4 s = reinterpret_cast<const float4 *>(bufs.buf[k])[pixel];
        if (bufs.bufB[k]) { // HR_ESTIMATOR_ALL_LIGHTS: the pass's four partial sums meet here, in order, then the sample joins the frame
            const size_t framePixels = (size_t)(bufs.bufB[k] - bufs.buf[k]) >> 2;
            for (int j = 1; j <= 3; ++j) {
                const float4 t = reinterpret_cast<const float4 *>(bufs.buf[k])[pixel + j * framePixels];
                s.x = s.x + t.x, s.y = s.y + t.y, s.z = s.z + t.z;
            }
        }
        a.x = a.x + s.x, a.y = a.y + s.y, a.z = a.z + s.z, a.w = a.w + s.w;
    }
    reinterpret_cast<float4 *>(fr.fb)[pixel] = a;
}

// ----------------------------------------------------------------------------------- shard exchange
// dense copy of a rank's pixels, in ownedPixel order (coalesced 8x8 blocks); `unpack` is the inverse into a full frame
__global__ __launch_bounds__(kBlock) void k_pack_owned(FrameDev fr, const float4 *__restrict__ frame, float4 *__restrict__ packed, int unpack,
                                                       float4 *__restrict__ full)
{
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (uint32_t)(fr.nOwnedTiles * fr.tile * fr.tile)) return;
    int x = 0, y = 0;
    const bool in = ownedPixel(fr, gid, x, y);
    const uint32_t pixel = (uint32_t)(y * fr.W + x);
    if (unpack) {
        if (in) full[pixel] = packed[gid];
    } else {
        packed[gid] = in ? frame[pixel] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

void launchPackOwned(const LaunchCfg &cfg, const FrameDev &fr, const float *frame, float *packed, int unpack, float *full)
{
    const int threads = fr.nOwnedTiles * fr.tile * fr.tile;
    if (threads <= 0) return;
    hipLaunchKernelGGL(k_pack_owned, dim3((threads + kBlock - 1) / kBlock), dim3(kBlock), 0, cfg.stream, fr, reinterpret_cast<const float4 *>(frame),
                       reinterpret_cast<float4 *>(packed), unpack, reinterpret_cast<float4 *>(full));
}

// ------------------------------------------------------------------------------------------ display
// displayGL.frag on the accumulation buffer: one thread per pixel, row-major (coalesced 16-byte reads, 4- or 16-byte writes)
__global__ __launch_bounds__(kBlock) void k_display(FrameDev fr, hr_display_params P, int format, void *__restrict__ out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint32_t)(fr.W * fr.H)) return;
    const int x = (int)(i % (uint32_t)fr.W), y = (int)(i / (uint32_t)fr.W);
    const bool owned = (((y / fr.tile) * fr.tilesX + (x / fr.tile)) % fr.world) == fr.rank;
    const float4 px = reinterpret_cast<const float4 *>(fr.fb)[i];
    if (format == HR_DISPLAY_HDR_RGBA32F) { // saveScreenshot's HDR path (HeatrayRenderer.cpp:1633-1645)
        float4 o = make_float4(0.0f, 0.0f, 0.0f, owned ? px.w : 0.0f);
        if (owned && px.w != 0.0f) {
            const float divisor = 1.0f / px.w;
            o.x = px.x * divisor, o.y = px.y * divisor, o.z = px.z * divisor;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
        return;
    }
    float c[3] = {0.0f, 0.0f, 0.0f};
    if (owned) displayFragment(px, ((float)x + 0.5f) / (float)fr.W, ((float)y + 0.5f) / (float)fr.H, P, c);
    if (format == HR_DISPLAY_RGBA32F)
        reinterpret_cast<float4 *>(out)[i] = make_float4(c[0], c[1], c[2], owned ? 1.0f : 0.0f);
    else
        reinterpret_cast<uint32_t *>(out)[i] = owned ? (toByte(c[0]) | (toByte(c[1]) << 8) | (toByte(c[2]) << 16) | 0xFF000000u) : 0u;
}

// -------------------------------------------------------------------------------------------- trace
// Work items of one launch: for every in-flight pass k, its closest-hit queue followed by its occlusion
// queue.  segStart[2k] / segStart[2k+1] are the first global indices of the two.
HRD void buildSegStarts(const StepTable *tbl, uint32_t *segStart /* LDS, 2*kMaxSegs+1 */, bool closestOnly, bool skipPackets = false)
{
    // Queue lengths are read by one thread per queue, all at once (a serial loop over up to 96 passes, two dependent
    // global loads each, used to cost ~0.15 ms at the start of every launch on a small shard); then the first wave turns
    // the lengths into exclusive prefix sums, four entries per lane.
    const int n = tbl->nSeg;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const SegDev &sg = tbl->seg[k];
        segStart[2 * k] = (sg.closestEnabled && !(skipPackets && sg.packets)) ? closestCount(sg) : 0u;
        segStart[2 * k + 1] = closestOnly ? 0u : occlusionCount(sg);
        if (blockIdx.x == 0) { // (a queue longer than what the host provided for: its tail was dropped when it was written)
            if (sg.closestEnabled && sg.packets != 2u && *sg.qCountIn > closestCap(sg)) queueOverflow(tbl, OVF_CLOSEST_IN, (uint32_t)k, *sg.qCountIn);
            if (!closestOnly && *sg.sCountIn > sg.sInCap) queueOverflow(tbl, OVF_OCCLUSION_IN, (uint32_t)k, *sg.sCountIn);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int m = 2 * n; // entries to scan; entry m receives the total
        constexpr int kPer = (2 * kMaxSegs + 1 + 63) / 64; // entries per lane of the first wave
        const int first = (int)threadIdx.x * kPer;
        uint32_t v[kPer];
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            v[j] = (first + j < m) ? segStart[first + j] : 0u;
            sum += v[j];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        uint32_t acc = incl - sum;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (first + j <= m) segStart[first + j] = acc;
            acc += v[j];
        }
    }
    __syncthreads();
}


#ifndef HR_TAIL_ROUNDS
#define HR_TAIL_ROUNDS 3
#endif
static const unsigned long long kNoHitKey = ~0ull;

#ifdef HR_TAILPROF
// Experiment builds only: when does the work queue run dry, when does the launch end, how long is the longest ray?
__device__ unsigned long long g_tailprof[24]; // [0] min start clock, [1] min exhaustion clock, [2] max end clock, [3] max steps of a ray, [4] sum steps, [5] rays
#endif

// The host's view of the queue lengths (StepTable::hostCounts): the launch's first workgroup stores the closest-hit queue length of every
// table entry to pinned host memory (system scope), then the step's number.  Not inlined: k_trace sits exactly at the register count that
// gives five waves per SIMD, and this prologue must not move it (inlined it cost 4 VGPRs: four waves, -4.5 %).
__device__ __attribute__((noinline)) void reportQueueLengths(const StepTable *tbl, const uint32_t *segStart)
{
    if (!tbl->hostCounts) return;
    for (int k = (int)threadIdx.x; k < tbl->nSeg; k += kTraceBlock)
        __hip_atomic_store(&tbl->hostCounts[k], !tbl->seg[k].closestEnabled ? 0u : (tbl->seg[k].packets == 2u ? closestCap(tbl->seg[k]) : closestCount(tbl->seg[k])), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM); // (packets == 2: the packet kernel runs BESIDE this one and is still filling the queue: its capacity is the bound)
    if (threadIdx.x < 4 && tbl->hostProbe) // (the packet probe's totals so far)
        __hip_atomic_store(&tbl->hostProbe[threadIdx.x], __hip_atomic_load(&tbl->probe[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(tbl->hostSeq, tbl->seqValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the slab test's per-ray constants for the 32-byte nodes k_trace walks (the grid is folded into them: hr_trace.h)
HRD RayK traceFrame(const SceneDev &S, v3 o, v3 d)
{
    const float idx = safeInv(d.x), idy = safeInv(d.y), idz = safeInv(d.z);
    return rayFrame32(S, o, idx, idy, idz);
}

template <bool STATS>
__global__ __launch_bounds__(kTraceBlock, 5) void k_trace(const SceneDev *__restrict__ Sp, const int *__restrict__ leafKeys, const Node32 *__restrict__ nodes32,
                                                  const Tri *__restrict__ tris, StepTable *__restrict__ tbl, Stats *stats)
{
    __shared__ int stack[kTraceWaves][kStackLDS][64];
    __shared__ uint32_t segStart[2 * kMaxSegs + 1];
    // merge slots of the drain phase (below): one per ray a wave held when the work queue ran dry
    __shared__ unsigned long long mKey[kTraceWaves][64]; // min over the ray's fragments of (t bits, prim, face bit); kNoHitKey: none
    __shared__ uint32_t mCount[kTraceWaves][64];         // fragments still traversing
    __shared__ float2 mUV[kTraceWaves][64];              // barycentrics that belong to mKey
    __shared__ uint32_t mDonor[kTraceWaves][64];         // k-th donating lane of this round
    const SceneDev &S = *Sp;
    stats += blockIdx.x & (kStatSlots - 1);
    const unsigned long long clk0 = wall_clock64();
    buildSegStarts(tbl, segStart, false, true); // (camera rays that travel as packets are k_raygen_packets' business)
    const int nSeg2 = 2 * tbl->nSeg;
    if (blockIdx.x == 0) reportQueueLengths(tbl, segStart);
    const uint32_t total = segStart[nSeg2];
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    int *stackLane = &stack[wave][0][lane];
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    const int kRefill = tbl->refillLanes, kTriPhase = tbl->triPhaseLanes;
    const uint32_t fetchMax = (uint32_t)tbl->fetchMax, fetchMin = (uint32_t)tbl->fetchMin;
    // Camera rays are coherent: a wave that works through a LONGER run of consecutive pixels refills its idle lanes with neighbours of
    // the rays it still holds, so its lanes keep walking the same part of the tree (a launch of camera rays alone: 9.3 instead of 11.0 ms
    // with 256- instead of 64-ray chunks); the incoherent rays of later stages gain nothing from that and balance better in small chunks
    // (profiles/r3am_fetch_chunks.txt).  The passes injected this step are the last entries of the table.
    // (only when every resident wave gets at least eight such chunks: with fewer — a tile shard — the coarser grain costs more in balance
    // than the coherence brings)
    const uint32_t primaryStart = tbl->primaryFromSeg < tbl->nSeg ? segStart[2 * tbl->primaryFromSeg] : 0xFFFFFFFFu;
    const uint32_t fmPrimary = (uint32_t)tbl->fetchMaxPrimary & 0xFFFFu, fmGate = (uint32_t)tbl->fetchMaxPrimary >> 16;
    const bool longRuns = primaryStart < total && (unsigned long long)(total - primaryStart) >= (unsigned long long)fmGate * fmPrimary * gridDim.x * kTraceWaves;
    const uint32_t fetchMaxPrimary = (longRuns && fmPrimary > fetchMax) ? fmPrimary : fetchMax;
    const uint32_t headsLog2 = tbl->headsLog2, nHeads = 1u << headsLog2;
    const uint32_t wavesTimes2PerRange = ((2u * gridDim.x * kTraceWaves) >> headsLog2) + 1u;
    uint32_t home = blockIdx.x & (nHeads - 1u); // wave-uniform: the range of the index space this wave fetches from
    uint32_t lastBase = 0; // wave-uniform: where the global cursor stood at this wave's previous reservation

    // ---- per-lane traversal state (one ray per lane, refilled from the work pool when a lane finishes)
    int cur = kSentinel, sp = 0;
    uint32_t item = 0xFFFFFFFFu; // global work index of the ray this lane holds
    int segIdx = 0;              // 2k: closest-hit ray of pass k, 2k+1: occlusion ray of pass k
    uint32_t local = 0;          // index inside that queue
    v3 o(0.0f), d(0.0f);
    float idx = 0, idy = 0, idz = 0, oix = 0, oiy = 0, oiz = 0, tmax = 0, tlim = 0;
    uint32_t skipPrim = 0xFFFFFFFFu;
    HitRec best;
    best.prim = kMissPrim, best.t = 0, best.u = 0, best.v = 0;
    int ovf[kStackOvf];
    const float tmin = S.rayEps, hitPad = S.hitPad;
    const int rootRef = (S.nTris == 0) ? kSentinel : (S.rootLeafCount > 0 ? ~(0 | ((S.rootLeafCount - 1) << 28)) : 0);

    uint32_t poolLo = 0, poolHi = 0; // wave-uniform: indices this wave has reserved and not handed out yet
    bool exhausted = false;          // wave-uniform: the global cursor ran past the end
    // Small launches (the late stages of a pass hold a few hundred to a few thousand rays; a tile shard's stages even earlier): up to
    // tbl->staticPerWave rays per resident wave are dealt out statically — wave g takes items [g * per, (g + 1) * per) — instead of
    // through the global cursor.  No same-address atomics (5120 waves learning from one counter that nothing is left took ~60 us of
    // every such launch), every CU gets some of the rays instead of a few waves getting 64 each, and the lanes a wave has left over
    // take subtrees of its rays at once (the drain phase below): a late-stage launch went from ~210 to 50-80 us.  With only a few
    // 64-ray chunks per wave the cursor also balances badly (+-1 chunk is +-40 %): 865 k rays on 5120 waves take 0.80 ms dealt out
    // against 1.30 ms fetched; from ~500 rays per wave on the cursor wins (profiles/r3e_static_deal.txt).
    const uint32_t gridWaves = gridDim.x * kTraceWaves;
    const bool staticDeal = (unsigned long long)total <= (unsigned long long)gridWaves * (unsigned long long)(tbl->staticPerWave > 0 ? tbl->staticPerWave : 0);
    if (staticDeal) {
        const uint32_t per = (total + gridWaves - 1u) / gridWaves;
        const uint32_t gw = wave * gridDim.x + blockIdx.x; // the first gridDim.x chunks go to different workgroups
        const unsigned long long lo = (unsigned long long)gw * per;
        poolLo = lo < total ? (uint32_t)lo : total;
        poolHi = (lo + per < total) ? (uint32_t)(lo + per) : total;
        if (poolLo == poolHi) exhausted = true;
    }
    uint32_t nvC = 0, ntC = 0, nvA = 0, ntA = 0, nacc = 0;

    int pend = 0; // postponed leaf (a negative leaf reference) or 0: the lane keeps descending while a leaf waits
    uint32_t slot = lane;  // merge slot of the ray (fragment) this lane holds during the drain phase
    bool draining = false; // wave-uniform: the merge slots are initialised
#ifdef HR_TAILPROF
    const unsigned long long tStart = wall_clock64();
    const unsigned long long cStart = clock64(); // shader clock, against the 100 MHz wall clock: the frequency the kernel really ran at
    unsigned long long tExh = 0;
    uint32_t mySteps = 0, maxSteps = 0;
    unsigned long long sumSteps = 0, nRays = 0, nGiven = 0, drainIters = 0, drainLanes = 0;
    unsigned long long triPhases = 0, triLanes = 0, nodeRounds = 0, nodeLanes = 0; // (wave-level, counted by lane 0)
#endif
    for (;;) {
        // ---------------- refill idle lanes (persistent threads with dynamic fetch)
        bool idle = (item == 0xFFFFFFFFu);
        unsigned long long idleMask = __ballot(idle);
        int nIdle = __popcll(idleMask);
        if (!exhausted && (nIdle >= kRefill || nIdle == 64)) {
            for (int round = 0; round < 2 && nIdle > 0; ++round) {
                if (poolLo == poolHi) { // reserve another chunk of the global index space
                    if (staticDeal) { // (this wave's share of a small launch has been handed out)
                        exhausted = true;
#ifdef HR_TAILPROF
                        tExh = wall_clock64();
#endif
                        break;
                    }
                    // the next chunk of this wave's current range, or of the next range that still has work (hr_kernels.h: StepTable::heads)
                    bool got = false;
                    uint32_t base = 0, hi = 0;
                    for (uint32_t tries = 0; tries < nHeads; ++tries) {
                        const uint32_t rLo = (uint32_t)(((unsigned long long)total * home) >> headsLog2);
                        const uint32_t rHi = (uint32_t)(((unsigned long long)total * (home + 1u)) >> headsLog2);
                        // chunk ~ (work left in the range) / (2 x the waves that started on it), from the cursor value this wave saw last
                        const uint32_t left = (lastBase >= rLo && lastBase < rHi) ? rHi - lastBase : rHi - rLo;
                        uint32_t chunk = left / wavesTimes2PerRange;
                        const uint32_t fm = (tries == 0 && lastBase >= primaryStart) ? fetchMaxPrimary : fetchMax;
                        chunk = chunk > fm ? fm : (chunk < fetchMin ? fetchMin : chunk);
                        uint32_t off = 0;
                        if (lane == 0) off = atomicAdd(&tbl->heads[home * 32u], chunk);
                        off = __shfl(off, 0);
                        if (off < rHi - rLo) {
                            base = rLo + off;
                            hi = (off + chunk < rHi - rLo) ? base + chunk : rHi;
                            got = true;
                            break;
                        }
                        home = (home + 1u) & (nHeads - 1u);
                    }
                    if (!got) {
                        exhausted = true;
#ifdef HR_TAILPROF
                        tExh = wall_clock64();
#endif
                        break;
                    }
                    lastBase = base;
                    poolLo = base;
                    poolHi = hi;
                }
                const uint32_t avail = poolHi - poolLo;
                const uint32_t rank = (uint32_t)__popcll(idleMask & ltMask);
                if (idle && rank < avail) {
                    item = poolLo + rank;
                    // which queue does the item belong to (at most 2*kMaxSegs entries)
                    int sI = 0, sHiB = nSeg2 - 1; // last queue whose first index is <= item
                    while (sI < sHiB) {
                        const int mid = (sI + sHiB + 1) >> 1;
                        if (item >= segStart[mid])
                            sI = mid;
                        else
                            sHiB = mid - 1;
                    }
                    segIdx = sI;
                    local = item - segStart[sI];
                    const SegDev &sg = tbl->seg[sI >> 1];
                    float4 a, b;
                    if (sI & 1) { // occlusion ray
                        a = G(sg.sqIn.A)[local], b = G(sg.sqIn.B)[local];
                        skipPrim = __float_as_uint(b.w);
                    } else {
                        a = G(sg.qin.A)[local], b = G(sg.qin.B)[local];
                        skipPrim = (uint32_t)G(sg.qin.D)[local].z;
                    }
                    o = v3(a.x, a.y, a.z), d = v3(b.x, b.y, b.z);
                    tmax = a.w, tlim = a.w;
                    {
                        const RayK f = traceFrame(S, o, d);
                        idx = f.idx, idy = f.idy, idz = f.idz, oix = f.oix, oiy = f.oiy, oiz = f.oiz;
                    }
                    best.prim = kMissPrim, best.t = tmax, best.u = 0.0f, best.v = 0.0f;
                    sp = 0;
                    pend = 0;
                    cur = rootRef;
                    idle = false;
                }
                const uint32_t taken = avail < (uint32_t)nIdle ? avail : (uint32_t)nIdle;
                poolLo += taken;
                idleMask = __ballot(idle);
                nIdle = __popcll(idleMask);
            }
        }
        // ---------------- drain phase: the queue is empty, so a launch now lasts as long as its longest ray (0.5 ms for a ray of
        // ~400 node steps, against ~70 on average).  Idle lanes therefore take over pending subtrees of the rays still in
        // flight in their wave: the closest hit is the lexicographic minimum of (t, prim) over ALL triangles, so it does not
        // matter which lane visits which subtree; the fragments of a ray meet in its merge slot.
        if (exhausted) {
            if (!draining) {
                draining = true;
                slot = lane;
                mKey[wave][lane] = kNoHitKey;
                mCount[wave][lane] = (item != 0xFFFFFFFFu) ? 1u : 0u;
            }
            if (item != 0xFFFFFFFFu) { // what the other fragments of this ray have found so far bounds this one too
                const unsigned long long k = mKey[wave][slot];
                if (k != kNoHitKey) {
                    if (segIdx & 1) {
                        cur = kSentinel, sp = 0, pend = 0; // occluded: nothing left to find
                    } else {
                        const float ts = __uint_as_float((uint32_t)(k >> 32));
                        tlim = ts < tlim ? ts : tlim;
                    }
                }
            }
            for (int round = 0; round < HR_TAIL_ROUNDS; ++round) { // a lane gives one subtree per round
            const bool canGive = item != 0xFFFFFFFFu && sp >= 1 && sp <= kStackLDS; // (entries beyond kStackLDS are private)
            const unsigned long long giveMask = __ballot(canGive);
            if (nIdle == 0 || giveMask == 0ull) break;
            {
                const uint32_t nGive = (uint32_t)__popcll(giveMask);
                const uint32_t nMove = nGive < (uint32_t)nIdle ? nGive : (uint32_t)nIdle;
                const uint32_t giveRank = (uint32_t)__popcll(giveMask & ltMask), idleRank = (uint32_t)__popcll(idleMask & ltMask);
                // (LDS hand-offs between lanes of ONE wave: the hardware executes a wave's LDS instructions in order, and the wave barriers
                // keep the compiler from moving or caching the plain accesses across them)
                if (canGive && giveRank < nMove) mDonor[wave][giveRank] = lane;
                __builtin_amdgcn_wave_barrier();
                const bool takes = idle && idleRank < nMove;
                const uint32_t src = takes ? ((volatile uint32_t *)mDonor[wave])[idleRank] : lane;
                // the ray travels by cross-lane reads (every lane executes them), the subtree through the donor's stack column
                const float sox = __shfl(o.x, (int)src), soy = __shfl(o.y, (int)src), soz = __shfl(o.z, (int)src);
                const float sdx = __shfl(d.x, (int)src), sdy = __shfl(d.y, (int)src), sdz = __shfl(d.z, (int)src);
                const float sTmax = __shfl(tmax, (int)src), sTlim = __shfl(tlim, (int)src);
                const uint32_t sSkip = (uint32_t)__shfl((int)skipPrim, (int)src), sItem = (uint32_t)__shfl((int)item, (int)src);
                const uint32_t sLocal = (uint32_t)__shfl((int)local, (int)src), sSlot = (uint32_t)__shfl((int)slot, (int)src);
                const int sSeg = __shfl(segIdx, (int)src);
                const int given = ((volatile int *)stack[wave][0])[src]; // the donor's OLDEST entry: the farthest subtree, usually the largest
                __builtin_amdgcn_wave_barrier(); // (read by the taker before the donor compacts its stack)
                if (canGive && giveRank < nMove) {
                    sp -= 1;
                    if (sp > 0) stackLane[0] = stackLane[sp * 64];
                }
                if (takes) {
                    item = sItem, segIdx = sSeg, local = sLocal, slot = sSlot, skipPrim = sSkip;
                    o = v3(sox, soy, soz), d = v3(sdx, sdy, sdz);
                    tmax = sTmax, tlim = sTlim;
                    {
                        const RayK f = traceFrame(S, o, d);
                        idx = f.idx, idy = f.idy, idz = f.idz, oix = f.oix, oiy = f.oiy, oiz = f.oiz;
                    }
                    best.prim = kMissPrim, best.t = tmax, best.u = 0.0f, best.v = 0.0f;
                    sp = 0, pend = 0, cur = given;
                    atomicAdd(&mCount[wave][slot], 1u);
                    idle = false;
#ifdef HR_TAILPROF
                    nGiven += 1;
#endif
                }
                nIdle -= (int)nMove;
                idleMask = __ballot(idle);
            }
            }
        }
        if (nIdle == 64) { // nothing in flight (finished rays were retired at the end of the previous round)
            if (!exhausted) continue;
            break;
        }
#ifdef HR_TAILPROF
        if (exhausted) drainIters += 1, drainLanes += (unsigned long long)(64 - nIdle);
#endif

        const bool isAny = (segIdx & 1) != 0;
        // ---------------- inner-node steps for every lane that holds an inner node
#pragma unroll
        for (int rep = 0; rep < HR_NODE_STEPS; ++rep) {
#ifdef HR_TAILPROF
            {
                const unsigned long long m = __ballot(cur >= 0 && cur != kSentinel);
                if (m) nodeRounds += 1, nodeLanes += (unsigned long long)__popcll(m);
            }
#endif
            if (cur >= 0 && cur != kSentinel) {
                if (STATS) {
                    if (isAny)
                        ++nvA;
                    else
                        ++nvC;
                }
#ifdef HR_TAILPROF
                ++mySteps;
#endif
                const RayK rk{idx, idy, idz, oix, oiy, oiz};
                nodeStep32(nodes32, cur, sp, stackLane, ovf, rk, tmin, tlim); // (two loads per visit: hr_trace.h)
            }
            // a lane that reached a leaf postpones it and keeps descending (speculative traversal); with a leaf already
            // postponed it is blocked until the wave runs the triangle phase
            if (cur < 0 && pend == 0) {
                pend = cur;
                HR_POP();
            }
        }
        // ---------------- triangle phase: run it once enough lanes wait for it, or when nobody can descend any more
        const unsigned long long blockedMask = __ballot(pend != 0 && (cur < 0 || cur == kSentinel));
        const unsigned long long nodeMask = __ballot(cur >= 0 && cur != kSentinel);
        // (while draining, lane utilisation no longer matters: a waiting leaf is tested at once)
        if (blockedMask != 0ull && (__popcll(blockedMask) >= (exhausted ? 1 : kTriPhase) || nodeMask == 0ull)) {
#ifdef HR_TAILPROF
            triPhases += 1, triLanes += (unsigned long long)__popcll(__ballot(pend != 0));
#endif
            if (pend != 0) {
                const int enc = ~pend;
                // a leaf child of node `enc >> 2` in slot 3 - (enc & 3) (hr_trace.h: nodeStep32): its triangle's index is ~(leafKeys[node] + slot),
                // Node4::c.w of that node in a compact array that stays in L2 (a root leaf — a scene of at most four triangles, no nodes at
                // all — keeps the (first, count) form).  Read HERE, in front of the triangle's loads: reading it where the leaf is put aside
                // (six more load sites in the unrolled node steps) measured 2-5 % slower (profiles/r5_node32_ab.txt)
                int first = enc & 0x0FFFFFFF, count = (enc >> 28) + 1;
                if (rootRef >= 0) first = ~(leafKeys[enc >> 2] + (3 - (enc & 3))), count = 1;
                pend = 0;
                for (int k = 0; k < count; ++k) {
                    const Tri &tr = tris[first + k];
                    const float4 tp = tr.p, tq = tr.q, trr = tr.r;
                    if (STATS) {
                        if (isAny)
                            ++ntA;
                        else
                            ++ntC;
                    }
                    const uint32_t prim = __float_as_uint(trr.y);
                    if (prim == skipPrim) continue;
                    const v3 v0(tp.x, tp.y, tp.z), e1(tp.w, tq.x, tq.y), e2(tq.z, tq.w, trr.x);
                    // Möller–Trumbore; the operation order is part of the arithmetic contract
                    const v3 pvec = cross(d, e2);
                    const float det = dot(e1, pvec);
                    if (det == 0.0f) continue;
                    const float inv = 1.0f / det;
                    const v3 tvec = o - v0;
                    const float u = dot(tvec, pvec) * inv;
                    if (!(u >= 0.0f) || u > 1.0f) continue;
                    const v3 qvec = cross(tvec, e1);
                    const float v = dot(d, qvec) * inv;
                    if (!(v >= 0.0f) || u + v > 1.0f) continue;
                    const float t = dot(e2, qvec) * inv;
                    if (!(t > tmin) || !(t < tmax)) continue;
                    // (the hit test's second half, hr_trace.h.  On the live triangle: re-reading it from L1 one axis at a time, to shorten
                    // the live ranges, measured 0.5-1.5 % slower — profiles/r5b_hitbox_ab.txt.  The kernel keeps its five waves per SIMD
                    // because its launch bounds say so: the compiler then allocates for 96 registers without spilling.)
                    if (!hitInTriBox(v0, e1, e2, o, d, t, hitPad)) continue;
                    if (isAny) {
                        if ((__float_as_uint(trr.z) & TF_NON_OCCLUDER) && alphaPasses(S, prim, u, v)) continue;
                        best.prim = 0u; // occluded (anything but kMissPrim)
                        cur = kSentinel;
                        sp = 0;
                        if (draining) atomicMin(&mKey[wave][slot], 0ull); // the ray's other fragments stop at their next round
                        break;
                    }
                    const uint32_t bp = best.prim & 0x7FFFFFFFu;
                    if (best.prim == kMissPrim || t < best.t || (t == best.t && prim < bp)) {
                        best.prim = prim | ((det > 0.0f) ? 0x80000000u : 0u);
                        best.t = t, best.u = u, best.v = v;
                        tlim = t;
                        if (draining) { // publish at once: subtrees handed to other lanes are speculative until a hit bounds them
                            const unsigned long long kk = ((unsigned long long)__float_as_uint(t) << 32) | ((unsigned long long)prim << 1) |
                                                          (unsigned long long)(det > 0.0f ? 1u : 0u);
                            atomicMin(&mKey[wave][slot], kk);
                            __builtin_amdgcn_wave_barrier();
                            if (((volatile unsigned long long *)mKey[wave])[slot] == kk) mUV[wave][slot] = make_float2(u, v);
                        }
                    }
                }
            }
        }
        // ---------------- retire finished rays
        if (draining && cur == kSentinel && pend == 0 && item != 0xFFFFFFFFu) {
            // a fragment is done: fold its result into the ray's slot; the last fragment writes the ray's result
            const bool hit = best.prim != kMissPrim;
            const unsigned long long myKey =
                !hit ? kNoHitKey
                     : (isAny ? 0ull
                              : (((unsigned long long)__float_as_uint(best.t) << 32) | ((unsigned long long)(best.prim & 0x7FFFFFFFu) << 1) |
                                 (unsigned long long)(best.prim >> 31)));
            if (hit) atomicMin(&mKey[wave][slot], myKey);
            __builtin_amdgcn_wave_barrier();
            if (hit && !isAny && ((volatile unsigned long long *)mKey[wave])[slot] == myKey) mUV[wave][slot] = make_float2(best.u, best.v);
            __builtin_amdgcn_wave_barrier();
            const uint32_t before = atomicSub(&mCount[wave][slot], 1u);
            if (before == 1u) {
                const unsigned long long k = ((volatile unsigned long long *)mKey[wave])[slot];
                const SegDev &sg = tbl->seg[segIdx >> 1];
                if (isAny) {
                    if (k == kNoHitKey) {
                        const float4 c = G(sg.sqIn.C)[local];
                        HR_GLOBAL float *px = G(sg.passbuf) + (size_t)__float_as_uint(c.w) * 4;
                        px[0] = px[0] + c.x;
                        px[1] = px[1] + c.y;
                        px[2] = px[2] + c.z;
                        ++nacc;
                    }
                } else {
                    HitRec h;
                    h.prim = kMissPrim, h.t = tmax, h.u = 0.0f, h.v = 0.0f;
                    if (k != kNoHitKey) {
                        const uint32_t lo = (uint32_t)k;
                        const float2 uv = make_float2(((volatile float *)&mUV[wave][slot])[0], ((volatile float *)&mUV[wave][slot])[1]);
                        h.prim = (lo >> 1) | (lo << 31), h.t = __uint_as_float((uint32_t)(k >> 32)), h.u = uv.x, h.v = uv.y;
                    }
                    G(sg.hits)[local] = h;
                }
            }
#ifdef HR_TAILPROF
            maxSteps = mySteps > maxSteps ? mySteps : maxSteps, sumSteps += mySteps, nRays += (before == 1u), mySteps = 0;
#endif
            item = 0xFFFFFFFFu;
        }
        if (cur == kSentinel && pend == 0 && item != 0xFFFFFFFFu) {
            const SegDev &sg = tbl->seg[segIdx >> 1];
            if (isAny) {
                if (best.prim == kMissPrim) { // unoccluded: the light's shader accumulates into the pass's sample
                    const float4 c = G(sg.sqIn.C)[local];
                    HR_GLOBAL float *px = G(sg.passbuf) + (size_t)__float_as_uint(c.w) * 4;
                    px[0] = px[0] + c.x;
                    px[1] = px[1] + c.y;
                    px[2] = px[2] + c.z;
                    ++nacc;
                }
            } else {
                G(sg.hits)[local] = best;
            }
#ifdef HR_TAILPROF
            maxSteps = mySteps > maxSteps ? mySteps : maxSteps, sumSteps += mySteps, nRays += 1, mySteps = 0;
#endif
            item = 0xFFFFFFFFu;
        }
    }
#ifdef HR_TAILPROF
    {
        const unsigned long long tEnd = wall_clock64();
        atomicMin(&g_tailprof[0], tStart);
        if (tExh) atomicMin(&g_tailprof[1], tExh);
        atomicMax(&g_tailprof[2], tEnd);
        atomicMax(&g_tailprof[3], (unsigned long long)maxSteps);
        atomicAdd(&g_tailprof[4], sumSteps);
        atomicAdd(&g_tailprof[5], nRays);
        atomicAdd(&g_tailprof[6], nGiven);
        if (blockIdx.x == 0 && threadIdx.x == 0) g_tailprof[18] = clock64() - cStart, g_tailprof[19] = tEnd - tStart;
        if (lane == 0) {
            atomicMax(&g_tailprof[7], drainIters);
            atomicAdd(&g_tailprof[16], drainLanes);
            atomicAdd(&g_tailprof[17], drainIters);
            atomicAdd(&g_tailprof[20], triPhases);
            atomicAdd(&g_tailprof[21], triLanes);
            atomicAdd(&g_tailprof[22], nodeRounds);
            atomicAdd(&g_tailprof[23], nodeLanes);
        }
        if (tExh && lane == 0) { // per-wave drain time in 0.05 ms buckets
            unsigned long long b = (tEnd - tExh) / 5000ull;
            atomicAdd(&g_tailprof[8 + (b > 15ull ? 15ull : b)], 1ull);
        }
    }
#endif

    nacc = waveSum(nacc);
    if (lane == 0 && nacc) atomicAdd(&stats->accumulates, (unsigned long long)nacc);
    if (STATS) {
        nvC = waveSum(nvC), ntC = waveSum(ntC), nvA = waveSum(nvA), ntA = waveSum(ntA);
        if (lane == 0) {
            atomicAdd(&stats->nodeVisits, (unsigned long long)nvC + nvA);
            atomicAdd(&stats->triTests, (unsigned long long)ntC + ntA);
            atomicAdd(&stats->nodeVisitsAny, (unsigned long long)nvA);
            atomicAdd(&stats->triTestsAny, (unsigned long long)ntA);
        }
    }
    // ---- the launch times itself (StepTable::clkStart): every workgroup folds its start and end into one of kClkSlots (min, max)
    // pairs — 80 atomics per address, spread over the launch; ONE counter of finished workgroups would put 1280 same-address
    // atomics (~12 ns each) into the tail of every launch.  The kernel behind this one (k_shade_sort) adds max - min to the counters.
    if (threadIdx.x == 0) {
        const uint32_t cs = blockIdx.x & (kClkSlots - 1);
        __hip_atomic_fetch_min(&tbl->clkStart[cs], clk0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(&tbl->clkEnd[cs], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long nc = 0, na = 0;
        for (int k = 0; k < tbl->nSeg; ++k) {
            nc += segStart[2 * k + 1] - segStart[2 * k];
            na += segStart[2 * k + 2] - segStart[2 * k + 1];
        }
        atomicAdd(&stats->raysClosest, nc);
        atomicAdd(&stats->raysAny, na);
    }
}

// ---------------------------------------------------------------------------------- trace, camera rays
// The closest-hit rays of a pass's FIRST stage are camera rays, and the camera rays of a few neighbouring pixels in the passes injected
// together are nearly the same ray (k_raygen_packets below).  64 such rays walk the tree as ONE packet: a single traversal state per wave — the
// node reference and the stack are wave-uniform, the node arrives through the scalar cache (no texture addresser), every lane tests the
// node's four child boxes with its own ray, and a child is entered when ANY lane's ray enters it.  A lane whose ray misses a subtree the
// wave walks anyway tests boxes and triangles it cannot hit: the hit is defined by the triangle test alone (hr_trace.h), so the result
// is the same closest hit, bit for bit; what it costs is the union of the 64 rays' node sets instead of their sum — and one node fetch,
// one box decode and one stack for 64 rays.  Children are ordered by the entry distance of the first lane that enters each.
#if defined(__HIP_DEVICE_COMPILE__)
#define HR_CONSTANT __attribute__((address_space(4))) // (uniform address + constant address space = scalar loads)
#else
#define HR_CONSTANT
#endif
typedef const Node4 HR_CONSTANT *ConstNodes;
typedef const Tri HR_CONSTANT *ConstTris;
static const int kPacketBlock = 64;

struct LaneStack { // wave-uniform stack held in the LANES of three registers: entry i is lane (i & 63) of register (i >> 6)
    int r0, r1, r2;
    static HRD void writeLane(int &r, int v, int l) // (clang has no builtin for it; value and lane are wave-uniform)
    {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(r) : "s"(v), "s"(l) : "m0"); // (one SGPR operand per instruction: the lane goes through M0)
    }
    HRD void push(int sp, int v)
    {
        if (sp < 64)
            writeLane(r0, v, sp);
        else if (sp < 128)
            writeLane(r1, v, sp - 64);
        else
            writeLane(r2, v, sp - 128);
    }
    HRD int at(int sp) const
    {
        if (sp < 64) return __builtin_amdgcn_readlane(r0, sp);
        if (sp < 128) return __builtin_amdgcn_readlane(r1, sp - 64);
        return __builtin_amdgcn_readlane(r2, sp - 128);
    }
};
static_assert(kStackLDS + kStackOvf <= 192, "the packet stack holds the deepest tree the builder can make");

HRD void cswapS(uint32_t &a, uint32_t &b)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo, b = hi;
}

// The hit test (hr_trace.h: Möller–Trumbore, then the hit point inside the triangle's half-padded box; the operation order is part of the
// arithmetic contract) on triangles [first, first + count) of the leaf-ordered array, every lane with its own ray
HRD void packetTriangles(ConstTris tris, int first, int count, v3 o, v3 d, float tmin, float tmax, float hitPad, uint32_t skipPrim, HitRec &best, float &tlim)
{
    for (int k = 0; k < count; ++k) {
        const float4 tp = tris[first + k].p, tq = tris[first + k].q, trr = tris[first + k].r;
        const uint32_t prim = __float_as_uint(trr.y);
        if (prim == skipPrim) continue;
        const v3 v0(tp.x, tp.y, tp.z), e1(tp.w, tq.x, tq.y), e2(tq.z, tq.w, trr.x);
        const v3 pvec = cross(d, e2);
        const float det = dot(e1, pvec);
        if (det == 0.0f) continue;
        const float inv = 1.0f / det;
        const v3 tvec = o - v0;
        const float u = dot(tvec, pvec) * inv;
        if (!(u >= 0.0f) || u > 1.0f) continue;
        const v3 qvec = cross(tvec, e1);
        const float v = dot(d, qvec) * inv;
        if (!(v >= 0.0f) || u + v > 1.0f) continue;
        const float t = dot(e2, qvec) * inv;
        if (!(t > tmin) || !(t < tmax)) continue;
        if (!hitInTriBox(v0, e1, e2, o, d, t, hitPad)) continue;
        const uint32_t bp = best.prim & 0x7FFFFFFFu;
        if (best.prim == kMissPrim || t < best.t || (t == best.t && prim < bp)) {
            best.prim = prim | ((det > 0.0f) ? 0x80000000u : 0u);
            best.t = t, best.u = u, best.v = v;
            tlim = t;
        }
    }
}

// One packet: every lane walks the wave's traversal with its own ray (o, d, tmax; a lane without a ray passes tmax = 0: it enters
// nothing and hits nothing).  A lane tests every triangle the PACKET reaches, whether or not its own ray entered the triangle's box:
// the hit test's answer does not depend on which triangles a traversal tests (hr_trace.h: hitInTriBox — a hit point lies inside every
// box above its triangle, so a ray that hits is a ray whose own traversal would have got there; rounds 4's per-lane "own box" bits,
// which made a lane test exactly what k_trace tests, are gone with the phantom hits they were there for).  STATS counts, per lane, the
// inner children (`nv`) and leaf children (`nt`) the lane's OWN box test entered — the figures of that ray traced alone; PROBE also
// counts, per node step, the children the PACKET entered (`entered`, wave-uniform) and the children this lane's own box test entered
// (`own`): what the packet costs against its rays traced one by one.
template <bool STATS, bool PROBE>
HRD void packetTraverse(const SceneDev &S, ConstNodes nodes, ConstTris tris, v3 o, v3 d, float tmax, uint32_t skipPrim, HitRec &best, uint32_t &nv, uint32_t &nt,
                        uint32_t &entered, uint32_t &own)
{
    const float tmin = S.rayEps;
    float tlim = tmax;
    const float idx = safeInv(d.x), idy = safeInv(d.y), idz = safeInv(d.z);
    const RayK rk = rayFrame(o, idx, idy, idz);
    best.prim = kMissPrim, best.t = tmax, best.u = 0.0f, best.v = 0.0f;
    LaneStack stk{0, 0, 0};
    int sp = 0;
    int cur = (S.nTris == 0) ? kSentinel : (S.rootLeafCount > 0 ? ~(0 | ((S.rootLeafCount - 1) << 28)) : 0);
    if (STATS) nv += (cur >= 0 && cur != kSentinel) ? 1u : 0u, nt += cur < 0 ? (uint32_t)S.rootLeafCount : 0u; // (the root)
    while (cur != kSentinel) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        if (cur >= 0) {
            const float4 a = nodes[cur].a;
            const uint4 qb = nodes[cur].b, qc = nodes[cur].c;
            const uint32_t meta = __float_as_uint(a.w);
            const uint32_t nInner = (meta >> 24) & 7u, nValid = meta >> 27;
            const int innerBase = (int)qc.z, leafKey = (int)qc.w;
            const float bx = __uint_as_float((meta & 0xFFu) << 23) * rk.idx;
            const float by = __uint_as_float(((meta >> 8) & 0xFFu) << 23) * rk.idy;
            const float bz = __uint_as_float(((meta >> 16) & 0xFFu) << 23) * rk.idz;
            const float ax = __builtin_fmaf(a.x, rk.idx, -rk.oix), ay = __builtin_fmaf(a.y, rk.idy, -rk.oiy), az = __builtin_fmaf(a.z, rk.idz, -rk.oiz);
            const uint32_t nX = rk.idx < 0.0f ? qb.w : qb.x, fX = rk.idx < 0.0f ? qb.x : qb.w;
            const uint32_t nY = rk.idy < 0.0f ? qc.x : qb.y, fY = rk.idy < 0.0f ? qb.y : qc.x;
            const uint32_t nZ = rk.idz < 0.0f ? qc.y : qb.z, fZ = rk.idz < 0.0f ? qb.z : qc.y;
            uint32_t key[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                key[c] = 0xFFFFFFFFu;
                if ((uint32_t)c >= nValid) continue; // (wave-uniform: a node has 3.2 children on average, the slots behind them cost nothing here)
                const float tnx = __builtin_fmaf((float)byteOf(nX, c), bx, ax), tfx = __builtin_fmaf((float)byteOf(fX, c), bx, ax);
                const float tny = __builtin_fmaf((float)byteOf(nY, c), by, ay), tfy = __builtin_fmaf((float)byteOf(fY, c), by, ay);
                const float tnz = __builtin_fmaf((float)byteOf(nZ, c), bz, az), tfz = __builtin_fmaf((float)byteOf(fZ, c), bz, az);
                const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
                const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
                const bool enters = tn <= tf;
                const unsigned long long m = __ballot(enters);
                if (PROBE) own += enters ? 1u : 0u, entered += m ? 1u : 0u;
                if (STATS) {
                    if ((uint32_t)c < nInner)
                        nv += enters ? 1u : 0u;
                    else
                        nt += enters ? 1u : 0u;
                }
                // the wave's key of the child: the entry distance of the first lane that enters it
                if (m) key[c] = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn), __ffsll((long long)m) - 1) & ~3u) | (uint32_t)c;
            }
            cswapS(key[0], key[1]), cswapS(key[2], key[3]), cswapS(key[0], key[2]), cswapS(key[1], key[3]), cswapS(key[1], key[2]);
#pragma unroll
            for (int j = 3; j >= 1; --j)
                if (key[j] != 0xFFFFFFFFu) {
                    const int sl = (int)(key[j] & 3u);
                    stk.push(sp, (((uint32_t)sl < nInner) ? innerBase : leafKey) + sl);
                    ++sp;
                }
            if (key[0] != 0xFFFFFFFFu) {
                const int sl = (int)(key[0] & 3u);
                cur = (((uint32_t)sl < nInner) ? innerBase : leafKey) + sl;
            } else if (sp > 0) {
                --sp;
                cur = stk.at(sp);
            } else {
                cur = kSentinel;
            }
        } else { // a leaf: its triangle(s)
            const int enc = ~cur;
            packetTriangles(tris, enc & 0x0FFFFFFF, (enc >> 28) + 1, o, d, tmin, tmax, S.hitPad, skipPrim, best, tlim);
            if (sp > 0) {
                --sp;
                cur = stk.at(sp);
            } else {
                cur = kSentinel;
            }
        }
    }
}

// pixel j (Morton order: 2x2, 4x2, 4x4 ... blocks are contiguous ranges) of an 8x8 patch
HRD uint32_t mortonPixel(uint32_t j)
{
    const uint32_t px = (j & 1u) | ((j >> 1) & 2u) | ((j >> 2) & 4u), py = ((j >> 1) & 1u) | ((j >> 2) & 2u) | ((j >> 3) & 4u);
    return py * 8u + px;
}

// Ray generation and the camera rays' traversal in one kernel, for 2^passesLog2 passes injected together (segs.n of them): a wave is
// the rays of (64 >> passesLog2) neighbouring pixels — a 2x2 block for sixteen passes, the 8x8 patch for one — in all these passes.
// The rays a pixel sends in consecutive passes differ by the sub-pixel jitter (and the lens sample) only: no other 64 rays of a
// render are as close to each other, and the packet's union of node sets shrinks accordingly (c3, the triangle fog: 3.0 x its rays'
// own node tests for one pass of an 8x8 patch, 1.8 x for sixteen passes of 2x2 pixels; profiles/r4u_packets.txt).  Everything
// k_raygen does happens here through the same function (cameraLane: perspective.rlsl, the pass sample's zero, the root cull with its miss shader); queue slots
// are reserved per pass and workgroup through LDS counters.  k_trace leaves these passes' first-stage queues alone (SegDev::packets).
#ifndef HR_RP_BLOCK
#define HR_RP_BLOCK 256
#endif
static const int kRpBlock = HR_RP_BLOCK; // threads per workgroup: queue slots are reserved once per workgroup and pass
// UNIFORM: the passes differ in their sample index only (the usual batch: same camera, same options), so everything else of the pass
// parameters is read once per wave through the scalar cache instead of once per lane from sixteen different table entries.
template <bool STATS, bool UNIFORM>
__global__ __launch_bounds__(kRpBlock) void k_raygen_packets(const SceneDev *__restrict__ Sp, const Node4 *__restrict__ nodesG, const Tri *__restrict__ trisG,
                                                            const StepTable *__restrict__ tbl, SegList segs, int passesLog2, FrameDev fr, Stats *stats)
{
    __shared__ uint32_t cnt[kMaxBatch], firstSlot[kMaxBatch];
    const SceneDev &S = *Sp;
    if (threadIdx.x < (uint32_t)kMaxBatch) cnt[threadIdx.x] = 0u;
    stats += blockIdx.x & (kStatSlots - 1);
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const bool swz = (passesLog2 & 256) != 0;
    passesLog2 &= 255;
    const uint32_t nPass = 1u << passesLog2, pass = lane & (nPass - 1u), npx = 64u >> passesLog2;
    // (passesLog2 bit 8, HR_TUNE pswz=1: workgroups go to the 8 XCDs round robin by their index; dealt like this, the 1024 / pixelsPerBlock
    // workgroups of a 32x32 tile all land on one XCD — its L2 fetches the tile's part of the tree once instead of all eight doing so)
    uint32_t b = blockIdx.x;
    if (swz) {
        const uint32_t perTile = 1024u / ((uint32_t)(kRpBlock / 64) * (64u >> passesLog2)); // workgroups per tile (a power of two)
        const uint32_t x = b & 7u, i = b >> 3;
        b = ((i / perTile) * 8u + x) * perTile + (i % perTile);
    }
    const uint32_t m = (b * (uint32_t)(kRpBlock / 64) + wave) * npx + (lane >> passesLog2); // owned pixel, patches in Morton order
    const SegDev &seg = tbl->seg[segs.seg[pass]];
    hr_pass_params pp = UNIFORM ? tbl->seg[segs.seg[0]].pp : seg.pp;
    if (UNIFORM) pp.sample_index = seg.pp.sample_index;
    int x = 0, y = 0;
    const bool inFrame = ownedPixel(fr, (m & ~63u) + mortonPixel(m & 63u), x, y);
    CameraLane c;
    cameraLane(S, pp, seg, fr, inFrame, x, y, c);
    const Ray &r = c.r;
    const uint32_t pixel = c.pixel;
    const bool active = c.active, enqueue = c.enqueue;
    uint32_t nAcc = c.nAcc;
    __syncthreads();
    const uint32_t rank = enqueue ? atomicAdd(&cnt[pass], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < nPass) firstSlot[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(tbl->seg[segs.seg[threadIdx.x]].qCountIn, cnt[threadIdx.x]) : 0u;
    __syncthreads();
    const uint32_t slot = firstSlot[pass] + rank;
    const bool fits = slot < closestCap(seg);
    if (enqueue && fits) storeRay(seg.qin, slot, r, pixel, 0xFFFFFFFFu);
    if (enqueue && !fits) queueOverflow(tbl, OVF_CAMERA, (uint32_t)segs.seg[pass], slot + 1u);
    const unsigned long long enqMask = __ballot(enqueue);
    const uint32_t n = waveSum(active ? 1u : 0u);
    nAcc = waveSum(nAcc);
    if (lane == 0) {
        if (n) atomicAdd(&stats->paths, (unsigned long long)n);
        if (n) atomicAdd(&stats->raysClosest, (unsigned long long)n); // (the culled ones were traced by the root test, the others are traced below)
        if (nAcc) atomicAdd(&stats->accumulates, (unsigned long long)nAcc);
    }
    if (enqMask == 0ull) return;
    HitRec best;
    uint32_t nv = 0, nt = 0, entered = 0, own = 0;
    packetTraverse<STATS, false>(S, (ConstNodes)(uintptr_t)nodesG, (ConstTris)(uintptr_t)trisG, enqueue ? r.o : v3(0.0f), enqueue ? r.d : v3(0.0f, 0.0f, 1.0f),
                                 enqueue ? r.maxT : 0.0f, 0xFFFFFFFFu, best, nv, nt, entered, own);
    if (enqueue && fits) G(seg.hits)[slot] = best;
    if (STATS) { // per ray: the nodes and triangles its OWN box tests reached (what the ray costs traced alone; the packet's union is the probe's business)
        nv = waveSum(enqueue ? nv : 0u), nt = waveSum(enqueue ? nt : 0u);
        if (lane == 0) atomicAdd(&stats->nodeVisits, (unsigned long long)nv), atomicAdd(&stats->triTests, (unsigned long long)nt);
    }
}

// The selector's probe (hr_core.hip): a launch of its own on a side stream that depends on nothing the pipeline writes.  It generates
// the camera rays of every kProbeStride-th 8x8 patch of one pass itself, walks them as packets and writes nothing but three totals:
// probe[0] += children the packet entered x its rays, probe[1] += children the rays' own box tests entered, probe[2] += 1 per wave,
// probe[3] += rays walked.
static const int kProbeStride = 32;
__global__ __launch_bounds__(kPacketBlock) void k_packet_probe(const SceneDev *__restrict__ Sp, const Node4 *__restrict__ nodesG, const Tri *__restrict__ trisG,
                                                              hr_pass_params pp, int passesLog2, FrameDev fr, unsigned long long *probe)
{
    const SceneDev &S = *Sp;
    int x = 0, y = 0;
    // the wave's 64 rays: 2^passesLog2 consecutive passes of 64 >> passesLog2 neighbouring pixels (0: one pass of an 8x8 patch)
    const uint32_t lane = threadIdx.x, pass = lane & ((1u << passesLog2) - 1u), pix = lane >> passesLog2, npx = 64u >> passesLog2;
    const uint32_t group = blockIdx.x & ((1u << passesLog2) - 1u); // which of the patch's pixel groups this wave samples
    pp.sample_index += (int)pass;
    const bool inFrame = ownedPixel(fr, blockIdx.x * 64u * (uint32_t)kProbeStride + mortonPixel(group * npx + pix), x, y);
    Ray r;
    r.valid = false;
    bool active = inFrame;
    if (active) active = generatePrimary(S, pp, fr.W, fr.H, x, y, r);
    if (active && S.nTris > 0 && S.rootLeafCount == 0 && rootMissed(S.nodes, r.o, r.d, S.rayEps, r.maxT)) active = false; // (k_raygen's cull)
    uint32_t entered = 0, own = 0;
    if (__ballot(active)) {
        HitRec best;
        uint32_t nv = 0, nt = 0;
        packetTraverse<false, true>(S, (ConstNodes)(uintptr_t)nodesG, (ConstTris)(uintptr_t)trisG, active ? r.o : v3(0.0f), active ? r.d : v3(0.0f, 0.0f, 1.0f),
                                    active ? r.maxT : 0.0f, 0xFFFFFFFFu, best, nv, nt, entered, own);
    }
    const uint32_t nRays = (uint32_t)__popcll(__ballot(active));
    own = waveSum(active ? own : 0u);
    if (laneId() == 0) {
        if (nRays) atomicAdd(&probe[0], (unsigned long long)entered * nRays), atomicAdd(&probe[1], (unsigned long long)own), atomicAdd(&probe[3], (unsigned long long)nRays);
        __threadfence();
        atomicAdd(&probe[2], 1ull);
    }
}

// MEASUREMENT ONLY (HR_TUNE sprobe=1|2; VERDICT r4 item 6: "packets beyond the camera"): how coherent are the OCCLUSION rays of a queue, taken 64
// consecutive entries at a time as they lie there?  Same four totals as k_packet_probe, for the occlusion queues of the table entries listed
// (closest-hit semantics up to each ray's own tmax: an any-hit packet could end earlier, so the packet's share is an upper bound).
__global__ __launch_bounds__(kPacketBlock) void k_shadow_probe(const SceneDev *__restrict__ Sp, const Node4 *__restrict__ nodesG, const Tri *__restrict__ trisG,
                                                              const StepTable *__restrict__ tbl, SegList segs, unsigned long long *probe)
{
    const SceneDev &S = *Sp;
    const SegDev &sg = tbl->seg[segs.seg[blockIdx.y]];
    const uint32_t n = occlusionCount(sg), i = blockIdx.x * 64u + threadIdx.x;
    if (blockIdx.x * 64u >= n) return;
    const bool active = i < n;
    float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 1, 0);
    if (active) a = G(sg.sqIn.A)[i], b = G(sg.sqIn.B)[i];
    HitRec best;
    uint32_t nv = 0, nt = 0, entered = 0, own = 0;
    packetTraverse<false, true>(S, (ConstNodes)(uintptr_t)nodesG, (ConstTris)(uintptr_t)trisG, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), active ? a.w : 0.0f,
                                active ? __float_as_uint(b.w) : 0xFFFFFFFFu, best, nv, nt, entered, own);
    const uint32_t nRays = (uint32_t)__popcll(__ballot(active));
    own = waveSum(active ? own : 0u);
    if (laneId() == 0 && nRays)
        atomicAdd(&probe[0], (unsigned long long)entered * nRays), atomicAdd(&probe[1], (unsigned long long)own), atomicAdd(&probe[2], 1ull), atomicAdd(&probe[3], (unsigned long long)nRays);
}
void launchShadowProbe(hipStream_t stream, const SceneDev *S, const Node4 *nodes, const Tri *tris, const StepTable *tbl, const SegList &segs, uint32_t maxRays, unsigned long long *probe)
{
    if (segs.n <= 0 || maxRays == 0) return;
    hipLaunchKernelGGL(k_shadow_probe, dim3((maxRays + 63u) / 64u, segs.n), dim3(kPacketBlock), 0, stream, S, nodes, tris, tbl, segs, probe);
}

// ------------------------------------------------------------------------------------------- shade
#ifndef HR_SHADE_BLOCK
#define HR_SHADE_BLOCK 256
#endif
static const int kShadeBlock = HR_SHADE_BLOCK; // queue slots are reserved once per workgroup and pass (fewer same-address atomics)
// ------------------------------------------------------------------------- shade, split by shader
// Since round 3 the shading stage is three kernels (the fused single kernel of rounds 1-2: profiles/experiments/).
//
//   k_shade_sort            one lane per closest-hit ray: a ray that hit nothing runs its defaultPrimitive's shader right here (the
//                           environment lookup; nothing at all for rl_NullPrimitive); a ray that hit something is appended to its
//                           pass's hit list — PBR hits from the front, glass hits from the back.  52 VGPRs, bandwidth-bound.
//   k_shade_hit<MODE, 0>    physicallyBased.rlsl on the PBR hit lists: every lane of every wave runs the material shader
//   k_shade_hit<MODE, 1>    glass.rlsl on the glass hit lists (launched only when the scene has a glass material)
//
// In the single kernel a wave carried the hit fraction of its 64 rays (~26 % on c3) through the material code — regrouping inside a
// 256-ray workgroup raised that to 38 % — and every instantiation paid the register budget of PBR + glass + miss together
// (128 VGPRs + 120 B of scratch).  Which lane shades which ray never matters: every ray carries its pixel, and a pixel has at most
// one closest-hit ray per pass and stage.
template <class F> HRD void buildStarts(uint32_t *start /* LDS, kMaxSegs + 1 */, int n, F countOf)
{
    for (int k = threadIdx.x; k < n; k += blockDim.x) start[k] = countOf(k);
    __syncthreads();
    if (threadIdx.x < 64) {
        constexpr int kPer = (kMaxSegs + 1 + 63) / 64;
        const int first = (int)threadIdx.x * kPer;
        uint32_t v[kPer];
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            v[j] = (first + j < n) ? start[first + j] : 0u;
            sum += v[j];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        uint32_t acc = incl - sum;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (first + j <= n) start[first + j] = acc;
            acc += v[j];
        }
    }
    __syncthreads();
}
HRD int findStart(const uint32_t *start, int n, uint32_t item) // last entry whose first index is <= item
{
    int lo = 0, hiB = n - 1;
    while (lo < hiB) {
        const int mid = (lo + hiB + 1) >> 1;
        if (item >= start[mid])
            lo = mid;
        else
            hiB = mid - 1;
    }
    return lo;
}

static const int kSortBlock = 256;
__global__ __launch_bounds__(kSortBlock) void k_shade_sort(const SceneDev *__restrict__ Sp, const StepTable *__restrict__ tbl, Stats *stats)
{
    __shared__ uint32_t start[kMaxSegs + 1];
    __shared__ uint32_t scratch[2 + kSortBlock / 64];
    const SceneDev &S = *Sp;
    stats += blockIdx.x & (kStatSlots - 1);
    const int nSeg = tbl->nSeg;
    if (blockIdx.x == 0 && threadIdx.x == 0) { // duration of the k_trace launch of this step, by its own clock readings (hr_kernels.h: clkStart)
        unsigned long long lo = ~0ull, hi = 0ull;
        for (int k = 0; k < kClkSlots; ++k) {
            const unsigned long long a = tbl->clkStart[k], b = tbl->clkEnd[k];
            lo = a < lo ? a : lo, hi = b > hi ? b : hi;
        }
        if (hi > lo) atomicAdd(&stats->traceTicks, hi - lo);
        const unsigned long long idx = atomicAdd(&stats->traceLaunches, 1ull); // (this thread always adds to the first copy of the counters)
        if (tbl->stepLog) {
            unsigned long long *rec = tbl->stepLog + 3ull * (idx % (unsigned long long)kStepLogCap);
            rec[0] = lo, rec[1] = hi, rec[2] = (unsigned long long)(uint32_t)nSeg | ((unsigned long long)(tbl->group & 0xFFu) << 16) | ((unsigned long long)tbl->nInjectedNow << 32);
        }
        if (tbl->hostCameraCount) { // (the packet kernel ran beside k_trace, whose report could only give these queues' capacity)
            unsigned long long sum = 0, cnt = 0;
            for (int k = tbl->primaryFromSeg; k < nSeg; ++k)
                if (tbl->seg[k].packets == 2u) sum += closestCount(tbl->seg[k]), ++cnt;
            if (cnt) __hip_atomic_store(tbl->hostCameraCount, (uint32_t)(sum / cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    buildStarts(start, nSeg, [&](int k) { return tbl->seg[k].closestEnabled ? closestCount(tbl->seg[k]) : 0u; });
    const uint32_t total = start[nSeg];
    const bool glassToo = tbl->hasGlass != 0;
    uint32_t nAccum = 0;
    for (uint32_t base = blockIdx.x * kSortBlock; base < total; base += gridDim.x * kSortBlock) {
        const uint32_t i = base + threadIdx.x;
        const bool live = i < total;
        const int sI = live ? findStart(start, nSeg, i) : 0;
        const uint32_t li = live ? i - start[sI] : 0u;
        int cls = -1; // 0: PBR hit, 1: glass hit
        if (live) {
            const SegDev &sg = tbl->seg[sI];
            uint32_t hp = G(sg.hits)[li].prim;
            if (hp == kMissPrim) {
                // a ray that hits nothing runs its defaultPrimitive's shader (none for rl_NullPrimitive)
                const uint32_t meta = (uint32_t)G(sg.qin.D)[li].x;
                if (((meta >> 24) & 7u) == (uint32_t)MISS_ENV) {
                    const float4 b = G(sg.qin.B)[li], c = G(sg.qin.C)[li];
                    ShaderT<0> sh(S, sg.pp, G(sg.passbuf) + (size_t)__float_as_uint(c.w) * 4);
                    sh.performAccumulate(sh.environmentRadiance(v3(b.x, b.y, b.z), v3(c.x, c.y, c.z)));
                    nAccum += sh.nAccum;
                }
            } else {
                const uint32_t mid = G(S.attrs)[hp & 0x7FFFFFFFu].matflags & kMatMask;
                if (mid < (uint32_t)S.nMaterials) {
                    const int32_t type = G(S.materials)[mid].type;
                    cls = type == HR_MAT_PBR ? 0 : (type == HR_MAT_GLASS ? 1 : -1);
                }
            }
        }
        const uint32_t last = (total - base < (uint32_t)kSortBlock ? total - base : (uint32_t)kSortBlock) - 1u;
        const int sLo = findStart(start, nSeg, base), sHi = findStart(start, nSeg, base + last); // (uniform over the workgroup)
        if (sLo == sHi) { // the usual case: one reservation per workgroup and list
            const SegDev &sg = tbl->seg[sLo];
            // (a list entry is one of the queue's rays, and there are at most hitCap of those: the slots cannot run out — checked all the same)
            const uint32_t pSlot = blockReserve(cls == 0, sg.pCount, scratch);
            if (cls == 0 && pSlot < sg.hitCap) G(sg.hitIdx)[pSlot] = li;
            if (cls == 0 && pSlot >= sg.hitCap) queueOverflow(tbl, OVF_HIT_LIST, (uint32_t)sLo, pSlot + 1u);
            if (glassToo) {
                const uint32_t gSlot = blockReserve(cls == 1, sg.gCount, scratch);
                if (cls == 1 && gSlot < sg.hitCap) G(sg.hitIdx)[sg.hitCap - 1u - gSlot] = li;
                if (cls == 1 && gSlot >= sg.hitCap) queueOverflow(tbl, OVF_HIT_LIST, (uint32_t)sLo, gSlot + 1u);
            }
        } else { // a batch that straddles passes: one reservation per wave, list and pass present in the wave
            unsigned long long todo = __ballot(cls >= 0);
            while (todo != 0ull) {
                const int s = __shfl(sI, __ffsll((long long)todo) - 1);
                const bool mine = cls >= 0 && sI == s;
                todo &= ~__ballot(mine);
                const SegDev &sg = tbl->seg[s];
                const uint32_t pSlot = waveReserve(mine && cls == 0, sg.pCount);
                if (mine && cls == 0 && pSlot < sg.hitCap) G(sg.hitIdx)[pSlot] = li;
                if (mine && cls == 0 && pSlot >= sg.hitCap) queueOverflow(tbl, OVF_HIT_LIST, (uint32_t)s, pSlot + 1u);
                const uint32_t gSlot = waveReserve(mine && cls == 1, sg.gCount);
                if (mine && cls == 1 && gSlot < sg.hitCap) G(sg.hitIdx)[sg.hitCap - 1u - gSlot] = li;
                if (mine && cls == 1 && gSlot >= sg.hitCap) queueOverflow(tbl, OVF_HIT_LIST, (uint32_t)s, gSlot + 1u);
            }
        }
    }
    nAccum = waveSum(nAccum);
    if (laneId() == 0 && nAccum) atomicAdd(&stats->accumulates, (unsigned long long)nAccum);
}

#ifndef HR_HIT_MINBLOCKS
#define HR_HIT_MINBLOCKS 4
#endif
// CLS 0: the PBR hit lists, CLS 1: the glass hit lists
template <int MODE, int CLS>
__global__ __launch_bounds__(kShadeBlock, HR_HIT_MINBLOCKS) void k_shade_hit(const SceneDev *__restrict__ Sp, const StepTable *__restrict__ tbl, Stats *stats)
{
    __shared__ uint32_t start[kMaxSegs + 1];
    __shared__ uint32_t scratch[2 + kShadeBlock / 64];
    const SceneDev &S = *Sp;
    stats += blockIdx.x & (kStatSlots - 1);
    const int nSeg = tbl->nSeg;
    buildStarts(start, nSeg, [&](int k) {
        const uint32_t listed = tbl->seg[k].closestEnabled ? *(CLS == 0 ? tbl->seg[k].pCount : tbl->seg[k].gCount) : 0u;
        return listed < tbl->seg[k].hitCap ? listed : tbl->seg[k].hitCap;
    });
    const uint32_t total = start[nSeg];
    constexpr bool LOD = (MODE & 1) != 0, ALL = (MODE & 2) != 0;
    uint32_t nShaded = 0, nAccum = 0;
    for (uint32_t base = blockIdx.x * kShadeBlock; base < total; base += gridDim.x * kShadeBlock) {
        const uint32_t i = base + threadIdx.x;
        const bool live = i < total;
        const int sI = live ? findStart(start, nSeg, i) : 0;
        Ray next;
        ExtraRay nee, extra[3]; // (occlusion rays in their compact form: direction, range, the value their light's shader adds; they start at the hit point)
        nee.valid = next.valid = extra[0].valid = extra[1].valid = extra[2].valid = false;
        uint32_t pixel = 0, prim = 0xFFFFFFFFu;
        v3 hitP(0.0f);
        if (live) {
            const SegDev &sg = tbl->seg[sI];
            const uint32_t j = i - start[sI];
            const uint32_t li = G(sg.hitIdx)[CLS == 0 ? j : sg.hitCap - 1u - j];
            const float4 a = G(sg.qin.A)[li], b = G(sg.qin.B)[li], c = G(sg.qin.C)[li];
            const int4 dm = G(sg.qin.D)[li];
            const HitRec h = G(sg.hits)[li];
            Ray in;
            in.o = v3(a.x, a.y, a.z), in.d = v3(b.x, b.y, b.z), in.maxT = a.w, in.extraT = b.w;
            in.weight = v3(c.x, c.y, c.z);
            pixel = __float_as_uint(c.w);
            const uint32_t meta = (uint32_t)dm.x;
            in.sequenceID = (int)(meta & 0xFFu), in.depth = (int)((meta >> 8) & 0xFFFFu);
            in.missKind = (int)((meta >> 24) & 7u), in.missIdx = (int)((meta >> 27) & 7u);
            in.sequenceIndexOffset = dm.y;
            in.occlusionTest = false, in.valid = true;
            in.coneW = in.coneG = 0.0f;
            if (LOD) unpackCone((uint32_t)dm.w, in.coneW, in.coneG);
            ShaderT<MODE> sh(S, sg.pp, G(sg.passbuf) + (size_t)pixel * 4);
            prim = h.prim & 0x7FFFFFFFu;
            uint32_t mid;
            const typename ShaderT<MODE>::Surface sf = sh.surface(in, prim, (h.prim >> 31) != 0u, h.t, h.u, h.v, mid);
            sh.setFootprint(in, sf.normal, h.t, prim);
            const HR_GLOBAL hr_material &M = G(S.materials)[mid]; // (k_shade_sort listed the ray because mid is a material of this class)
            ++nShaded;
            if (CLS == 1)
                sh.glass(in, sf, h.t, M, nee, next, extra[0]);
            else
                sh.physicallyBased(in, sf, M, nee, next, extra[0], extra[1], extra[2]);
            hitP = sf.P; // (where the occlusion rays start)
            nAccum += sh.nAccum;
        }
        // the emitted rays leave through compacted appends to the pass's queues
        const uint32_t last = (total - base < (uint32_t)kShadeBlock ? total - base : (uint32_t)kShadeBlock) - 1u;
        const int sLo = findStart(start, nSeg, base), sHi = findStart(start, nSeg, base + last); // (uniform over the workgroup)
        if (sLo == sHi) {
            const SegDev &sg = tbl->seg[sLo];
            const bool wantS = live && nee.valid;
            const uint32_t sSlot = blockReserve(wantS, sg.sCountOut, scratch);
            if (wantS && sSlot >= sg.sOutCap) queueOverflow(tbl, OVF_OCCLUSION_OUT, (uint32_t)sLo, sSlot + 1u);
            if (wantS && sSlot < sg.sOutCap) {
                G(sg.sqOut.A)[sSlot] = make_float4(hitP.x, hitP.y, hitP.z, nee.maxT);
                G(sg.sqOut.B)[sSlot] = make_float4(nee.d.x, nee.d.y, nee.d.z, __uint_as_float(prim));
                G(sg.sqOut.C)[sSlot] = make_float4(nee.value.x, nee.value.y, nee.value.z, __uint_as_float(pixel));
            }
            if (ALL) {
                // each extra ray adds to a partial sum of its own (no two rays of a launch may write one pixel): partial sum j + 1 lies
                // (j + 1) frames behind the first in the pass's buffer — the trace kernel just sees a pixel index beyond the frame
                const uint32_t framePixels = (uint32_t)((sg.passbufB - sg.passbuf) >> 2);
#pragma unroll
                for (int j = 0; j < (CLS == 1 ? 1 : 3); ++j) {
                    const bool wantX = live && extra[j].valid;
                    const uint32_t sx = blockReserve(wantX, sg.sCountOut, scratch);
                    if (wantX && sx >= sg.sOutCap) queueOverflow(tbl, OVF_OCCLUSION_OUT, (uint32_t)sLo, sx + 1u);
                    if (wantX && sx < sg.sOutCap) {
                        G(sg.sqOut.A)[sx] = make_float4(hitP.x, hitP.y, hitP.z, extra[j].maxT);
                        G(sg.sqOut.B)[sx] = make_float4(extra[j].d.x, extra[j].d.y, extra[j].d.z, __uint_as_float(prim));
                        G(sg.sqOut.C)[sx] = make_float4(extra[j].value.x, extra[j].value.y, extra[j].value.z, __uint_as_float(pixel + (uint32_t)(j + 1) * framePixels));
                    }
                }
            }
            const bool wantQ = live && next.valid;
            const uint32_t qSlot = blockReserve(wantQ, sg.qCountOut, scratch);
            if (wantQ && qSlot < sg.hitCap) storeRay(sg.qout, qSlot, next, pixel, prim);
            if (wantQ && qSlot >= sg.hitCap) queueOverflow(tbl, OVF_CLOSEST_OUT, (uint32_t)sLo, qSlot + 1u);
        } else {
            unsigned long long todo = __ballot(live && (nee.valid || next.valid || (ALL && (extra[0].valid || extra[1].valid || extra[2].valid))));
            while (todo != 0ull) {
                const int s = __shfl(sI, __ffsll((long long)todo) - 1);
                const bool mine = live && sI == s;
                todo &= ~__ballot(mine);
                const SegDev &sg = tbl->seg[s];
                const bool wantS = mine && nee.valid;
                const uint32_t sSlot = waveReserve(wantS, sg.sCountOut);
                if (wantS && sSlot >= sg.sOutCap) queueOverflow(tbl, OVF_OCCLUSION_OUT, (uint32_t)s, sSlot + 1u);
                if (wantS && sSlot < sg.sOutCap) {
                    G(sg.sqOut.A)[sSlot] = make_float4(hitP.x, hitP.y, hitP.z, nee.maxT);
                    G(sg.sqOut.B)[sSlot] = make_float4(nee.d.x, nee.d.y, nee.d.z, __uint_as_float(prim));
                    G(sg.sqOut.C)[sSlot] = make_float4(nee.value.x, nee.value.y, nee.value.z, __uint_as_float(pixel));
                }
                if (ALL) {
                    const uint32_t framePixels = (uint32_t)((sg.passbufB - sg.passbuf) >> 2);
#pragma unroll
                    for (int j = 0; j < (CLS == 1 ? 1 : 3); ++j) {
                        const bool wantX = mine && extra[j].valid;
                        const uint32_t sx = waveReserve(wantX, sg.sCountOut);
                        if (wantX && sx >= sg.sOutCap) queueOverflow(tbl, OVF_OCCLUSION_OUT, (uint32_t)s, sx + 1u);
                        if (wantX && sx < sg.sOutCap) {
                            G(sg.sqOut.A)[sx] = make_float4(hitP.x, hitP.y, hitP.z, extra[j].maxT);
                            G(sg.sqOut.B)[sx] = make_float4(extra[j].d.x, extra[j].d.y, extra[j].d.z, __uint_as_float(prim));
                            G(sg.sqOut.C)[sx] = make_float4(extra[j].value.x, extra[j].value.y, extra[j].value.z, __uint_as_float(pixel + (uint32_t)(j + 1) * framePixels));
                        }
                    }
                }
                const bool wantQ = mine && next.valid;
                const uint32_t qSlot = waveReserve(wantQ, sg.qCountOut);
                if (wantQ && qSlot < sg.hitCap) storeRay(sg.qout, qSlot, next, pixel, prim);
                if (wantQ && qSlot >= sg.hitCap) queueOverflow(tbl, OVF_CLOSEST_OUT, (uint32_t)s, qSlot + 1u);
            }
        }
    }
    nShaded = waveSum(nShaded), nAccum = waveSum(nAccum);
    if (laneId() == 0) {
        if (nShaded) atomicAdd(&stats->shadedHits, (unsigned long long)nShaded);
        if (nAccum) atomicAdd(&stats->accumulates, (unsigned long long)nAccum);
    }
}

// ------------------------------------------------------------------------------------ debug trace
__global__ __launch_bounds__(kBlock) void k_debug_trace(const SceneDev *__restrict__ Sp, int n, const float *__restrict__ o,
                                                        const float *__restrict__ d, const float *__restrict__ tmax,
                                                        const int *__restrict__ skip, int anyHit, hr_hit *__restrict__ out)
{
    __shared__ int stack[kWavesPerBlock][kStackLDS][64];
    const SceneDev &S = *Sp;
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    int *stackLane = &stack[wave][0][lane];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const v3 ro(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    const float tm = tmax ? tmax[i] : __builtin_inff();
    const uint32_t sk = skip ? (uint32_t)skip[i] : 0xFFFFFFFFu;
    HitRec h;
    uint32_t nv = 0, nt = 0;
    hr_hit r;
    if (anyHit) {
        traverse<true, false>(S, ro, rd, S.rayEps, tm, sk, stackLane, h, nv, nt);
        r.prim = (h.prim == kMissPrim) ? -1 : 0;
        r.t = r.u = r.v = 0.0f;
    } else {
        traverse<false, false>(S, ro, rd, S.rayEps, tm, sk, stackLane, h, nv, nt);
        const bool hit = h.prim != kMissPrim;
        r.prim = hit ? (int)(h.prim & 0x7FFFFFFFu) : -1;
        r.t = hit ? h.t : 0.0f, r.u = hit ? h.u : 0.0f, r.v = hit ? h.v : 0.0f;
    }
    out[i] = r;
}

// ------------------------------------------------------------------------------------- launchers
static int ownedThreads(const FrameDev &fr) { return fr.nOwnedTiles * fr.tile * fr.tile; }

void launchRaygen(const LaunchCfg &cfg, const SceneDev *S, const StepTable *tbl, const SegList &segs, const FrameDev &fr, Stats *stats)
{
    const int threads = ownedThreads(fr);
    if (threads <= 0 || segs.n <= 0) return;
    hipLaunchKernelGGL(k_raygen, dim3((threads + kRaygenBlock - 1) / kRaygenBlock, segs.n), dim3(kRaygenBlock), 0, cfg.stream, S, tbl, segs, fr, stats);
}

void launchResolve(const LaunchCfg &cfg, const FrameDev &fr, const PassBufList &bufs)
{
    const int threads = ownedThreads(fr);
    if (threads <= 0 || bufs.n <= 0) return;
    hipLaunchKernelGGL(k_resolve, dim3((threads + kBlock - 1) / kBlock), dim3(kBlock), 0, cfg.stream, fr, bufs);
}

void launchTrace(const LaunchCfg &cfg, const SceneDev *S, const int *leafKeys, const Node32 *nodes32, const Tri *tris, StepTable *tbl, Stats *stats)
{
    const int grid = cfg.numCUs * cfg.traceBlocksPerCU * (kBlock / kTraceBlock); // traceBlocksPerCU counts 256-thread workgroups
    if (cfg.collectStats)
        hipLaunchKernelGGL(k_trace<true>, dim3(grid), dim3(kTraceBlock), 0, cfg.stream, S, leafKeys, nodes32, tris, tbl, stats);
    else
        hipLaunchKernelGGL(k_trace<false>, dim3(grid), dim3(kTraceBlock), 0, cfg.stream, S, leafKeys, nodes32, tris, tbl, stats);
}

void launchRaygenPackets(const LaunchCfg &cfg, const SceneDev *S, const Node4 *nodes, const Tri *tris, const StepTable *tbl, const SegList &segs,
                         const FrameDev &fr, Stats *stats, bool uniformParams)
{
    const int threads = ownedThreads(fr);
    if (threads <= 0 || segs.n <= 0) return;
    int passesLog2 = 0;
    while ((2 << passesLog2) <= segs.n) ++passesLog2; // (the caller passes a power of two)
    const int pixelsPerBlock = (kRpBlock / 64) * (64 >> passesLog2);
    dim3 grid((threads + pixelsPerBlock - 1) / pixelsPerBlock);
    if (cfg.packetSwizzle && fr.tile == 32) { // (whole groups of 8 tiles: the kernel's ownedPixel() turns away what lies beyond the frame)
        const uint32_t group = 8u * (1024u / (uint32_t)pixelsPerBlock);
        grid.x = (grid.x + group - 1u) / group * group;
        passesLog2 |= 256;
    }
    if (cfg.collectStats && uniformParams)
        hipLaunchKernelGGL((k_raygen_packets<true, true>), grid, dim3(kRpBlock), 0, cfg.stream, S, nodes, tris, tbl, segs, passesLog2, fr, stats);
    else if (cfg.collectStats)
        hipLaunchKernelGGL((k_raygen_packets<true, false>), grid, dim3(kRpBlock), 0, cfg.stream, S, nodes, tris, tbl, segs, passesLog2, fr, stats);
    else if (uniformParams)
        hipLaunchKernelGGL((k_raygen_packets<false, true>), grid, dim3(kRpBlock), 0, cfg.stream, S, nodes, tris, tbl, segs, passesLog2, fr, stats);
    else
        hipLaunchKernelGGL((k_raygen_packets<false, false>), grid, dim3(kRpBlock), 0, cfg.stream, S, nodes, tris, tbl, segs, passesLog2, fr, stats);
}

// returns the number of waves launched: each adds one to probe[2] when it is done
int launchPacketProbe(hipStream_t stream, const SceneDev *S, const Node4 *nodes, const Tri *tris, const hr_pass_params &pp, int passesLog2, const FrameDev &fr,
                      unsigned long long *probe)
{
    const int threads = ownedThreads(fr);
    if (threads <= 0) return 0;
    const int packets = (threads + kPacketBlock - 1) / kPacketBlock;
    const int grid = (packets + kProbeStride - 1) / kProbeStride;
    hipLaunchKernelGGL(k_packet_probe, dim3(grid), dim3(kPacketBlock), 0, stream, S, nodes, tris, pp, passesLog2, fr, probe);
    return grid;
}

template <int MODE> static void launchShadeHit(const LaunchCfg &cfg, int grid, const SceneDev *S, const StepTable *tbl, Stats *stats)
{
    hipLaunchKernelGGL((k_shade_hit<MODE, 0>), dim3(grid), dim3(kShadeBlock), 0, cfg.stream, S, tbl, stats);
    if (cfg.hasGlass) hipLaunchKernelGGL((k_shade_hit<MODE, 1>), dim3(grid), dim3(kShadeBlock), 0, cfg.stream, S, tbl, stats);
}

void launchShade(const LaunchCfg &cfg, const SceneDev *S, const StepTable *tbl, Stats *stats)
{
    const int grid = cfg.numCUs * cfg.shadeBlocksPerCU;
    // four instantiations: bit 0 = HR_TEXTURE_LOD_CONE, bit 1 = HR_ESTIMATOR_ALL_LIGHTS compiled in; the plain one runs until a pass asks for more
    const int mode = (cfg.textureLod ? 1 : 0) | (cfg.allLights ? 2 : 0);
    hipLaunchKernelGGL(k_shade_sort, dim3(cfg.numCUs * 8), dim3(kSortBlock), 0, cfg.stream, S, tbl, stats);
    switch (mode) {
    case 0: launchShadeHit<0>(cfg, grid, S, tbl, stats); break;
    case 1: launchShadeHit<1>(cfg, grid, S, tbl, stats); break;
    case 2: launchShadeHit<2>(cfg, grid, S, tbl, stats); break;
    default: launchShadeHit<3>(cfg, grid, S, tbl, stats); break;
    }
}

void launchDisplay(const LaunchCfg &cfg, const FrameDev &fr, const hr_display_params &P, int format, void *out)
{
    const int n = fr.W * fr.H;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_display, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cfg.stream, fr, P, format, out);
}

void launchDebugTrace(const LaunchCfg &cfg, const SceneDev *S, int n, const float *o, const float *d, const float *tmax, const int *skip,
                      int anyHit, hr_hit *out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_debug_trace, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cfg.stream, S, n, o, d, tmax, skip, anyHit, out);
}

size_t hitRecordSize() { return sizeof(HitRec); }

} // namespace hr

#ifdef HR_TAILPROF
extern "C" int hr_debug_tailprof(unsigned long long *out8, int reset)
{
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hr::g_tailprof), sizeof(unsigned long long) * 24) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {~0ull, ~0ull, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(hr::g_tailprof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
