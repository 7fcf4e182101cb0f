// hr_render.hip — the per-pass wavefront loop: primary-ray generation, persistent-threads BVH
// traversal (closest hit + occlusion), SoA shading with wave-level queue compaction, accumulation.
//
// Replaces rlRenderFrame() (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386) and the RLSL
// programs it runs (Resources/shaders/perspective.rlsl, physicallyBased.rlsl, glass.rlsl, *Light.rlsl,
// accumulator.rlsl).  Kernel sequence of one pass (all on one stream):
//
//   raygen -> { trace(i): closest hits of queue i  +  occlusion rays emitted by shade(i-1)
//               shade(i): materials / miss shaders, emits queue i+1 and occlusion queue i } x (depth+1)
//          -> trace of the last occlusion queue
//
// Every pixel has at most one live path and one live occlusion ray, and kernels are stream-ordered, so
// the accumulation buffer is updated with plain read-modify-write in a fixed per-pixel order
// (A+=1, then per bounce: emissive, NEE light, environment) — reproducible bit for bit.
#include "hr_kernels.h"
#include "hr_shade.h"
#include "hr_trace.h"

namespace hr {

static const int kBlock = 256;
static const int kWavesPerBlock = kBlock / 64;

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

HRD uint32_t waveSum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

HRD uint32_t packMeta(const Ray &r)
{
    return (uint32_t)(r.sequenceID & 0xFF) | ((uint32_t)(r.depth & 0xFFFF) << 8) | ((uint32_t)r.missKind << 24) | ((uint32_t)r.missIdx << 27);
}

HRD void storeRay(const RayQueue &q, uint32_t slot, const Ray &r, uint32_t pixel, uint32_t srcPrim)
{
    q.A[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.maxT);
    q.B[slot] = make_float4(r.d.x, r.d.y, r.d.z, r.extraT);
    q.C[slot] = make_float4(r.weight.x, r.weight.y, r.weight.z, __uint_as_float(pixel));
    q.D[slot] = make_int4((int)packMeta(r), r.sequenceIndexOffset, (int)srcPrim, 0);
}

// ------------------------------------------------------------------------------------------ raygen
__global__ __launch_bounds__(kBlock) void k_raygen(const SceneDev *__restrict__ Sp, hr_pass_params pp, FrameDev fr, RayQueue q,
                                                   Counters *ctr, Stats *stats)
{
    const SceneDev &S = *Sp;
    const uint32_t gid = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t perTile = (uint32_t)(fr.tile * fr.tile);
    const uint32_t tileSlot = gid / perTile, within = gid % perTile;
    bool active = tileSlot < (uint32_t)fr.nOwnedTiles;
    int x = 0, y = 0;
    if (active) {
        const int tileId = fr.rank + (int)tileSlot * fr.world;
        const int tx = tileId % fr.tilesX, ty = tileId / fr.tilesX;
        // 8x8-pixel blocks inside the tile so that one wave covers a compact screen patch
        const int blk = (int)(within >> 6), l = (int)(within & 63u), bpr = fr.tile >> 3;
        x = tx * fr.tile + (blk % bpr) * 8 + (l & 7);
        y = ty * fr.tile + (blk / bpr) * 8 + (l >> 3);
        active = x < fr.W && y < fr.H;
    }
    Ray r;
    r.valid = false;
    if (active) active = generatePrimary(S, pp, fr.W, fr.H, x, y, r);
    const uint32_t pixel = (uint32_t)(y * fr.W + x);
    if (active) fr.fb[(size_t)pixel * 4 + 3] += 1.0f; // perspective.rlsl:60 accumulate(vec4(0,0,0,1))
    const uint32_t slot = waveReserve(active, &ctr->qCount[0]);
    if (active) storeRay(q, slot, r, pixel, 0xFFFFFFFFu);
    const uint32_t n = waveSum(active ? 1u : 0u);
    if (laneId() == 0 && n) atomicAdd(&stats->paths, (unsigned long long)n);
}

// ------------------------------------------------------------------------------- closest-hit trace
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_trace_closest(const SceneDev *__restrict__ Sp, RayQueue q, HitRec *__restrict__ hits,
                                                          Counters *ctr, Stats *stats, int slot)
{
    __shared__ int stack[kWavesPerBlock][kStackLDS][64];
    const SceneDev &S = *Sp;
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    int *stackLane = &stack[wave][0][lane];
    const uint32_t count = ctr->qCount[slot];
    uint32_t nv = 0, nt = 0;
    // persistent threads: each wave pulls 64-ray batches until the queue is drained
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctr->qHead[slot], 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        const uint32_t i = base + lane;
        if (i < count) {
            const float4 a = q.A[i], b = q.B[i];
            const uint32_t src = (uint32_t)q.D[i].z;
            HitRec h;
            traverse<false, STATS>(S, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), S.rayEps, a.w, src, stackLane, h, nv, nt);
            hits[i] = h;
        }
    }
    if (STATS) {
        nv = waveSum(nv), nt = waveSum(nt);
        if (lane == 0) {
            atomicAdd(&stats->nodeVisits, (unsigned long long)nv);
            atomicAdd(&stats->triTests, (unsigned long long)nt);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->raysClosest, (unsigned long long)count);
}

// --------------------------------------------------------------------------------- occlusion trace
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_trace_shadow(const SceneDev *__restrict__ Sp, ShadowQueue sq, float *__restrict__ fb, Counters *ctr,
                                                         Stats *stats, int slot)
{
    __shared__ int stack[kWavesPerBlock][kStackLDS][64];
    const SceneDev &S = *Sp;
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    int *stackLane = &stack[wave][0][lane];
    const uint32_t count = ctr->sCount[slot];
    uint32_t nv = 0, nt = 0, nacc = 0;
    while (true) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctr->sHead[slot], 64u);
        base = __shfl(base, 0);
        if (base >= count) break;
        const uint32_t i = base + lane;
        if (i < count) {
            const float4 a = sq.A[i], b = sq.B[i];
            HitRec h;
            traverse<true, STATS>(S, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), S.rayEps, a.w, __float_as_uint(b.w), stackLane, h, nv, nt);
            if (h.prim == kMissPrim) { // unoccluded: the light's shader accumulates (single owner per pixel)
                const float4 c = sq.C[i];
                float *px = fb + (size_t)__float_as_uint(c.w) * 4;
                px[0] = px[0] + c.x;
                px[1] = px[1] + c.y;
                px[2] = px[2] + c.z;
                ++nacc;
            }
        }
    }
    nacc = waveSum(nacc);
    if (lane == 0 && nacc) atomicAdd(&stats->accumulates, (unsigned long long)nacc);
    if (STATS) {
        nv = waveSum(nv), nt = waveSum(nt);
        if (lane == 0) {
            atomicAdd(&stats->nodeVisits, (unsigned long long)nv);
            atomicAdd(&stats->triTests, (unsigned long long)nt);
            atomicAdd(&stats->nodeVisitsAny, (unsigned long long)nv);
            atomicAdd(&stats->triTestsAny, (unsigned long long)nt);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats->raysAny, (unsigned long long)count);
}

// ------------------------------------------------------------------------------------------- shade
__global__ __launch_bounds__(kBlock) void k_shade(const SceneDev *__restrict__ Sp, hr_pass_params pp, float *__restrict__ fb, RayQueue qin,
                                                  const HitRec *__restrict__ hits, RayQueue qout, ShadowQueue sq, Counters *ctr, Stats *stats,
                                                  int slot)
{
    const SceneDev &S = *Sp;
    const uint32_t count = ctr->qCount[slot];
    const uint32_t lane = laneId();
    uint32_t nShaded = 0, nAccum = 0;
    // wave-uniform trip count: every lane of a wave takes part in the compaction ballots
    for (uint32_t base = (blockIdx.x * kBlock + (threadIdx.x & ~63u)); base < count; base += gridDim.x * kBlock) {
        const uint32_t i = base + lane;
        const bool live = i < count;
        Ray nee, next;
        nee.valid = next.valid = false;
        uint32_t pixel = 0, prim = 0xFFFFFFFFu;
        v3 neeValue(0.0f);
        if (live) {
            const float4 a = qin.A[i], b = qin.B[i], c = qin.C[i];
            const int4 dm = qin.D[i];
            const HitRec h = hits[i];
            Ray in;
            in.o = v3(a.x, a.y, a.z), in.d = v3(b.x, b.y, b.z), in.maxT = a.w, in.extraT = b.w;
            in.weight = v3(c.x, c.y, c.z);
            pixel = __float_as_uint(c.w);
            const uint32_t meta = (uint32_t)dm.x;
            in.sequenceID = (int)(meta & 0xFFu), in.depth = (int)((meta >> 8) & 0xFFFFu);
            in.missKind = (int)((meta >> 24) & 7u), in.missIdx = (int)((meta >> 27) & 7u);
            in.sequenceIndexOffset = dm.y;
            in.occlusionTest = false, in.valid = true;
            Shader sh(S, pp, fb + (size_t)pixel * 4);
            if (h.prim == kMissPrim) {
                // a ray that hits nothing runs its defaultPrimitive's shader (none for rl_NullPrimitive)
                if (in.missKind == MISS_ENV) sh.performAccumulate(sh.environmentRadiance(in.d, in.weight));
            } else {
                prim = h.prim & 0x7FFFFFFFu;
                uint32_t mid;
                const Shader::Surface sf = sh.surface(in, prim, (h.prim >> 31) != 0u, h.t, h.u, h.v, mid);
                if (mid < (uint32_t)S.nMaterials) {
                    const hr_material &M = S.materials[mid];
                    if (M.type == HR_MAT_GLASS) {
                        ++nShaded;
                        sh.glass(in, sf, h.t, M, nee, next);
                    } else if (M.type == HR_MAT_PBR) {
                        ++nShaded;
                        sh.physicallyBased(in, sf, M, nee, next);
                    }
                }
                if (nee.valid) nee.valid = sh.lightShaderValue(nee, neeValue);
            }
            nAccum += sh.nAccum;
        }
        const uint32_t sSlot = waveReserve(nee.valid, &ctr->sCount[slot]);
        if (nee.valid) {
            sq.A[sSlot] = make_float4(nee.o.x, nee.o.y, nee.o.z, nee.maxT);
            sq.B[sSlot] = make_float4(nee.d.x, nee.d.y, nee.d.z, __uint_as_float(prim));
            sq.C[sSlot] = make_float4(neeValue.x, neeValue.y, neeValue.z, __uint_as_float(pixel));
        }
        const uint32_t qSlot = waveReserve(next.valid, &ctr->qCount[slot + 1]);
        if (next.valid) storeRay(qout, qSlot, next, pixel, prim);
    }
    nShaded = waveSum(nShaded), nAccum = waveSum(nAccum);
    if (lane == 0) {
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
static int gridFor(const LaunchCfg &cfg, int blocksPerCU) { return cfg.numCUs * blocksPerCU; }

void launchRaygen(const LaunchCfg &cfg, const SceneDev *S, const hr_pass_params &pp, const FrameDev &fr, RayQueue q, Counters *ctr, Stats *stats)
{
    const long long threads = (long long)fr.nOwnedTiles * fr.tile * fr.tile;
    if (threads <= 0) return;
    const int blocks = (int)((threads + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(kBlock), 0, cfg.stream, S, pp, fr, q, ctr, stats);
}

void launchTraceClosest(const LaunchCfg &cfg, const SceneDev *S, RayQueue q, void *hits, Counters *ctr, Stats *stats, int slot)
{
    const int grid = gridFor(cfg, cfg.traceBlocksPerCU);
    if (cfg.collectStats)
        hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(kBlock), 0, cfg.stream, S, q, (HitRec *)hits, ctr, stats, slot);
    else
        hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(kBlock), 0, cfg.stream, S, q, (HitRec *)hits, ctr, stats, slot);
}

void launchTraceShadow(const LaunchCfg &cfg, const SceneDev *S, ShadowQueue sq, float *fb, Counters *ctr, Stats *stats, int slot)
{
    const int grid = gridFor(cfg, cfg.traceBlocksPerCU);
    if (cfg.collectStats)
        hipLaunchKernelGGL(k_trace_shadow<true>, dim3(grid), dim3(kBlock), 0, cfg.stream, S, sq, fb, ctr, stats, slot);
    else
        hipLaunchKernelGGL(k_trace_shadow<false>, dim3(grid), dim3(kBlock), 0, cfg.stream, S, sq, fb, ctr, stats, slot);
}

void launchShade(const LaunchCfg &cfg, const SceneDev *S, const hr_pass_params &pp, float *fb, RayQueue qin, const void *hits, RayQueue qout,
                 ShadowQueue sq, Counters *ctr, Stats *stats, int slot)
{
    const int grid = gridFor(cfg, cfg.shadeBlocksPerCU);
    hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, cfg.stream, S, pp, fb, qin, (const HitRec *)hits, qout, sq, ctr, stats, slot);
}

void launchDebugTrace(const LaunchCfg &cfg, const SceneDev *S, int n, const float *o, const float *d, const float *tmax, const int *skip,
                      int anyHit, hr_hit *out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_debug_trace, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cfg.stream, S, n, o, d, tmax, skip, anyHit, out);
}

size_t hitRecordSize() { return sizeof(HitRec); }

} // namespace hr
