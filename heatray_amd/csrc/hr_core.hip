// hr_core.hip — host side of libhrcore: the C-ABI of include/hrcore.h over the HIP kernels.
//
// Owns device memory (scene, BVH, tables, ray queues, accumulation buffer) and sequences the kernels of
// a pass.  There is no CPU rendering path in this library: without a usable HIP device every entry point
// that needs one fails with HR_ERR_DEVICE.
//
//   hr_ctx.h          the context (what the opaque handle points to), the error macros
//   hr_core.hip       this file: context life cycle, frame, display, read-backs, statistics
//   hr_scene.inl      geometry ingest, commit (build / refit / tree cache), textures, materials, lights, sample tables
//   hr_pipeline.inl   ray memory, the macro step, batching and the packet selector, hr_render_pass, the step log
// (one translation unit: the two .inl files are sections of this one, included below)
#include "hr_ctx.h"

static const int kTableRing = 4;
static int drainPipeline(hr_ctx *c);
// A kernel found a ray queue longer than its capacity (hr_render.hip: queueOverflow): rays were dropped, the frame is not the render
// that was asked for.  Sticky until hr_clear; every call that hands finished work to the caller reports it.
static int overflowCheck(hr_ctx *c)
{
    if (!c->hOverflow || c->hOverflow[0] == 0u) return HR_OK;
    static const char *kinds[] = {"?", "camera rays", "closest-hit queue (input)", "occlusion queue (input)", "closest-hit queue (emitted rays)", "occlusion queue (emitted rays)", "hit list"};
    const uint32_t kind = c->hOverflow[0];
    c->err = std::string("ray queue overflow: ") + kinds[kind < 7u ? kind : 0u] + " of table entry " + std::to_string(c->hOverflow[2]) + " in macro step " +
             std::to_string(c->hOverflow[1] ? c->hOverflow[1] - 1u : 0u) + " held " + std::to_string(c->hOverflow[3]) +
             " rays, more than the host provided for; rays were dropped (hr_clear resets the frame and this report)";
    return HR_ERR_DEVICE;
}
// finish every enqueued pass and wait for the device: required before anything the in-flight kernels read changes
static int quiesce(hr_ctx *c)
{
    int rc = drainPipeline(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HR_OK;
}
#define QUIESCE(ctx)              \
    do {                          \
        int rc_ = quiesce(ctx);   \
        if (rc_) return rc_;      \
    } while (0)

static int occupiedSlots(const hr_ctx *c, int group = -1);
// Batching and the pass pipeline are driven by the passes that follow.  A caller that issues ONE pass per displayed frame (the viewer
// at its refresh rate) must not wait for a batch to fill, nor for ten more passes to push this one through its stages: when the oldest
// unfinished request is more than 4 ms old AND no group has work in flight on the device, everything requested is completed now
// (enqueued, not waited for).  A caller that issues passes faster than the device renders them never meets both conditions for long:
// its passes keep travelling in full batches through a full pipeline.  Called by the progressive (display) read-backs.
static int completeForSlowCaller(hr_ctx *c)
{
    if (c->oldestWaitingNs == 0 || (c->pendingInject.empty() && occupiedSlots(c) == 0)) return HR_OK;
    const unsigned long long now = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (c->tuneSlowMs <= 0 || now - c->oldestWaitingNs <= 1000000ull * (unsigned long long)c->tuneSlowMs) return HR_OK; // (HR_TUNE slow=0: never, for tests of the lag itself)
    for (int g = 0; g < c->nGroups; ++g) {
        const hipError_t q = hipStreamQuery(c->groups[g].stream);
        if (q == hipErrorNotReady) return HR_OK; // work in flight: the pipeline is being fed
        HIP_TRY(c, q);                           // (anything else is a real error of an earlier launch)
    }
    return drainPipeline(c);
}
static void freeLagged(hr_ctx::Lagged &L)
{
    for (int k = 0; k < 3; ++k) {
        if (L.pinned[k]) hipHostFree(L.pinned[k]);
        hipFree(L.dev[k]);
        if (L.ev[k]) hipEventDestroy(L.ev[k]);
        L.pinned[k] = nullptr, L.dev[k] = nullptr, L.ev[k] = nullptr, L.pending[k] = false;
    }
    L.bytes = 0;
}

static int ensureLagged(hr_ctx *c, hr_ctx::Lagged &L, size_t bytes, bool withDevice)
{
    if (L.bytes >= bytes && (!withDevice || L.dev[0])) return HR_OK;
    freeLagged(L);
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(c, hipHostMalloc(&L.pinned[k], bytes, hipHostMallocDefault));
        if (withDevice) HIP_TRY(c, hipMalloc(&L.dev[k], bytes));
        HIP_TRY(c, hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming));
    }
    L.bytes = bytes;
    return HR_OK;
}

// slot to fill now; afterwards `finishLagged` picks what to hand out
static int beginLagged(hr_ctx::Lagged &L) { return L.turn++ % 3; }
static int finishLagged(hr_ctx *c, hr_ctx::Lagged &L, int k, int32_t format, const void **out, uint32_t *passes)
{
    HIP_TRY(c, hipEventRecord(L.ev[k], c->stream));
    L.pending[k] = true, L.epoch[k] = c->snapshotEpoch, L.format[k] = format;
    L.passes[k] = (uint32_t)(c->nextResolveOrder - c->resolvedAtClear);
    const int prev = (k + 2) % 3;
    // nothing in flight (e.g. right after a complete readback): the current snapshot is final, hand it out itself
    const bool idle = c->pendingInject.empty() && occupiedSlots(c) == 0;
    const int use = (!idle && L.pending[prev] && L.epoch[prev] == c->snapshotEpoch && L.format[prev] == format && L.passes[prev] > 0) ? prev : k;
    HIP_TRY(c, hipEventSynchronize(L.ev[use]));
    *out = L.pinned[use];
    if (passes) *passes = L.passes[use];
    return HR_OK;
}

static void freeQueues(hr_ctx *c)
{
    for (hr_ctx::PassSlot &ps : c->slots) {
        hipFree(ps.passbuf);
        if (ps.evFinal) hipEventDestroy(ps.evFinal);
        if (ps.evResolved) hipEventDestroy(ps.evResolved);
        ps = hr_ctx::PassSlot();
    }
    for (hr_ctx::Group &G : c->groups) {
        hipFree(G.arena[0].base), hipFree(G.arena[1].base), hipFree(G.scratch.base);
        G.arena[0] = G.arena[1] = G.scratch = hr_ctx::Group::Region();
        G.arenaHighWater = 0;
    }
    c->nSlotsAllocated = 0;
    c->queueCapacity = 0;
    std::memset(c->stageSeen, 0, sizeof(c->stageSeen)); // (memory budget: back to the guarantee until the stages have been seen again)
}

// how many passes may be in flight: a slot holds a pass buffer (four partial sums once HR_ESTIMATOR_ALL_LIGHTS has been used); the rays
// live in the groups' step arenas, about 7 KB per owned pixel of the frame for both generations of a step's young passes (below)
static void slotBudget(hr_ctx *c)
{
    size_t freeB = 0, totalB = 0;
    c->maxSlots = kMaxSlots;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        const size_t fbBytes = (size_t)c->W * c->H * 4 * sizeof(float);
        const size_t k = c->allLightsUsed ? 4 : 1;
        const size_t perSlot = fbBytes * k + sizeof(Counters);
        const size_t fit = (freeB / 2) / perSlot; // at most half of the free device memory for pass slots
        c->maxSlots = fit < 1 ? 1 : (fit > (size_t)kMaxSlots ? kMaxSlots : (int)fit);
    }
}

static void freeTree(hr_ctx *c)
{
    hipFree(c->tree.nodes), hipFree(c->tree.nodes32), hipFree(c->tree.leafKeys), hipFree(c->tree.tris), hipFree(c->tree.nodeBox), hipFree(c->tree.slotOfPrim);
    c->tree = BuildResult{};
    c->nodes = nullptr, c->tris = nullptr, c->treeTris = 0;
}
static void freeSceneDevice(hr_ctx *c)
{
    freeTree(c);
    hipFree(c->attrs), hipFree(c->attrsExt), hipFree(c->trisPrim), hipFree(c->dG);
    c->attrs = nullptr, c->attrsExt = nullptr, c->trisPrim = nullptr, c->dG = nullptr;
    c->attrsCap = c->attrsExtCap = c->trisPrimCap = c->dGCap = 0;
}

template <class T> static int ensureCap(hr_ctx *c, T **p, size_t *cap, size_t need)
{
    if (*cap >= need && *p) return HR_OK;
    hipFree(*p);
    *p = nullptr, *cap = 0;
    HIP_TRY(c, hipMalloc((void **)p, sizeof(T) * (need ? need : 1)));
    *cap = need;
    return HR_OK;
}

extern "C" {

uint32_t hr_abi_version(void) { return HR_ABI_VERSION; }

int hr_ctx_create(const hr_ctx_desc *desc, hr_ctx **out)
{
    if (!out) return HR_ERR_INVALID;
    *out = nullptr;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) return HR_ERR_DEVICE;
    hr_ctx *c = new hr_ctx();
    if (desc) {
        c->device = desc->device_id;
        c->rank = desc->rank;
        c->world = desc->world > 0 ? desc->world : 1;
        c->tile = desc->tile_size > 0 ? desc->tile_size : 32;
        c->stream = (hipStream_t)desc->stream;
        c->memBudget = desc->memory_budget;
        c->collectStats = (desc->flags & HR_CTX_COLLECT_STATS) != 0;
        c->timeKernels = (desc->flags & HR_CTX_TIME_KERNELS) != 0;
    }
    if (c->device < 0 || c->device >= nDev || c->rank < 0 || c->rank >= c->world || (c->tile & 7) != 0) {
        delete c;
        return HR_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess) {
        delete c;
        return HR_ERR_DEVICE;
    }
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char *t = getenv("HR_TUNE")) {
        auto find = [&](const char *key) -> const char * { // key at the start of the string or right after a comma
            for (const char *p = strstr(t, key); p; p = strstr(p + 1, key))
                if (p == t || p[-1] == ',') return p;
            return nullptr;
        };
        auto get = [&](const char *key, int &dst) {
            if (const char *p = find(key)) dst = atoi(p + strlen(key));
        };
        get("tri=", c->tuneTri), get("refill=", c->tuneRefill), get("blocks=", c->tuneBlocks), get("depth=", c->tuneDepth);
        get("sblocks=", c->tuneShadeBlocks), get("batch=", c->tuneBatch), get("fmax=", c->tuneFetchMax), get("fmin=", c->tuneFetchMin);
        get("groups=", c->tuneGroups), get("prio=", c->tunePrio), get("refit=", c->tuneRefit), get("sdeal=", c->tuneStaticDeal), get("guard=", c->tuneGuardPct), get("ploc=", c->tunePloc), get("packets=", c->tunePackets), get("corun=", c->tuneCorun), get("cmin=", c->tuneCorunMin), get("cblocks=", c->tuneCorunBlocks), get("plog=", c->tuneProbeLog2), get("pswz=", c->tunePacketSwizzle), get("punion=", c->tunePacketUnion), get("plocr=", c->tunePlocRadius), get("fprim=", c->tuneFetchPrimary), get("fgate=", c->tuneFetchGate), get("heads=", c->tuneHeads), get("slow=", c->tuneSlowMs), get("ovf=", c->tuneOverflowTest), get("sprobe=", c->tuneShadowProbe), get("tblk=", c->tuneTableKernel);
        c->tuneBlocksSet = find("blocks=") != nullptr;
        if (c->tuneDepth < 1 || c->tuneDepth > kMaxSlots) c->tuneDepth = kMaxSlots;
        if (c->tuneGroups < 0 || c->tuneGroups > kMaxGroups) c->tuneGroups = 0;
    }
    // Memory the host READS WHILE A KERNEL THAT WRITES IT IS RUNNING (queue lengths, probe totals, the overflow report): coherent
    // (uncached on the device side, fine-grained) whatever HIP_HOST_COHERENT says — hipHostMallocDefault leaves that to the environment
    const unsigned kHostSpun = hipHostMallocCoherent | hipHostMallocMapped;
    bool groupsOk = true;
    for (int g = 0; g < kMaxGroups; ++g) {
        hr_ctx::Group &G = c->groups[g];
        // Worker streams at the highest stream priority, created before anything else uses the device queues: HIP maps the
        // streams of one priority to a small pool of hardware queues, and two groups that land on the same queue do not
        // overlap at all — which happened as soon as the application had a few streams of its own (torch side stream, RCCL).
        // The high-priority pool is practically empty, so the groups get a hardware queue each.  (Creating them lazily, at
        // the first macro step, measurably loses overlap on half- and quarter-frame shards.)
        {
            int least = 0, greatest = 0;
            groupsOk = groupsOk && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
            groupsOk = groupsOk && hipStreamCreateWithPriority(&G.stream, hipStreamNonBlocking, c->tunePrio ? greatest : 0) == hipSuccess;
            groupsOk = groupsOk && hipStreamCreateWithPriority(&G.streamB, hipStreamNonBlocking, c->tunePrio ? greatest : 0) == hipSuccess &&
                       hipEventCreateWithFlags(&G.evFork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&G.evJoin, hipEventDisableTiming) == hipSuccess;
        }
        groupsOk = groupsOk && hipMalloc(&G.dTables, sizeof(StepTable) * kTableRing) == hipSuccess;
        groupsOk = groupsOk && hipHostMalloc((void **)&G.hTables, sizeof(StepTable) * kTableRing, hipHostMallocDefault) == hipSuccess &&
                   hipHostGetDevicePointer((void **)&G.dTablesHost, G.hTables, 0) == hipSuccess;
        groupsOk = groupsOk && hipEventCreateWithFlags(&G.evUser, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < kTableRing; ++k) groupsOk = groupsOk && hipEventCreateWithFlags(&G.tableCopied[k], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < kTableRing; ++k) groupsOk = groupsOk && hipEventCreateWithFlags(&G.statusEv[k], hipEventDisableTiming) == hipSuccess;
        groupsOk = groupsOk && hipHostMalloc((void **)&G.hQCount, sizeof(uint32_t) * kTableRing * kMaxSlots * kMaxBounceSlots, hipHostMallocDefault) == hipSuccess;
        groupsOk = groupsOk && hipHostMalloc((void **)&G.hCounts, sizeof(uint32_t) * (kTableRing * kMaxSegs + 1), kHostSpun) == hipSuccess;
        groupsOk = groupsOk && hipHostMalloc((void **)&G.hSeq, sizeof(unsigned long long) * kTableRing, kHostSpun) == hipSuccess;
        groupsOk = groupsOk && hipHostGetDevicePointer((void **)&G.dCounts, G.hCounts, 0) == hipSuccess &&
                   hipHostGetDevicePointer((void **)&G.dSeq, (void *)G.hSeq, 0) == hipSuccess;
        groupsOk = groupsOk && hipHostMalloc((void **)&G.hProbe, sizeof(unsigned long long) * kTableRing * 4, kHostSpun) == hipSuccess &&
                   hipHostGetDevicePointer((void **)&G.dProbeHost, (void *)G.hProbe, 0) == hipSuccess;
        if (groupsOk)
            G.hCounts[kTableRing * kMaxSegs] = 0u; // (the last word: StepTable::hostCameraCount)
        if (groupsOk)
            for (int k = 0; k < kTableRing; ++k) G.hSeq[k] = 0ull, G.hProbe[4 * k] = 0ull, G.hProbe[4 * k + 1] = 0ull, G.hProbe[4 * k + 2] = 0ull, G.hProbe[4 * k + 3] = 0ull;
        std::memset(G.statusOrder, 0, sizeof(G.statusOrder));
    }
    groupsOk = groupsOk && hipHostMalloc((void **)&c->hOverflow, 4 * sizeof(uint32_t), kHostSpun) == hipSuccess &&
               hipHostGetDevicePointer((void **)&c->dOverflowHost, (void *)c->hOverflow, 0) == hipSuccess;
    if (groupsOk) c->hOverflow[0] = c->hOverflow[1] = c->hOverflow[2] = c->hOverflow[3] = 0u;
    if (!groupsOk || hipMalloc(&c->dScene, sizeof(SceneDev)) != hipSuccess ||
        hipMalloc(&c->dStats, sizeof(Stats) * kStatSlots) != hipSuccess || hipMalloc(&c->dScratch, sizeof(uint32_t) * 6 * kBoundSlots) != hipSuccess ||
        hipMalloc(&c->dZero, 64) != hipSuccess || hipMalloc(&c->dProbe, 64) != hipSuccess || hipMemset(c->dProbe, 0, 64) != hipSuccess ||
        hipStreamCreateWithFlags(&c->probeStream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->evProbeA, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->evProbeB, hipEventDisableTiming) != hipSuccess || hipMalloc(&c->dCounters, sizeof(Counters) * kMaxSlots) != hipSuccess ||
        hipMalloc(&c->dStepLog, sizeof(unsigned long long) * 3 * kStepLogCap) != hipSuccess) {
        delete c;
        return HR_ERR_DEVICE;
    }
    hipMemset(c->dStats, 0, sizeof(Stats) * kStatSlots);
    hipMemset(c->dZero, 0, 64);
    *out = c;
    return HR_OK;
}

int hr_ctx_destroy(hr_ctx *c)
{
    if (!c) return HR_OK;
    hipSetDevice(c->device);
    if (getenv("HR_DEBUG_PIPE"))
        fprintf(stderr, "hr_ctx %p: ray-memory growths %llu (last sizes summed %.1f MiB); queue-length waits %llu, of which %llu had to spin, %.2f ms in total\n", (void *)c,
                c->dbgGrowths, (double)c->dbgGrowBytes / 1048576.0, c->dbgWaits, c->dbgWaitSpun, (double)c->dbgWaitNs * 1e-6);
    drainPipeline(c);
    hipStreamSynchronize(c->stream);
    if (c->dShadowProbe) {
        unsigned long long t[4] = {0, 0, 0, 0};
        hipDeviceSynchronize();
        hipMemcpy(t, c->dShadowProbe, sizeof(t), hipMemcpyDeviceToHost);
        fprintf(stderr, "shadow probe (%s): %llu occlusion rays in %llu packets of 64 consecutive queue entries: union factor U = %.3f, %.1f child boxes entered per ray\n",
                c->tuneShadowProbe == 2 ? "every stage" : "first bounce", t[3], t[2], t[1] ? (double)t[0] / (double)t[1] : 0.0, t[3] ? (double)t[1] / (double)t[3] : 0.0);
        hipFree(c->dShadowProbe);
    }
    if (c->probeStream) hipStreamSynchronize(c->probeStream), hipStreamDestroy(c->probeStream);
    if (c->evProbeA) hipEventDestroy(c->evProbeA);
    if (c->evProbeB) hipEventDestroy(c->evProbeB);
    c->drainTimes();
    for (hipEvent_t e : c->eventPool) hipEventDestroy(e);
    if (c->evPack) hipEventDestroy(c->evPack);
    freeQueues(c);
    freeSceneDevice(c);
    for (Texture &t : c->textures) hipFree(t.dpx);
    hipFree(c->fbInternal);
    if (c->pinned) hipHostFree(c->pinned);
    hipFree(c->dDisplay);
    c->meshReleaseAll();
    for (int k = 0; k < 2; ++k) {
        if (c->stage[k]) hipHostFree(c->stage[k]);
        if (c->stageEv[k]) hipEventDestroy(c->stageEv[k]);
    }
    hipFree(c->dConsts);
    hipFree(c->dEnvRowCdf), hipFree(c->dEnvColCdf), hipFree(c->dEnvProb), hipFree(c->dEnvRowGuide), hipFree(c->dEnvColGuide);
    if (c->hConsts) hipHostFree(c->hConsts);
    if (c->pinnedDisplay) hipHostFree(c->pinnedDisplay);
    hipFree(c->dMaterials), hipFree(c->dTextures), hipFree(c->dSeq), hipFree(c->dAperture), hipFree(c->dSeqOffsets);
    hipFree(c->dTexDensity);
    for (Texture &t : c->textures) hipFree(t.dmips);
    hipFree(c->dScene), hipFree(c->dStats), hipFree(c->dScratch), hipFree(c->dZero), hipFree(c->dProbe), hipFree(c->dCounters), hipFree(c->dStepLog);
    if (c->hOverflow) hipHostFree((void *)c->hOverflow);
    for (hr_ctx::Group &G : c->groups) {
        if (G.hQCount) hipHostFree(G.hQCount);
        if (G.hCounts) hipHostFree(G.hCounts);
        if (G.hSeq) hipHostFree((void *)G.hSeq);
        if (G.hProbe) hipHostFree((void *)G.hProbe);
        for (hipEvent_t e : G.statusEv)
            if (e) hipEventDestroy(e);
        if (G.stream) hipStreamDestroy(G.stream);
        if (G.streamB) hipStreamDestroy(G.streamB);
        if (G.evFork) hipEventDestroy(G.evFork);
        if (G.evJoin) hipEventDestroy(G.evJoin);
        hipFree(G.dTables);
        if (G.hTables) hipHostFree(G.hTables);
        if (G.evUser) hipEventDestroy(G.evUser);
        for (hipEvent_t e : G.tableCopied)
            if (e) hipEventDestroy(e);
    }
    delete c;
    return HR_OK;
}

const char *hr_last_error(const hr_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int hr_ctx_set_stream(hr_ctx *c, void *stream)
{
    ENTER(c);
    QUIESCE(c);
    c->stream = (hipStream_t)stream;
    return HR_OK;
}

// the frame geometry of `rank` of `world` (the ctx's own when they match its description)
static FrameDev frameOf(const hr_ctx *c, int32_t rank, int32_t world)
{
    FrameDev f = c->frame;
    f.rank = rank, f.world = world;
    const int nTiles = f.tilesX * f.tilesY;
    f.nOwnedTiles = nTiles > rank ? (nTiles - rank + world - 1) / world : 0;
    return f;
}

int hr_frame_packed_slots(hr_ctx *c, int32_t rank, int32_t world, uint64_t *n_slots)
{
    ENTER(c);
    if (!n_slots || world <= 0 || rank < 0 || rank >= world) FAIL(c, HR_ERR_INVALID, "bad rank / world");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    *n_slots = (uint64_t)frameOf(c, rank, world).nOwnedTiles * (uint64_t)(c->tile * c->tile);
    return HR_OK;
}

int hr_frame_pack_owned(hr_ctx *c, void *device_out, void *stream)
{
    ENTER(c);
    if (!device_out) FAIL(c, HR_ERR_INVALID, "null output");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (st != c->stream) { // the resolves enqueued so far run on the ctx stream: order the copy behind them
        if (!c->evPack) HIP_TRY(c, hipEventCreateWithFlags(&c->evPack, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->evPack, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(st, c->evPack, 0));
    }
    launchPackOwned(c->cfg(st), fr, c->fb(), (float *)device_out, 0, nullptr);
    HIP_TRY(c, hipGetLastError());
    if (st != c->stream) { // ... and the next resolve (which rewrites the frame) behind the copy
        HIP_TRY(c, hipEventRecord(c->evPack, st));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evPack, 0));
    }
    return HR_OK;
}

int hr_frame_unpack(hr_ctx *c, int32_t src_rank, int32_t world, const void *device_packed, void *device_full_frame, void *stream)
{
    ENTER(c);
    if (!device_packed || !device_full_frame) FAIL(c, HR_ERR_INVALID, "null argument");
    if (world <= 0 || src_rank < 0 || src_rank >= world) FAIL(c, HR_ERR_INVALID, "bad rank / world");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    launchPackOwned(c->cfg(stream ? (hipStream_t)stream : c->stream), frameOf(c, src_rank, world), nullptr, (float *)device_packed, 1,
                    (float *)device_full_frame);
    HIP_TRY(c, hipGetLastError());
    return HR_OK;
}

static size_t displayPixelBytes(int32_t format) { return format == HR_DISPLAY_RGBA8 ? 4 : 16; }

int hr_display(hr_ctx *c, const hr_display_params *params, int32_t format, void *device_out, uint32_t *passes_shown)
{
    ENTER(c);
    if (!params || !device_out) FAIL(c, HR_ERR_INVALID, "null argument");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    const bool progressive = (format & HR_DISPLAY_PROGRESSIVE) != 0;
    format &= ~HR_DISPLAY_PROGRESSIVE;
    if (format < HR_DISPLAY_RGBA8 || format > HR_DISPLAY_HDR_RGBA32F) FAIL(c, HR_ERR_INVALID, "unknown display format");
    if (!progressive) {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    launchDisplay(c->cfg(c->stream), fr, *params, format, device_out);
    HIP_TRY(c, hipGetLastError());
    if (passes_shown) *passes_shown = (uint32_t)(c->nextResolveOrder - c->resolvedAtClear); // (the resolves enqueued before this kernel, same stream)
    return HR_OK;
}

int hr_frame_passes_resolved(hr_ctx *c, uint64_t *passes)
{
    ENTER(c);
    if (!passes) FAIL(c, HR_ERR_INVALID, "null output");
    *passes = c->nextResolveOrder - c->resolvedAtClear;
    return HR_OK;
}

int hr_display_readback(hr_ctx *c, const hr_display_params *params, int32_t format, const void **pixels, int32_t *width, int32_t *height, uint32_t *passes_shown)
{
    ENTER(c);
    if (!pixels) FAIL(c, HR_ERR_INVALID, "null output");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    {
        const int rc = overflowCheck(c);
        if (rc) return rc;
    }
    const size_t need = (size_t)c->W * c->H * 16;
    if (c->displayBytes < need) {
        hipFree(c->dDisplay);
        if (c->pinnedDisplay) hipHostFree(c->pinnedDisplay);
        c->dDisplay = nullptr, c->pinnedDisplay = nullptr, c->displayBytes = 0;
        HIP_TRY(c, hipMalloc(&c->dDisplay, need));
        HIP_TRY(c, hipHostMalloc(&c->pinnedDisplay, need, hipHostMallocDefault));
        c->displayBytes = need;
    }
    if (format & HR_DISPLAY_PROGRESSIVE) { // lagged, like hr_readback_progressive
        int rc = completeForSlowCaller(c);
        if (rc) return rc;
        rc = ensureLagged(c, c->progDisplay, need, true);
        if (rc) return rc;
        const int k = beginLagged(c->progDisplay);
        rc = hr_display(c, params, format, c->progDisplay.dev[k], nullptr);
        if (rc) return rc;
        const size_t nb = (size_t)c->W * c->H * displayPixelBytes(format & ~HR_DISPLAY_PROGRESSIVE);
        HIP_TRY(c, hipMemcpyAsync(c->progDisplay.pinned[k], c->progDisplay.dev[k], nb, hipMemcpyDeviceToHost, c->stream));
        // the parameters are part of the snapshot's identity: a change of settings must not hand out an old image
        int32_t key = format;
        for (size_t i = 0; i < sizeof(*params) / 4; ++i) key = key * 31 + ((const int32_t *)params)[i];
        rc = finishLagged(c, c->progDisplay, k, key, pixels, passes_shown); // (the passes of the snapshot handed out, which may be the previous call's)
        if (rc) return rc;
        if (width) *width = c->W;
        if (height) *height = c->H;
        return HR_OK;
    }
    int rc = hr_display(c, params, format, c->dDisplay, passes_shown);
    if (rc) return rc;
    const size_t bytes = (size_t)c->W * c->H * displayPixelBytes(format & ~HR_DISPLAY_PROGRESSIVE);
    HIP_TRY(c, hipMemcpyAsync(c->pinnedDisplay, c->dDisplay, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *pixels = c->pinnedDisplay;
    if (width) *width = c->W;
    if (height) *height = c->H;
    return HR_OK;
}

int hr_synchronize(hr_ctx *c)
{
    ENTER(c);
    QUIESCE(c);
    return overflowCheck(c);
}

// ------------------------------------------------------------------------------------------ frame
int hr_frame_resize(hr_ctx *c, int32_t w, int32_t h)
{
    ENTER(c);
    if (w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) FAIL(c, HR_ERR_INVALID, "bad frame size");
    QUIESCE(c);
    c->W = w, c->H = h;
    c->snapshotEpoch++;
    c->probeCountdown = 0; // (packet selector: another resolution)
    freeLagged(c->progFrame), freeLagged(c->progDisplay);
    hipFree(c->fbInternal);
    c->fbInternal = nullptr;
    c->fbExternal = nullptr;
    const size_t fbBytes = (size_t)w * h * 4 * sizeof(float);
    HIP_TRY(c, hipMalloc(&c->fbInternal, fbBytes));
    HIP_TRY(c, hipMemsetAsync(c->fbInternal, 0, fbBytes, c->stream));
    if (c->pinnedBytes < fbBytes) {
        if (c->pinned) hipHostFree(c->pinned);
        c->pinned = nullptr, c->pinnedBytes = 0;
        HIP_TRY(c, hipHostMalloc((void **)&c->pinned, fbBytes, hipHostMallocDefault));
        c->pinnedBytes = fbBytes;
    }
    FrameDev &f = c->frame;
    f.W = w, f.H = h, f.rank = c->rank, f.world = c->world, f.tile = c->tile;
    f.tilesX = (w + c->tile - 1) / c->tile, f.tilesY = (h + c->tile - 1) / c->tile;
    const int nTiles = f.tilesX * f.tilesY;
    f.nOwnedTiles = nTiles > c->rank ? (nTiles - c->rank + c->world - 1) / c->world : 0;
    // one path per owned pixel and pass: queue capacity = owned tiles x tile^2; pass slots are allocated on demand
    freeQueues(c);
    c->queueCapacity = (uint32_t)f.nOwnedTiles * (uint32_t)(c->tile * c->tile);
    // how many passes may be in flight: each slot holds two ray queues, an occlusion queue, hit records and a pass buffer
    slotBudget(c);
    {
        // Paths per macro step worth launching for.  A trace launch ends in a tail of a few long rays (0.5-0.7 ms whatever it
        // carries) and every pass costs depth + 2 dependent launches, so passes requested back to back are collected and injected
        // together: 9 passes of a 1080p frame per step measured 1715 against 1607 Mrays/s (128 passes) and 1488 against 1400
        // (20 passes) for one at a time on two pipeline groups (profiles/r2b_batch_sweep*.txt).  Round 3, with the small launches
        // dealt out statically and the shading stage split: 11-14 passes per step are another 3-4 % over 9 at 20 passes and 2 % at
        // 128 (profiles/r3n_batch_sweep.txt); 12 it is (120 pass buffers of a 1080p frame and the step arenas: ~21 GB of the 288; 53 GB before round 4 sized the queues by stage).  A caller that asks for
        // pixels after every pass (hr_readback) completes what is pending, so batching never delays a displayed frame.
        const long long target = 12ll * 1920ll * 1080ll;
        const long long own = c->queueCapacity ? c->queueCapacity : 1;
        long long b = (target + own - 1) / own;
        c->injectBatch = (int)(b < 1 ? 1 : (b > HR_BATCH_CAP ? HR_BATCH_CAP : b));
        if (c->tuneBatch > 0) c->injectBatch = c->tuneBatch;
        // Two pipeline groups (their steps alternate on two streams, so one group's trace tail and its shade / raygen run
        // under the other group's trace) pay off only while a launch carries little work: +11..14 % on a 1080p frame at one pass
        // per step; with several passes per step one group (five trace workgroups per CU) is as fast or faster (1697 vs 1696
        // Mrays/s at 8 passes per step) and needs half the pass slots.
        c->nGroups = c->tuneGroups > 0 ? c->tuneGroups : (c->injectBatch >= 4 || c->queueCapacity > 4200000u ? 1 : 2);
        c->nextGroup = 0;
        if (!c->tuneBlocksSet) c->tuneBlocks = c->nGroups > 1 ? 3 : 5;
        c->pendingInject.clear();
    }
    return HR_OK;
}

int hr_frame_bind_external(hr_ctx *c, void *deviceRgba)
{
    ENTER(c);
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    QUIESCE(c);
    c->fbExternal = (float *)deviceRgba;
    c->snapshotEpoch++;
    return HR_OK;
}

int hr_frame_device_ptr(hr_ctx *c, void **deviceRgba)
{
    ENTER(c);
    if (c->W <= 0 || !deviceRgba) FAIL(c, HR_ERR_INVALID, "no frame");
    *deviceRgba = c->fb();
    return HR_OK;
}

// --------------------------------------------------------------------------------------- geometry
#include "hr_scene.inl"
#include "hr_pipeline.inl"

int hr_readback(hr_ctx *c, const float **rgba, int32_t *w, int32_t *h)
{
    ENTER(c);
    if (c->W <= 0 || !rgba) FAIL(c, HR_ERR_INVALID, "no frame");
    const size_t bytes = (size_t)c->W * c->H * 4 * sizeof(float);
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    HIP_TRY(c, hipMemcpyAsync(c->pinned, c->fb(), bytes, hipMemcpyDeviceToHost, c->stream));
    QUIESCE(c);
    {
        const int rc = overflowCheck(c);
        if (rc) return rc;
    }
    *rgba = c->pinned;
    if (w) *w = c->W;
    if (h) *h = c->H;
    return HR_OK;
}

int hr_readback_progressive(hr_ctx *c, const float **rgba, int32_t *w, int32_t *h, uint32_t *passes)
{
    ENTER(c);
    if (c->W <= 0 || !rgba) FAIL(c, HR_ERR_INVALID, "no frame");
    const size_t bytes = (size_t)c->W * c->H * 4 * sizeof(float);
    int rc = completeForSlowCaller(c);
    if (rc == HR_OK) rc = overflowCheck(c);
    if (rc) return rc;
    // no drain: the resolves enqueued so far are ordered before this copy on the ctx stream
    rc = ensureLagged(c, c->progFrame, bytes, false);
    if (rc) return rc;
    const int k = beginLagged(c->progFrame);
    HIP_TRY(c, hipMemcpyAsync(c->progFrame.pinned[k], c->fb(), bytes, hipMemcpyDeviceToHost, c->stream));
    const void *out = nullptr;
    rc = finishLagged(c, c->progFrame, k, 0, &out, passes);
    if (rc) return rc;
    *rgba = (const float *)out;
    if (w) *w = c->W;
    if (h) *h = c->H;
    return HR_OK;
}

int hr_debug_trace(hr_ctx *c, int32_t n, const float *o, const float *d, const float *tmax, const int32_t *skip, int32_t anyHit, hr_hit *out)
{
    ENTER(c);
    if (!c->committed) FAIL(c, HR_ERR_INVALID, "scene not committed");
    if (n <= 0 || !o || !d || !out) FAIL(c, HR_ERR_INVALID, "bad arguments");
    int rc = uploadScene(c);
    if (rc) return rc;
    float *dO = nullptr, *dD = nullptr, *dT = nullptr;
    int *dS = nullptr;
    hr_hit *dH = nullptr;
    HIP_TRY(c, hipMalloc(&dO, (size_t)n * 12));
    HIP_TRY(c, hipMalloc(&dD, (size_t)n * 12));
    HIP_TRY(c, hipMalloc(&dH, (size_t)n * sizeof(hr_hit)));
    HIP_TRY(c, hipMemcpy(dO, o, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(dD, d, (size_t)n * 12, hipMemcpyHostToDevice));
    if (tmax) {
        HIP_TRY(c, hipMalloc(&dT, (size_t)n * 4));
        HIP_TRY(c, hipMemcpy(dT, tmax, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    if (skip) {
        HIP_TRY(c, hipMalloc(&dS, (size_t)n * 4));
        HIP_TRY(c, hipMemcpy(dS, skip, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    launchDebugTrace(c->cfg(c->stream), c->dScene, n, dO, dD, dT, dS, anyHit, dH);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dH, (size_t)n * sizeof(hr_hit), hipMemcpyDeviceToHost);
    hipFree(dO), hipFree(dD), hipFree(dT), hipFree(dS), hipFree(dH);
    HIP_TRY(c, e);
    return HR_OK;
}

} // extern "C"
