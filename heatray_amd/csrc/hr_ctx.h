// hr_ctx.h — the context behind the opaque hr_ctx handle of include/hrcore.h: everything one context owns on the device and the
// state of its pass pipeline, plus the error macros of the host side.  Included by hr_core.hip only (ONE translation unit: the
// host side's helpers keep internal linkage); hr_scene.inl and hr_pipeline.inl are that file's two large sections.
#pragma once
#include "hr_kernels.h"
#include "hr_trace.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

using namespace hr;

namespace {

struct Texture {
    void *dpx = nullptr;
    float *dmips = nullptr; // levels >= 1 (HR_TEXTURE_LOD_CONE), built on first use
    TexDesc desc{};
    bool alive = false;
};

// One submesh.  Its vertex attributes and indices live in ONE device block, uploaded when the mesh is added (straight from the
// caller's planar buffers through a pinned staging ring, with the caller's strides: nothing is de-interleaved or kept on the host).
struct Geom {
    bool alive = false;
    int nVerts = 0;
    uint32_t nIdx = 0;
    int mode = HR_TRIANGLES;
    float world[16];
    int frontFaceCW = 0, isOccluder = 1, material = 0;
    char *dBlock = nullptr;   // inside chunk `chunk` of the context's mesh arena
    int chunk = -1;
    size_t blockBytes = 0;
    size_t off[7] = {0, 0, 0, 0, 0, 0, 0}; // byte offsets of pos, nrm, uv, tan, bit, col, idx in the block
    bool has[6] = {false, false, false, false, false, false};
    int stride[6] = {3, 3, 2, 3, 3, 3};     // floats between consecutive vertices
    uint32_t nTris() const { return mode == HR_TRIANGLE_STRIP ? (nIdx >= 3 ? nIdx - 2 : 0u) : nIdx / 3; }
};

} // namespace

static const int kMaxGroups = 3;
static const int kMaxSlots = 2 * kMaxSegs; // passes in flight over all groups
#ifndef HR_BATCH_CAP
#define HR_BATCH_CAP 32 // most passes injected per macro step (small frames / tile shards reach it: 1/8 of a 1080p frame runs 6.6 % faster with 32 than with 12, profiles/r2p_shard_batch.txt)
#endif

struct hr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool collectStats = false;
    bool textureLodUsed = false; // a pass has asked for HR_TEXTURE_LOD_CONE (kernel variant, see LaunchCfg)
    bool allLightsUsed = false;  // a pass has asked for HR_ESTIMATOR_ALL_LIGHTS: pass slots hold two occlusion rays per path and a second partial sum
    int rank = 0, world = 1, tile = 32;
    int numCUs = 256;
    std::string err;

    // frame
    int W = 0, H = 0;
    float *fbInternal = nullptr, *fbExternal = nullptr;
    float *pinned = nullptr;
    size_t pinnedBytes = 0;
    hipEvent_t evPack = nullptr; // orders hr_frame_pack_owned on a foreign stream against the resolves on the ctx stream
    void *dDisplay = nullptr, *pinnedDisplay = nullptr; // display resolve: device staging + pinned host copy
    // Progressive snapshots are handed out one call late from rotating buffers: the host then waits for a copy enqueued a
    // whole call ago instead of for everything it has just enqueued, so the GPU always has the next step queued
    // (waiting for the latest copy cost 0.8 ms of idle GPU per pass).
    struct Lagged {
        void *pinned[3] = {nullptr, nullptr, nullptr};
        void *dev[3] = {nullptr, nullptr, nullptr}; // device staging (display snapshots only)
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        uint32_t passes[3] = {0, 0, 0};
        unsigned long long epoch[3] = {0, 0, 0};
        int32_t format[3] = {-1, -1, -1};
        bool pending[3] = {false, false, false};
        size_t bytes = 0;
        int turn = 0;
    };
    Lagged progFrame, progDisplay;
    unsigned long long snapshotEpoch = 1; // bumped by clear / resize / bind: older snapshots are not handed out any more
    size_t displayBytes = 0;
    FrameDev frame{};
    uint32_t queueCapacity = 0;
    // Pipeline of in-flight passes (hr_render.hip header): every slot owns the queues, hit records, counters and
    // the pass buffer of one pass.
    struct PassSlot {
        bool allocated = false, active = false;
        bool finished = false;   // every stage has been enqueued; the slot is held until its turn to resolve comes
        bool everResolved = false;
        unsigned long long resolvedAt = 0; // value of nextResolveOrder when this slot's last pass was resolved
        int group = 0;           // pipeline group (worker stream) the pass runs on
        hipEvent_t evFinal = nullptr;    // recorded on the worker stream after the pass's last stage
        hipEvent_t evResolved = nullptr; // recorded on the caller's stream after the pass buffer was added to the frame
        // Passes that finish in one macro step, and passes one k_resolve launch adds, share ONE recorded event (a record or a wait is a
        // packet of ~4-8 us on its stream: twelve of each per batch delayed the resolve of a batch by 0.1 ms and its next injection by as
        // much).  The slot that owns the recorded event may be reused later; whoever waits has enqueued the wait before that (same call).
        hipEvent_t finalEv = nullptr;    // the event to wait on for this pass's last stage (some slot's evFinal)
        hipEvent_t resolvedEv = nullptr; // ... for the launch that added this pass buffer to the frame (some slot's evResolved)
        int step = 0, nIter = 0;
        unsigned long long order = 0; // injection order (passes resolve in this order)
        hr_pass_params pp{};
        // The pass's rays live in its group's step arenas (Group::arena): what its last step's shading emitted, i.e. what its next
        // step traces.  Only the pass buffer belongs to the slot.
        RayQueue qcur{};       // closest-hit rays of the pass's next stage
        ShadowQueue scur{};    // occlusion rays of the pass's next stage
        uint32_t capCur = 0;   // rays qcur can hold (= upper bound of what it holds)
        uint32_t sCapCur = 0;  // occlusion rays scur can hold
        float *passbuf = nullptr;
        float *passbufB = nullptr; // second partial sum (allLightsUsed): passbuf + W * H * 4, same allocation
        Counters *ctr = nullptr;
    };
    PassSlot slots[kMaxSlots];
    int nSlotsAllocated = 0;
    int maxSlots = kMaxSlots; // bounded by device memory at resize
    // Pipeline groups: independent pass pipelines on their own HIP streams, stepped alternately, so that the tail of one
    // group's persistent trace kernel (waves running dry) is back-filled by the other group's kernels.  Resolves run on
    // the caller's stream, strictly in pass order.
    struct Group {
        hipStream_t stream = nullptr;
        StepTable *dTables = nullptr;   // ring of device step tables
        StepTable *hTables = nullptr;   // pinned staging ring
        StepTable *dTablesHost = nullptr; // ... as the device addresses it
        hipEvent_t tableCopied[4] = {nullptr, nullptr, nullptr, nullptr};
        bool tableUsed[4] = {false, false, false, false};
        unsigned long long stepCounter = 0;
        hipEvent_t evUser = nullptr;    // caller-stream state this group has to wait for
        bool needUserSync = true;
        // Pass-through scenes (single-sided / alpha-masked materials): a pass has no fixed number of stages, so after every
        // macro step the closest-queue lengths of all pass slots are copied to a pinned ring; the host reads the copy of TWO
        // steps ago (a step that has long finished while newer ones are still queued: it never waits for work it has just
        // enqueued) and retires the passes whose queue ran empty.
        uint32_t *hQCount = nullptr;    // [kStatusRing][kMaxSlots][kMaxBounceSlots], pinned
        // Ray memory of the group (round 4).  A pass used to own two ray queues, an occlusion queue, hit records and a hit list, all
        // sized for EVERY owned pixel, for its whole life: 196 B x pixels x 120 slots = 53 GB for a 1080p render, while a pass past
        // its first bounce holds a few percent of the pixels.  Now every macro step carves what it needs out of three regions:
        //   arena[t & 1]  what step t's shading emits (closest-hit and occlusion rays of every in-flight pass), read by step t + 1;
        //   scratch       what lives inside one step: the injected passes' camera rays, hit records, hit lists.
        // A queue is sized by an upper bound of what can arrive in it: a ray emits at most one continuation ray and kS occlusion rays,
        // so the bound is the length of the pass's closest-hit queue ONE stage earlier — which k_trace itself reports: its first
        // workgroup writes, when it starts, the queue lengths of its step table to pinned host memory and then the step's number
        // (hCounts / hSeq; no packet on the stream).  Preparing step t the host waits for step t - 1's report (by then step t - 2 has
        // finished and all of step t - 1 is still queued: the device never runs dry); only a pass's FIRST stage is sized by pixels.
        // Regions grow on demand (a synchronisation of the group's stream, during the first passes of a render).
        struct Region {
            char *base = nullptr;
            size_t cap = 0;
        };
        Region arena[2], scratch;
        size_t arenaHighWater = 0; // most either half ever needed: both halves are kept that large (consecutive steps see the same load)
        uint32_t *hCounts = nullptr;                   // [kTableRing][kMaxSegs], pinned: closest-hit queue length per table entry
        volatile unsigned long long *hSeq = nullptr;   // [kTableRing], pinned: step number + 1 whose lengths the entry holds
        uint32_t *dCounts = nullptr;                   // the same two arrays as the device addresses them
        unsigned long long *dSeq = nullptr;
        hipStream_t streamB = nullptr;                 // HR_TUNE corun=1: the fused packet kernel of a step runs here, beside k_trace (experiment)
        hipEvent_t evFork = nullptr, evJoin = nullptr;
        volatile unsigned long long *hProbe = nullptr; // [kTableRing][3], pinned: the packet probe's totals as of that step's k_trace (packet selector below)
        unsigned long long *dProbeHost = nullptr;      // ... as the device addresses it
        int countN[4] = {0, 0, 0, 0};                  // entries of the step table that went with ring entry r
        int countSlot[4][HR_MAX_SEGS];                 // ... their pass slots
        unsigned long long countOrder[4][HR_MAX_SEGS]; // ... and passes (order + 1)
        hipEvent_t statusEv[4] = {nullptr, nullptr, nullptr, nullptr};
        bool statusUsed[4] = {false, false, false, false};
        unsigned long long statusOrder[4][2 * HR_MAX_SEGS]; // pass (order + 1) a slot held when the snapshot was taken, 0 = none
    };
    Group groups[kMaxGroups];
    int nGroups = 2;      // groups in use: chosen per frame size in hr_frame_resize unless HR_TUNE fixes it
    int tuneGroups = 0;   // HR_TUNE="groups=N" (0 = automatic)
    int tunePrio = 1;     // HR_TUNE="prio=0": worker streams at normal priority
    int tuneBlocksSet = 0; // HR_TUNE="blocks=N" given
    int nextGroup = 0;
    unsigned long long nextResolveOrder = 0;
    unsigned long long resolvedAtClear = 0; // value of nextResolveOrder at the last hr_clear
    // Passes requested but not yet injected: when a shard is small (multi-GPU tiles, small frames) several passes are
    // injected per macro step so that every launch still carries about a full 1080p pass worth of rays.
    std::deque<hr_pass_params> pendingInject;
    unsigned long long oldestWaitingNs = 0; // steady-clock time of the oldest pass request not yet completed by a drain (0: none)
    int injectBatch = 1;
    int lastDepth = -1;
    unsigned long long injected = 0;
    uint32_t *dZero = nullptr;    // a zero word (occlusion count of a pass's first step)
    Counters *dCounters = nullptr; // one per pass slot, contiguous (copied to the host in one piece in pass-through scenes)
    unsigned long long *dStepLog = nullptr; // kStepLogCap records of three words (StepTable::stepLog)
    // Sticky report of a ray queue that turned out longer than its capacity (hr_render.hip: queueOverflow): four pinned, coherent words
    // the kernels write — kind of queue, step, table entry, count.  Checked wherever the caller learns about finished work.
    volatile uint32_t *hOverflow = nullptr;
    uint32_t *dOverflowHost = nullptr; // ... as the device addresses them
    // hr_ctx_desc::memory_budget: device bytes the pipeline may hold for rays and pass buffers (0: unlimited).  Bounds the passes
    // injected per step (budgetBatch): first by what a batch needs when every queue is as long as it can get, then — once a full
    // pipeline has shown the real lengths — by what it was seen to need (rayBytesSeen / batchSeen), with a fifth on top.
    unsigned long long memBudget = 0;
    // ray memory a pass needs at stage s of its life — what a step carves for it in the arena / in scratch —, the largest per-pass average
    // seen so far (0: that stage has not been seen since the last resize / commit: it counts as long as it can possibly get)
    double stageArenaSeen[kMaxBounceSlots] = {0}, stageScratchSeen[kMaxBounceSlots] = {0};
    bool stageSeen[kMaxBounceSlots] = {false};
    int tuneTableKernel = 1;  // HR_TUNE="tblk=0": the step table goes to the device by hipMemcpyAsync instead of a fetch kernel reading its pinned entry (0.2-0.7 % slower: profiles/r5k_table_fetch.txt)
    int tuneShadowProbe = 0;  // HR_TUNE="sprobe=1|2" (measurement): walk the occlusion queues of the first bounce (1) / of every stage (2) as packets of 64 consecutive rays and print their union factor when the context goes
    unsigned long long *dShadowProbe = nullptr;
    int tuneOverflowTest = 0;          // HR_TUNE="ovf=1|2|3": TEST ONLY — halve one bound so that a queue overflows (1: camera rays, 2: a stage's closest-hit bound, 3: occlusion rays)

    // Mesh blocks come out of an arena of 64 MB chunks (bump allocation inside a chunk): a hipMalloc per submesh is a device-wide
    // synchronisation of ~0.1 ms each, which adds up for the scenes the reference loads (hundreds of submeshes).  A chunk whose last
    // mesh has been removed is empty again: one such chunk is kept for the next add (a lone dynamic mesh that is removed and re-added
    // every frame costs no hipFree + hipMalloc), further ones are released, and their entries in the vector are reused.
    struct MeshChunk {
        char *base = nullptr;
        size_t cap = 0, used = 0;
        int live = 0;
    };
    std::vector<MeshChunk> meshChunks;
    char *meshAlloc(size_t bytes, int *chunkOut)
    {
        const size_t need = (bytes + 255) & ~(size_t)255;
        // the newest chunk first (it is the one being filled), then any other with room (e.g. one that ran empty)
        for (int i = (int)meshChunks.size() - 1; i >= 0; --i) {
            MeshChunk &k = meshChunks[i];
            if (k.base && k.cap - k.used >= need) {
                char *p = k.base + k.used;
                k.used += need, k.live += 1;
                *chunkOut = i;
                return p;
            }
        }
        MeshChunk k;
        k.cap = need > ((size_t)64 << 20) ? need : ((size_t)64 << 20);
        if (hipMalloc((void **)&k.base, k.cap) != hipSuccess) return nullptr;
        k.used = need, k.live = 1;
        for (size_t i = 0; i < meshChunks.size(); ++i)
            if (!meshChunks[i].base) { // a released chunk's entry (other meshes refer to chunks by index, so entries never move)
                meshChunks[i] = k;
                *chunkOut = (int)i;
                return k.base;
            }
        meshChunks.push_back(k);
        *chunkOut = (int)meshChunks.size() - 1;
        return k.base;
    }
    void meshRelease(int chunk)
    {
        if (chunk < 0 || chunk >= (int)meshChunks.size()) return;
        MeshChunk &k = meshChunks[chunk];
        if (!k.base || --k.live > 0) return;
        k.live = 0, k.used = 0; // empty: its space is handed out again
        int spare = 0;
        for (const MeshChunk &o : meshChunks) spare += (o.base && o.live == 0) ? 1 : 0;
        if (spare > 1 || k.cap > ((size_t)64 << 20)) { // keep ONE empty default-sized chunk
            hipFree(k.base);
            k.base = nullptr, k.cap = 0;
        }
    }
    void meshReleaseAll()
    {
        for (MeshChunk &k : meshChunks) hipFree(k.base);
        meshChunks.clear();
    }
    // scene (host mirror)
    std::vector<Geom> geoms;
    std::vector<Texture> textures;
    std::vector<hr_material> materials;
    hr_lights lights{};
    int32_t blockNx = 0, blockNy = 0, blockCoords[32] = {0};
    // importance table of the environment map (HR_ESTIMATOR_ENV_MIS), built on the device when a pass first asks for it
    float *dEnvRowCdf = nullptr, *dEnvColCdf = nullptr, *dEnvProb = nullptr;
    uint16_t *dEnvRowGuide = nullptr, *dEnvColGuide = nullptr;
    int envW = 0, envH = 0, envTex = -2;
    float envMeanLum = 0.0f;
    bool committed = false, sceneDirty = true, hasPassthrough = false;
    bool hasGlass = false; // some material is glass (decides whether the glass shading kernel is launched)
    // What changed since the last commit decides what a commit does: a change of the set of geometries rebuilds the tree, a
    // change of transforms only (Scene::applyTransform while the user drags a slider) REFITS it — same topology, every box
    // recomputed bottom-up on the device, no allocation, one synchronisation at the end.
    bool topologyDirty = true, transformDirty = false;
    int tuneRefit = 1;        // HR_TUNE="refit=0": always rebuild
    // pipeline diagnostics (HR_DEBUG_PIPE=1 prints them when the context is destroyed)
    unsigned long long dbgGrowths = 0, dbgGrowBytes = 0, dbgWaits = 0, dbgWaitNs = 0, dbgWaitSpun = 0;
    // ---- packet selector.  The camera rays of the passes injected together can be traced one ray per lane by k_trace, or 64 at a time as a
    // packet by k_raygen_packets (hr_render.hip): 2^k passes of 64 >> k neighbouring pixels per wave.  The packet walks the UNION of its
    // rays' node sets: it wins where that union is small against the sum — meshes, and since a pixel's rays in consecutive passes differ by
    // the jitter only, even the benchmark's triangle fog at 16 passes per packet (1.8 x; one pass of an 8x8 patch: 3.0 x, which loses).
    // Which it is depends on scene, camera and resolution, so it is measured: every kProbeEvery-th injecting step — and the first after a
    // commit, a resize or a change of camera — a probe kernel on a side stream makes the camera rays of every 32nd group of pixels of
    // one injected pass and its companions itself and walks them as packets of the shape in use, writing nothing but
    //     U = (children the packet entered x its rays) / (children the rays' own box tests entered)
    // and how many child boxes a ray enters.  Packets are used while U < punion / 100 (profiles/r4u_packets.txt).  The totals come back
    // with the queue lengths k_trace reports (no synchronisation).  Either way the hits are the same bits.
    int tunePackets = 2;   // HR_TUNE="packets=0|1|2": never / always / by the probe (default)
    // The packet kernel is VALU-bound and leaves the texture addressers idle (busy 1.0 / 0.16); k_trace without the camera rays is the
    // other way round (0.70 / 0.94).  So a step's packet kernel runs BESIDE its k_trace, on a second stream (fork after the table copy,
    // join before the shading kernels), and k_trace leaves it room: 3 workgroups per CU instead of 5 when the camera rays are a good part
    // of the step's work, 4 when they are little (a step that injects few passes beside many in flight); c3 2100 -> 2390 Mrays/s at 128
    // passes, 2025 -> 2150 at 20 (profiles/r4v_corun.txt).
    // Only where k_trace IS bound by the addressers, i.e. where rays walk far: the probe also reports how many child boxes a camera ray
    // enters (c3 76, c5 75, c3d 162: +8..13 %; c2 35: no difference; terrain 10, c1 5: k_trace is VALU-bound itself there and loses 6 %).
    int tuneCorun = 1;       // HR_TUNE="corun=0|1|2": never (the packet kernel in front of k_trace on the group's stream) / by the probe / always
    int tuneCorunMin = 50;   // HR_TUNE="cmin=N": beside k_trace when a probed camera ray enters at least N child boxes
    int tunePacketSwizzle = 1; // HR_TUNE="pswz=0|1": k_raygen_packets deals whole 32x32 tiles to the XCDs (workgroup index -> XCD is round robin) instead of consecutive 16-pixel patches: a tile's part of the tree goes through ONE L2 (+0.3-0.7 % on c3 / c2 / c5, profiles/r5ak_packet_xcd.txt)
    int tuneProbeLog2 = -1;  // HR_TUNE="plog=N" (measurement only): the selector's probe walks packets of 2^N passes x 64 >> N pixels instead of the shape in use
    int tuneCorunBlocks = 0; // HR_TUNE="cblocks=N": fix k_trace's workgroups per CU in such a step (0: 3 or 4 by the step's mix)
    int tunePacketUnion = 220; // HR_TUNE="punion=N": packets while U < N / 100 (measured break-even ~2.3: terrain at 1.97 +7..11 %, c5 at 2.07 +3..4 %)
    bool packetsOn = false;
    uint32_t lastCameraCount = 0; // camera rays per pass behind the root cull, as last reported
    int probeCountdown = 0;              // injecting steps until the next probe
    bool probePending = false;
    unsigned long long probeStep = 0;    // step (of group 0) that carried the pending probe
    unsigned long long probeSeen[4] = {0, 0, 0, 0}; // totals of the report the last decision was taken on
    double lastOwnPerRay = 0.0;         // child boxes a probed camera ray entered: how long the scene's traversals are
    unsigned long long probeWaves = 0;   // waves of the pending probe: it is complete when the third total has grown by as many
    double lastUnion = 0.0;              // U of the last probe (HR_DEBUG_PIPE prints it)
    float probeCamera[21] = {0};         // fov, aspect, focus distance, aperture, view matrix, interactive mode of the probed pass
    unsigned long long *dProbe = nullptr; // three device counters the probe launches add to (never reset)
    hipStream_t probeStream = nullptr;   // the probe runs beside the pipeline: it makes its own camera rays and writes only the counters
    hipEvent_t evProbeA = nullptr, evProbeB = nullptr; // scene and tables as the group's stream sees them -> probe may start; probe done
    bool probeGuard = false;             // evProbeB has not been waited for yet (drainPipeline does: the scene may change afterwards)
    int tunePloc = 1, tunePlocRadius = 16; // HR_TUNE="ploc=0|1|2,plocr=N": tree builder (hr_build.hip: buildLBVH keeps the cheaper of the radix tree and PLOC)
    int tuneGuardPct = 125;   // HR_TUNE="guard=N": a refit whose boxes' area exceeds N % of the built tree's rebuilds instead (profiles/r3j_instanced_refit.txt)
    // persistent device arrays of the committed scene (grow-only capacities, reused across commits)
    GeomDev *dG = nullptr;
    size_t dGCap = 0;
    Tri *trisPrim = nullptr;  // prim-order triangles (input of a full build)
    size_t trisPrimCap = 0;
    size_t attrsCap = 0, attrsExtCap = 0;
    BuildResult tree{};       // nodes, leaf-order triangles, node boxes, prim -> slot map, level ranges
    uint32_t treeTris = 0;
    SceneConsts *dConsts = nullptr;
    SceneConsts *hConsts = nullptr; // pinned
    float builtAreaSum = 0.0f; // (sum of the node boxes' areas) / (sum of the triangles' areas) right after the last full build (refit quality reference)
    std::string cachePath;          // hr_scene_cache
    // pinned staging ring for mesh uploads
    char *stage[2] = {nullptr, nullptr};
    hipEvent_t stageEv[2] = {nullptr, nullptr};
    bool stageBusy[2] = {false, false};
    int stageTurn = 0;
    hr_scene_info info{};

    // scene (device); nodes / tris alias tree.nodes / tree.tris
    Node4 *nodes = nullptr;
    Tri *tris = nullptr;
    TriAttr *attrs = nullptr;
    TriAttrExt *attrsExt = nullptr;
    hr_material *dMaterials = nullptr;
    size_t dMaterialsCap = 0;
    TexDesc *dTextures = nullptr;
    size_t dTexturesCap = 0;
    float *dTexDensity = nullptr; // HR_TEXTURE_LOD_CONE: per-triangle level offset, rebuilt after every commit once the mode was used
    size_t texDensityCap = 0;
    bool texDensityStale = true;
    float2 *dSeq = nullptr, *dAperture = nullptr, *dSeqOffsets = nullptr;
    int nSeq = 0, seqLen = 0, nSeqOffsets = 0;
    SceneDev hScene{};
    SceneDev *dScene = nullptr;
    Stats *dStats = nullptr;
    uint32_t *dScratch = nullptr; // 8 words: ordered bounds etc.

    // optional per-kernel timing (HR_CTX_TIME_KERNELS)
    bool timeKernels = false;
    struct Timed {
        int kind;
        hipEvent_t e0, e1;
        bool e0Shared; // e0 is the e1 of the entry before (timeNext): one record between two kernels enqueued back to back
    };
    std::vector<Timed> pending;
    std::vector<hipEvent_t> eventPool;
    float kernelMs[HR_KERNEL_COUNT] = {0, 0, 0, 0};
    uint32_t kernelLaunches[HR_KERNEL_COUNT] = {0, 0, 0, 0};
    hipEvent_t getEvent()
    {
        hipEvent_t e = nullptr;
        if (!eventPool.empty()) {
            e = eventPool.back();
            eventPool.pop_back();
        } else {
            hipEventCreate(&e);
        }
        return e;
    }
    void timeBegin(int kind, hipStream_t st)
    {
        if (!timeKernels) return;
        Timed t{kind, getEvent(), getEvent(), false};
        hipEventRecord(t.e0, st);
        pending.push_back(t);
    }
    // The kernel timed last ends and the next one begins at ONE event (a record is a packet of several microseconds between the two)
    void timeNext(int kind, hipStream_t st)
    {
        if (!timeKernels) return;
        const hipEvent_t mid = pending.back().e1;
        hipEventRecord(mid, st);
        pending.push_back(Timed{kind, mid, getEvent(), true});
    }
    void timeEnd(hipStream_t st)
    {
        if (!timeKernels) return;
        hipEventRecord(pending.back().e1, st);
    }
    void drainTimes()
    {
        if (pending.empty()) return;
        for (int g = 0; g < kMaxGroups; ++g)
            if (groups[g].stream) hipStreamSynchronize(groups[g].stream);
        hipStreamSynchronize(stream);
        for (Timed &t : pending) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
                kernelMs[t.kind] += ms;
                kernelLaunches[t.kind] += 1;
            }
            if (!t.e0Shared) eventPool.push_back(t.e0);
            eventPool.push_back(t.e1);
        }
        pending.clear();
    }

    float *fb() const { return fbExternal ? fbExternal : fbInternal; }
    // tuning knobs (defaults measured on MI355X; HR_TUNE="tri=4,refill=8,blocks=6,depth=12,batch=2,groups=2" overrides for experiments)
    int tuneTri = 2, tuneRefill = 16, tuneBlocks = 5, tuneShadeBlocks = 4, tuneDepth = kMaxSlots, tuneBatch = 0, tuneFetchMax = 64, tuneFetchMin = 64, tuneStaticDeal = 256, tuneFetchPrimary = 128, tuneFetchGate = 8, tuneHeads = 5, tuneSlowMs = 4;
    LaunchCfg cfg(hipStream_t st) const { return LaunchCfg{st, numCUs, tuneBlocks, tuneShadeBlocks, collectStats, textureLodUsed, allLightsUsed, hasGlass, tunePacketSwizzle}; }
};

#define FAIL(ctx, code, msg)  \
    do {                      \
        (ctx)->err = (msg);   \
        return (code);        \
    } while (0)

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return HR_ERR_DEVICE;                                                                 \
        }                                                                                         \
    } while (0)

#define ENTER(ctx)                                   \
    if (!(ctx)) return HR_ERR_INVALID;               \
    HIP_TRY(ctx, hipSetDevice((ctx)->device))

