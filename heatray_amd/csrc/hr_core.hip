// hr_core.hip — host side of libhrcore: the C-ABI of include/hrcore.h over the HIP kernels.
//
// Owns device memory (scene, BVH, tables, ray queues, accumulation buffer) and sequences the kernels of
// a pass.  There is no CPU rendering path in this library: without a usable HIP device every entry point
// that needs one fails with HR_ERR_DEVICE.
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
    LaunchCfg cfg(hipStream_t st) const { return LaunchCfg{st, numCUs, tuneBlocks, tuneShadeBlocks, collectStats, textureLodUsed, allLightsUsed, hasGlass}; }
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
        get("groups=", c->tuneGroups), get("prio=", c->tunePrio), get("refit=", c->tuneRefit), get("sdeal=", c->tuneStaticDeal), get("guard=", c->tuneGuardPct), get("ploc=", c->tunePloc), get("packets=", c->tunePackets), get("corun=", c->tuneCorun), get("cmin=", c->tuneCorunMin), get("cblocks=", c->tuneCorunBlocks), get("punion=", c->tunePacketUnion), get("plocr=", c->tunePlocRadius), get("fprim=", c->tuneFetchPrimary), get("fgate=", c->tuneFetchGate), get("heads=", c->tuneHeads), get("slow=", c->tuneSlowMs), get("ovf=", c->tuneOverflowTest), get("sprobe=", c->tuneShadowProbe), get("tblk=", c->tuneTableKernel);
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
static const size_t kStageBytes = (size_t)16 << 20;

// host bytes -> device through the pinned ring: while the DMA of one half runs, the CPU fills the other
static int stagedUpload(hr_ctx *c, char *dst, const char *src, size_t bytes)
{
    for (int k = 0; k < 2; ++k) {
        if (!c->stage[k]) {
            HIP_TRY(c, hipHostMalloc((void **)&c->stage[k], kStageBytes, hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&c->stageEv[k], hipEventDisableTiming));
        }
    }
    for (size_t at = 0; at < bytes; at += kStageBytes) {
        const size_t len = bytes - at < kStageBytes ? bytes - at : kStageBytes;
        const int k = c->stageTurn++ & 1;
        if (c->stageBusy[k]) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
        std::memcpy(c->stage[k], src + at, len);
        HIP_TRY(c, hipMemcpyAsync(dst + at, c->stage[k], len, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stageEv[k], c->stream));
        c->stageBusy[k] = true;
    }
    return HR_OK;
}

int hr_geom_add(hr_ctx *c, const hr_mesh_desc *d, hr_geom_id *out)
{
    ENTER(c);
    if (!d || !d->positions || !d->normals || !d->indices || d->n_vertices <= 0 || d->n_indices < 0)
        FAIL(c, HR_ERR_INVALID, "mesh needs positions, normals, indices");
    if (d->mode != HR_TRIANGLES && d->mode != HR_TRIANGLE_STRIP) FAIL(c, HR_ERR_INVALID, "unsupported draw mode");
    {
        uint32_t worst = 0; // (a plain max reduction: vectorises)
        for (int i = 0; i < d->n_indices; ++i) worst = d->indices[i] > worst ? d->indices[i] : worst;
        if (d->n_indices > 0 && worst >= (uint32_t)d->n_vertices) FAIL(c, HR_ERR_INVALID, "index out of range");
    }
    {
        // positions must be finite: a NaN box has no order, and the tree builders' progress arguments (and every slab test) assume one
        const int sb = d->position_stride == 0 ? 12 : d->position_stride;
        if (sb < 12) FAIL(c, HR_ERR_INVALID, "attribute stride smaller than the attribute");
        float worst = 0.0f;
        bool nan = false;
        for (int i = 0; i < d->n_vertices; ++i) {
            float p[3];
            std::memcpy(p, (const char *)d->positions + (size_t)i * (size_t)sb, 12);
            const float m = std::fmax(std::fabs(p[0]), std::fmax(std::fabs(p[1]), std::fabs(p[2]))); // (fmax drops a NaN operand: checked apart)
            worst = m > worst ? m : worst;
            nan = nan || p[0] != p[0] || p[1] != p[1] || p[2] != p[2];
        }
        if (nan || !(worst <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "vertex positions must be finite");
        for (int k = 0; k < 16; ++k)
            if (!(std::fabs(d->world_from_entity[k]) <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "world_from_entity must be finite");
    }
    const float *src[6] = {d->positions, d->normals, d->uvs, d->tangents, d->bitangents, d->colors};
    const int32_t strideB[6] = {d->position_stride, d->normal_stride, d->uv_stride, d->tangent_stride, d->bitangent_stride, d->color_stride};
    const int comps[6] = {3, 3, 2, 3, 3, 3};
    Geom g;
    g.alive = true;
    g.nVerts = d->n_vertices;
    g.nIdx = (uint32_t)d->n_indices;
    g.mode = d->mode;
    std::memcpy(g.world, d->world_from_entity, sizeof(g.world));
    g.frontFaceCW = d->front_face_cw, g.isOccluder = d->is_occluder, g.material = d->material_id;
    // layout of the device block: every attribute as the caller holds it (its stride included), then the indices
    size_t bytesOf[6] = {0, 0, 0, 0, 0, 0}, total = 0;
    std::vector<float> tight[6]; // only for strides that are not a multiple of four bytes (re-packed on the host)
    // One interleaved vertex buffer (every attribute a pointer into the same array of `stride`-byte vertices, as glTF loaders hand
    // them over) is uploaded ONCE and addressed with per-attribute offsets; uploading it once per attribute with its full stride
    // cost 3-6 x the device memory and PCIe traffic.
    const char *ilo = nullptr, *ihi = nullptr;
    int isb = 0;
    bool interleaved = true;
    int nAttr = 0;
    for (int a = 0; a < 6; ++a) {
        if (!src[a]) continue;
        ++nAttr;
        const char *p0 = (const char *)src[a], *p1 = p0 + comps[a] * sizeof(float);
        if (strideB[a] <= 0 || strideB[a] % 4 != 0 || (isb != 0 && strideB[a] != isb)) interleaved = false;
        isb = strideB[a];
        ilo = (!ilo || p0 < ilo) ? p0 : ilo;
        ihi = (!ihi || p1 > ihi) ? p1 : ihi;
    }
    interleaved = interleaved && nAttr >= 2 && (size_t)(ihi - ilo) <= (size_t)isb;
    if (interleaved) {
        const size_t span = (size_t)(g.nVerts - 1) * (size_t)isb + (size_t)(ihi - ilo);
        for (int a = 0; a < 6; ++a) {
            if (!src[a]) continue;
            g.has[a] = true;
            g.stride[a] = isb / 4;
            g.off[a] = (size_t)((const char *)src[a] - ilo);
            bytesOf[a] = 0;
        }
        total = (span + 15) & ~(size_t)15;
    } else {
        for (int a = 0; a < 6; ++a) {
            if (!src[a]) continue;
            g.has[a] = true;
            int sb = strideB[a] == 0 ? comps[a] * (int)sizeof(float) : strideB[a];
            if (sb < comps[a] * (int)sizeof(float) && sb != 0) FAIL(c, HR_ERR_INVALID, "attribute stride smaller than the attribute");
            if (sb % 4 != 0) {
                tight[a].resize((size_t)g.nVerts * comps[a]);
                for (int i = 0; i < g.nVerts; ++i) std::memcpy(&tight[a][(size_t)i * comps[a]], (const char *)src[a] + (size_t)i * sb, comps[a] * sizeof(float));
                sb = comps[a] * (int)sizeof(float);
            }
            g.stride[a] = sb / 4;
            bytesOf[a] = (size_t)(g.nVerts - 1) * sb + comps[a] * sizeof(float);
            g.off[a] = total;
            total += (bytesOf[a] + 15) & ~(size_t)15;
        }
    }
    g.off[6] = total;
    total += ((size_t)g.nIdx * 4 + 15) & ~(size_t)15;
    g.blockBytes = total;
    g.dBlock = c->meshAlloc(total ? total : 16, &g.chunk);
    if (!g.dBlock) FAIL(c, HR_ERR_DEVICE, "out of device memory for a mesh block");
    // (the 16-byte alignment padding behind each range is never uploaded, yet the tree cache's content hash covers the whole block:
    // recycled device memory there made the key differ from run to run)
    int rc = HR_OK;
    if (hipMemsetAsync(g.dBlock, 0, total ? total : 16, c->stream) != hipSuccess) {
        c->meshRelease(g.chunk); // (every error return behind meshAlloc gives the block back)
        FAIL(c, HR_ERR_DEVICE, "hipMemsetAsync of a mesh block failed");
    }
    if (interleaved) {
        rc = stagedUpload(c, g.dBlock, ilo, (size_t)(g.nVerts - 1) * (size_t)isb + (size_t)(ihi - ilo));
    } else {
        for (int a = 0; a < 6 && rc == HR_OK; ++a)
            if (g.has[a]) rc = stagedUpload(c, g.dBlock + g.off[a], tight[a].empty() ? (const char *)src[a] : (const char *)tight[a].data(), bytesOf[a]);
    }
    if (rc == HR_OK && g.nIdx) rc = stagedUpload(c, g.dBlock + g.off[6], (const char *)d->indices, (size_t)g.nIdx * 4);
    if (rc != HR_OK) {
        c->meshRelease(g.chunk);
        return rc;
    }
    c->geoms.push_back(g);
    c->committed = false, c->topologyDirty = true;
    if (out) *out = (hr_geom_id)c->geoms.size() - 1;
    return HR_OK;
}

int hr_geom_remove(hr_ctx *c, hr_geom_id id)
{
    ENTER(c);
    if (id < 0 || id >= (int)c->geoms.size() || !c->geoms[id].alive) FAIL(c, HR_ERR_INVALID, "bad geom id");
    for (int k = 0; k < 2; ++k) // an upload of this mesh may still be in flight (nothing to wait for when the staging ring is idle)
        if (c->stageBusy[k] && hipEventQuery(c->stageEv[k]) != hipSuccess) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
    c->meshRelease(c->geoms[id].chunk);
    c->geoms[id] = Geom();
    c->committed = false, c->topologyDirty = true;
    return HR_OK;
}

int hr_geom_set_transform(hr_ctx *c, hr_geom_id id, const float m[16])
{
    ENTER(c);
    if (id < 0 || id >= (int)c->geoms.size() || !c->geoms[id].alive || !m) FAIL(c, HR_ERR_INVALID, "bad geom id");
    for (int k = 0; k < 16; ++k)
        if (!(std::fabs(m[k]) <= 3.0e37f)) FAIL(c, HR_ERR_INVALID, "world_from_entity must be finite");
    std::memcpy(c->geoms[id].world, m, 16 * sizeof(float));
    c->committed = false, c->transformDirty = true;
    return HR_OK;
}

int hr_scene_clear(hr_ctx *c)
{
    ENTER(c);
    for (int k = 0; k < 2; ++k)
        if (c->stageBusy[k] && hipEventQuery(c->stageEv[k]) != hipSuccess) HIP_TRY(c, hipEventSynchronize(c->stageEv[k]));
    c->meshReleaseAll();
    c->geoms.clear();
    c->committed = false, c->topologyDirty = true;
    return HR_OK;
}

// ---- tree cache file (hr_scene_cache): header + nodes + node boxes + prim -> slot map
namespace {
struct CacheHeader {
    char magic[8];
    uint32_t version, nodeBytes;
    unsigned long long key;
    uint32_t nTris, nNodes, levels, rootLeafCount, triSlots, builder; // builder: which binary tree was collapsed (BuildResult::builder)
    uint32_t levelStart[kMaxLevels + 1];
    float costRadix, costPloc;     // the candidates' costs as hr_scene_info reports them
    unsigned long long payloadSum; // checksum of everything behind the header (the key covers the SCENE, not the file)
};
const uint32_t kCacheVersion = 4; // 4: two candidate builders (round 4) — the header says which tree the file holds, the key which builder options made it

// 64-bit checksum of the payload, eight bytes at a time (the files are tens to hundreds of MB)
unsigned long long payloadChecksum(const char *p, size_t bytes)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        unsigned long long w;
        std::memcpy(&w, p + i, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h ^= h >> 29;
    }
    unsigned long long tail = 0;
    if (i < bytes) std::memcpy(&tail, p + i, bytes - i);
    h = (h ^ tail ^ (unsigned long long)bytes) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 32);
}

// Everything the kernels index with comes out of the file: child ranges, leaf triangle slots, the prim -> slot map, the level table.
// The checksum only catches accidental damage (it is not cryptographic: the cache directory is trusted like the scene files are), so
// every index is range-checked before the arrays reach the device (an out-of-range child or slot is a GPU fault or a hang in
// k_refit4 / k_trace, not a wrong pixel), the node count is bounded before anything is allocated, and every inner child must lie in
// the next level's range, which proves the depth the stack check relies on.
bool cachedTreeIsSane(const CacheHeader &h, const char *nodesBytes, const uint32_t *slotOfPrim)
{
    if (h.triSlots < h.nTris || h.triSlots >= (1u << 28) || h.rootLeafCount > 4u || h.levels > (uint32_t)kMaxLevels) return false;
    if (h.nNodes >= (1u << 26) || h.nNodes > h.nTris) return false; // (every 4-wide node stands for one binary inner node)
    if (h.rootLeafCount > 0 && h.rootLeafCount > h.triSlots) return false;
    if (h.levelStart[0] != 0u) return false;
    for (uint32_t l = 0; l < h.levels; ++l)
        if (h.levelStart[l + 1] < h.levelStart[l] || h.levelStart[l + 1] > h.nNodes) return false;
    if (h.levels > 0 && h.levelStart[h.levels] != h.nNodes) return false;
    for (uint32_t i = 0; i < h.nTris; ++i)
        if (slotOfPrim[i] >= h.triSlots) return false;
    uint32_t level = 0;
    for (uint32_t i = 0; i < h.nNodes; ++i) {
        while (level + 1 < h.levels && i >= h.levelStart[level + 1]) ++level;
        Node4 n;
        std::memcpy(&n, nodesBytes + (size_t)i * sizeof(Node4), sizeof(Node4));
        uint32_t meta;
        std::memcpy(&meta, &n.a.w, 4);
        const uint32_t nInner = (meta >> 24) & 7u, nValid = meta >> 27;
        if (nValid > 4u || nInner > nValid) return false;
        if (nInner > 0) {
            // inner children are nodes innerBase .. innerBase + nInner - 1 and lie in the NEXT level's index range (breadth-first
            // allocation): that proves the depth the header claims, which bounds the traversal stack (3 entries per level)
            const uint32_t base = n.c.z;
            if (level + 1 >= h.levels) return false;
            if (base < h.levelStart[level + 1] || base >= h.levelStart[level + 2] || nInner > h.levelStart[level + 2] - base) return false;
        }
        for (uint32_t j = nInner; j < nValid; ++j) { // leaf child j is the triangle ~(leafKey + j)
            const uint32_t slot = ~(n.c.w + j);
            if (slot >= h.triSlots) return false;
        }
    }
    return true;
}
} // namespace

// digest of everything the tree depends on: geometry bytes (hashed on the device), transforms, modes, strides
static int sceneKey(hr_ctx *c, unsigned long long *key)
{
    unsigned long long *dKey = nullptr;
    HIP_TRY(c, hipMalloc(&dKey, 8));
    hipError_t e = hipMemsetAsync(dKey, 0, 8, c->stream);
    unsigned long long host = 0xC0FFEE1234ull;
    auto mix = [&](const void *p, size_t bytes) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < bytes; ++i) host = (host ^ b[i]) * 0x100000001B3ull; // FNV-1a over the small host-side fields
    };
    unsigned long long seed = 1;
    for (const Geom &g : c->geoms) {
        if (!g.alive || g.nTris() == 0) continue;
        mix(&g.nVerts, sizeof(g.nVerts)), mix(&g.nIdx, sizeof(g.nIdx)), mix(&g.mode, sizeof(g.mode)), mix(g.world, sizeof(g.world));
        mix(g.stride, sizeof(g.stride)), mix(g.off, sizeof(g.off)), mix(g.has, sizeof(g.has));
        if (e == hipSuccess) launchHashWords(c->stream, g.dBlock, g.blockBytes / 4, seed++, dKey);
    }
    unsigned long long dev = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&dev, dKey, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dKey);
    HIP_TRY(c, e);
    // (which tree a build produces also depends on the builder options: a file made with other ones is another scene's as far as the cache goes)
    mix(&c->tunePloc, sizeof(c->tunePloc)), mix(&c->tunePlocRadius, sizeof(c->tunePlocRadius));
    *key = host ^ (dev * 0x9E3779B97F4A7C15ull);
    return HR_OK;
}

// read the tree of this scene from the cache file; false: no usable file (the caller builds)
static bool loadTree(hr_ctx *c, unsigned long long key, uint32_t nTris, BuildResult *out)
{
    FILE *f = fopen(c->cachePath.c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "HRBVHTR", 8) == 0 && h.version == kCacheVersion &&
              h.nodeBytes == sizeof(Node4) && h.key == key && h.nTris == nTris && h.nNodes > 0 && h.nNodes <= nTris && h.nNodes < (1u << 26) &&
              h.levels <= (uint32_t)kMaxLevels;
    std::vector<char> buf;
    BuildResult br{};
    if (ok) {
        const size_t nb = (size_t)h.nNodes * sizeof(Node4), bb = (size_t)h.nNodes * sizeof(Box6), sb = (size_t)nTris * 4;
        buf.resize(nb + bb + sb);
        ok = fread(buf.data(), 1, buf.size(), f) == buf.size() && fgetc(f) == EOF; // exactly the payload: nothing missing, nothing appended
        ok = ok && payloadChecksum(buf.data(), buf.size()) == h.payloadSum;
        ok = ok && cachedTreeIsSane(h, buf.data(), reinterpret_cast<const uint32_t *>(buf.data() + nb + bb));
        if (ok) {
            ok = hipMalloc(&br.nodes, nb) == hipSuccess && hipMalloc(&br.nodes32, (size_t)h.nNodes * sizeof(Node32)) == hipSuccess && hipMalloc(&br.leafKeys, (size_t)h.nNodes * sizeof(int)) == hipSuccess && hipMalloc(&br.nodeBox, bb) == hipSuccess && hipMalloc(&br.slotOfPrim, sb) == hipSuccess &&
                 hipMalloc(&br.tris, sizeof(Tri) * (size_t)h.triSlots) == hipSuccess;
            ok = ok && hipMemcpy(br.nodes, buf.data(), nb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(br.nodeBox, buf.data() + nb, bb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(br.slotOfPrim, buf.data() + nb + bb, sb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(br.tris, 0xFF, sizeof(Tri) * (size_t)h.triSlots) == hipSuccess;
        }
    }
    fclose(f);
    if (!ok) {
        hipFree(br.nodes), hipFree(br.nodes32), hipFree(br.leafKeys), hipFree(br.nodeBox), hipFree(br.slotOfPrim), hipFree(br.tris);
        return false;
    }
    br.nNodes = (int32_t)h.nNodes, br.levels = (int32_t)h.levels, br.rootLeafCount = (int32_t)h.rootLeafCount, br.triSlots = h.triSlots;
    br.builder = h.builder == 1u ? 1 : 0, br.costRadix = h.costRadix, br.costPloc = h.costPloc;
    std::memcpy(br.levelStart, h.levelStart, sizeof(br.levelStart));
    *out = br;
    return true;
}

static void saveTree(hr_ctx *c, unsigned long long key, uint32_t nTris, const BuildResult &br)
{
    if (br.nNodes <= 0) return;
    const size_t nb = (size_t)br.nNodes * sizeof(Node4), bb = (size_t)br.nNodes * sizeof(Box6), sb = (size_t)nTris * 4;
    std::vector<char> buf(nb + bb + sb);
    if (hipMemcpy(buf.data(), br.nodes, nb, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(buf.data() + nb, br.nodeBox, bb, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(buf.data() + nb + bb, br.slotOfPrim, sb, hipMemcpyDeviceToHost) != hipSuccess)
        return;
    CacheHeader h{};
    std::memcpy(h.magic, "HRBVHTR", 8);
    h.version = kCacheVersion, h.nodeBytes = sizeof(Node4), h.key = key, h.nTris = nTris, h.nNodes = (uint32_t)br.nNodes, h.levels = (uint32_t)br.levels;
    h.rootLeafCount = (uint32_t)br.rootLeafCount, h.triSlots = br.triSlots;
    h.builder = (uint32_t)br.builder, h.costRadix = br.costRadix, h.costPloc = br.costPloc;
    std::memcpy(h.levelStart, br.levelStart, sizeof(h.levelStart));
    h.payloadSum = payloadChecksum(buf.data(), buf.size());
    const std::string tmp = c->cachePath + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (ok)
        rename(tmp.c_str(), c->cachePath.c_str());
    else
        remove(tmp.c_str());
}

// Device temporaries and timing events of one commit: released on every exit path.
namespace {
struct CommitScratch {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BuildResult br{};
    bool keepBuild = false;
    ~CommitScratch()
    {
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        if (!keepBuild) hipFree(br.nodes), hipFree(br.nodes32), hipFree(br.leafKeys), hipFree(br.tris), hipFree(br.nodeBox), hipFree(br.slotOfPrim);
    }
};
} // namespace

int hr_scene_commit(hr_ctx *c)
{
    ENTER(c);
    QUIESCE(c);
    // until this call succeeds there is no scene to render: a failed re-commit must not leave `committed` set over stale arrays
    c->committed = false, c->sceneDirty = true;
    CommitScratch cs;
    HIP_TRY(c, hipEventCreate(&cs.e0));
    HIP_TRY(c, hipEventCreate(&cs.e1));
    HIP_TRY(c, hipEventRecord(cs.e0, c->stream));
    if (!c->dConsts) {
        HIP_TRY(c, hipMalloc(&c->dConsts, sizeof(SceneConsts)));
        HIP_TRY(c, hipHostMalloc((void **)&c->hConsts, sizeof(SceneConsts), hipHostMallocDefault));
    }
    // ---- descriptors of the live geometries (their data is on the device already: hr_geom_add)
    std::vector<GeomDev> gd;
    uint32_t nTris = 0;
    bool anyExt = false;
    for (const Geom &g : c->geoms) {
        if (!g.alive || g.nTris() == 0) continue;
        GeomDev d{};
        const float *at[6];
        for (int a = 0; a < 6; ++a) at[a] = g.has[a] ? reinterpret_cast<const float *>(g.dBlock + g.off[a]) : nullptr;
        d.pos = at[0], d.nrm = at[1], d.uv = at[2], d.tan = at[3], d.bit = at[4], d.col = at[5];
        d.posStride = g.stride[0], d.nrmStride = g.stride[1], d.uvStride = g.stride[2], d.tanStride = g.stride[3], d.bitStride = g.stride[4],
        d.colStride = g.stride[5];
        d.idx = reinterpret_cast<const uint32_t *>(g.dBlock + g.off[6]);
        d.triOffset = nTris, d.nTris = g.nTris(), d.strip = g.mode == HR_TRIANGLE_STRIP;
        d.flags = (g.frontFaceCW ? TF_FRONT_CW : 0u) | (g.isOccluder ? 0u : TF_NON_OCCLUDER) | (g.has[2] ? TF_HAS_UV : 0u) |
                  ((g.has[3] && g.has[4]) ? TF_HAS_TANGENTS : 0u) | (g.has[5] ? TF_HAS_COLORS : 0u);
        d.material = (uint32_t)g.material;
        std::memcpy(d.world, g.world, sizeof(d.world));
        if (d.flags & (TF_HAS_TANGENTS | TF_HAS_COLORS)) anyExt = true;
        nTris += d.nTris;
        gd.push_back(d);
    }
    std::memset(&c->info, 0, sizeof(c->info));
    c->hScene.nodes = nullptr, c->hScene.nodes32 = nullptr, c->hScene.leafKeys = nullptr, c->hScene.tris = nullptr, c->hScene.attrs = nullptr, c->hScene.attrsExt = nullptr;
    c->hScene.nTris = 0, c->hScene.nNodes = 0, c->hScene.rootLeafCount = 0, c->hScene.rayEps = 0.0f, c->hScene.hitPad = 0.0f;
    if (nTris == 0) {
        freeTree(c);
    } else {
        int rc = ensureCap(c, &c->dG, &c->dGCap, gd.size());
        if (rc == HR_OK) rc = ensureCap(c, &c->attrs, &c->attrsCap, (size_t)nTris);
        if (rc == HR_OK && anyExt) rc = ensureCap(c, &c->attrsExt, &c->attrsExtCap, (size_t)nTris);
        if (rc != HR_OK) return rc;
        TriAttrExt *ext = anyExt ? c->attrsExt : nullptr;
        HIP_TRY(c, hipMemcpyAsync(c->dG, gd.data(), gd.size() * sizeof(GeomDev), hipMemcpyHostToDevice, c->stream));
        // A commit after transform edits only keeps the tree's topology: triangles are re-assembled straight into their leaf
        // slots and every level is refitted bottom-up.  No allocation, no host round trip before the last kernel.
        bool cacheHit = false;
        bool refit = c->tuneRefit && !c->topologyDirty && c->tree.nodes && c->treeTris == nTris && c->tree.rootLeafCount == 0;
        if (refit) {
            launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, c->tree.tris, c->tree.slotOfPrim, c->attrs, ext, c->dScratch);
            launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
            refitLBVH(c->stream, c->tree, nTris, c->dConsts);
            encodeNodes32(c->stream, c->tree, c->dConsts, nullptr); // (k_trace's copy of the nodes: every frame is re-encoded, the grid moves with the bounds)
            launchTriAreaSum(c->stream, c->tree.tris, c->tree.triSlots, c->dConsts);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // A refitted tree is only as good as its topology still fits the geometry: when the boxes have grown (an object moved
            // through or away from its neighbours; a rotation inflates axis-aligned boxes), rebuild.  Measured on an instanced scene
            // (16 objects, one travelling through the others, tools/r3_instanced_refit.py, profiles/r3j_instanced_refit.txt): the
            // refitted tree is 1.5 % slower than a fresh build at 1.11 x the built tree's box area, 4-8 % at 1.2-1.3 x, 8 % at 1.5 x,
            // 8-9 % when the object is flung away.  Box area is taken relative to the triangles' own area, which rigid motion leaves
            // alone and a scaling of the whole scene scales alike.  (Round 2 compared area / diagonal^2 with a threshold of 4: a flung
            // object grows the diagonal too, so that guard never fired.)  A rebuild of 1 M triangles costs 4.7 ms, a refit 0.3 ms.
            const SceneConsts &k = *c->hConsts;
            const float nowQ = k.triAreaSum > 0.0f ? k.areaSum / k.triAreaSum : 0.0f;
            if (c->builtAreaSum > 0.0f && nowQ > 0.01f * (float)c->tuneGuardPct * c->builtAreaSum) refit = false;
        }
        if (!refit) {
            rc = ensureCap(c, &c->trisPrim, &c->trisPrimCap, (size_t)nTris);
            if (rc != HR_OK) return rc;
            launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, c->trisPrim, nullptr, c->attrs, ext, c->dScratch);
            launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const SceneConsts k = *c->hConsts;
            unsigned long long key = 0;
            bool fromCache = false;
            if (!c->cachePath.empty()) {
                rc = sceneKey(c, &key);
                if (rc != HR_OK) return rc;
                fromCache = loadTree(c, key, nTris, &cs.br);
            }
            if (fromCache) { // the tree is this scene's: only the triangles have to be put into their leaf slots
                launchAssemble(c->stream, c->dG, (int)gd.size(), nTris, cs.br.tris, cs.br.slotOfPrim, c->attrs, ext, c->dScratch);
                launchSceneConsts(c->stream, c->dScratch, c->dConsts, nullptr);
                cacheHit = true;
            } else {
                const BuildOptions bo{c->tunePloc, c->tunePlocRadius, (kStackLDS + kStackOvf) / 3};
                const int brc = buildLBVH(c->stream, c->trisPrim, nTris, k.lo, k.hi, k.pad, c->dConsts, &cs.br, bo);
                if (brc != 0) FAIL(c, HR_ERR_DEVICE, brc == 3 ? "LBVH refit did not reach the root" : "LBVH build failed");
                if (!c->cachePath.empty()) saveTree(c, key, nTris, cs.br);
            }
            encodeNodes32(c->stream, cs.br, c->dConsts, nullptr); // (built or read from the cache: k_trace's 32-byte copy of the nodes)
            // the traversal stack holds at most 3 entries per level of inner nodes (hr_trace.h)
            if (3 * cs.br.levels > kStackLDS + kStackOvf) FAIL(c, HR_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
            if (cs.br.triSlots >= (1u << 28)) FAIL(c, HR_ERR_UNSUPPORTED, "scene too large: triangle slots do not fit a 28-bit leaf reference");
            if ((unsigned long long)cs.br.nNodes >= (1ull << 25)) FAIL(c, HR_ERR_UNSUPPORTED, "scene too large: a 32-byte node holds its children's base index in 25 bits (2^25 nodes, ~95 M triangles)");
            freeTree(c);
            c->tree = cs.br, cs.keepBuild = true;
            c->treeTris = nTris;
            launchAreaSum(c->stream, c->tree.nodeBox, (uint32_t)c->tree.nNodes, c->dConsts);
            launchTriAreaSum(c->stream, c->tree.tris, c->tree.triSlots, c->dConsts);
            HIP_TRY(c, hipMemcpyAsync(c->hConsts, c->dConsts, sizeof(SceneConsts), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->builtAreaSum = c->hConsts->triAreaSum > 0.0f ? c->hConsts->areaSum / c->hConsts->triAreaSum : 0.0f;
        }
        const SceneConsts &k = *c->hConsts;
        c->nodes = c->tree.nodes, c->tris = c->tree.tris;
        c->hScene.nodes = c->nodes, c->hScene.nodes32 = c->tree.nodes32, c->hScene.leafKeys = c->tree.leafKeys, c->hScene.tris = c->tris, c->hScene.attrs = c->attrs, c->hScene.attrsExt = ext;
        gridOf(k, c->hScene.gridLo, c->hScene.gridCell, c->hScene.gridCellExp);
        c->hScene.nTris = (int)nTris, c->hScene.nNodes = c->tree.nNodes, c->hScene.rootLeafCount = c->tree.rootLeafCount;
        c->hScene.rayEps = k.eps; // 1e-4 |diagonal|, SURVEY §8a a6
        c->hScene.hitPad = 0.5f * k.pad; // (hr_trace.h: hitInTriBox)
        for (int q = 0; q < 3; ++q) c->info.aabb_min[q] = k.lo[q], c->info.aabb_max[q] = k.hi[q];
        c->info.n_triangles = nTris, c->info.n_nodes = (uint64_t)c->tree.nNodes, c->info.ray_epsilon = k.eps;
        c->info.bvh_levels = (uint32_t)c->tree.levels;
        c->info.refitted = refit ? 1u : (cacheHit ? 2u : 0u);
        c->info.box_area_ratio = (c->builtAreaSum > 0.0f && k.triAreaSum > 0.0f) ? (k.areaSum / k.triAreaSum) / c->builtAreaSum : 0.0f;
        c->info.builder = (uint32_t)c->tree.builder, c->info.cost_radix = c->tree.costRadix, c->info.cost_ploc = c->tree.costPloc;
    }
    HIP_TRY(c, hipEventRecord(cs.e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(cs.e1));
    hipEventElapsedTime(&c->info.build_ms, cs.e0, cs.e1);
    c->committed = true;
    std::memset(c->stageSeen, 0, sizeof(c->stageSeen)); // (memory budget: another scene, other queue lengths)
    c->probeCountdown = 0; // (packet selector: another tree)
    c->sceneDirty = true;
    c->texDensityStale = true;
    c->topologyDirty = false, c->transformDirty = false;
    return HR_OK;
}

int hr_scene_cache(hr_ctx *c, const char *path)
{
    ENTER(c);
    c->cachePath = path ? path : "";
    return HR_OK;
}

int hr_scene_get_info(hr_ctx *c, hr_scene_info *out)
{
    ENTER(c);
    if (!c->committed || !out) FAIL(c, HR_ERR_INVALID, "scene not committed");
    *out = c->info;
    return HR_OK;
}

// --------------------------------------------------------------------------------------- textures
int hr_texture_create(hr_ctx *c, const hr_texture_desc *d, const void *pixels, hr_tex_id *out)
{
    ENTER(c);
    if (!d || !pixels || d->width <= 0 || d->height <= 0 || (d->channels != 1 && d->channels != 3 && d->channels != 4))
        FAIL(c, HR_ERR_INVALID, "bad texture descriptor");
    const size_t n = (size_t)d->width * d->height * d->channels;
    if (d->dtype != HR_TEX_U8 && d->dtype != HR_TEX_F32) FAIL(c, HR_ERR_INVALID, "bad texture dtype");
    // 8-bit data stays 8-bit in HBM (a quarter of the footprint and of the bytes per texel fetched; the sampler normalises
    // float(byte) / 255.0f on fetch, the conversion the reference's loader would otherwise leave to the RL texture unit)
    const size_t bytes = n * (d->dtype == HR_TEX_U8 ? 1 : sizeof(float));
    Texture t;
    HIP_TRY(c, hipMalloc(&t.dpx, bytes));
    HIP_TRY(c, hipMemcpy(t.dpx, pixels, bytes, hipMemcpyHostToDevice));
    t.desc = TexDesc{t.dpx, d->width, d->height, d->channels, d->wrap_s, d->wrap_t, d->filter, d->dtype, 0, nullptr, 0.0f, 0};
    t.alive = true;
    c->textures.push_back(t);
    c->sceneDirty = true;
    if (out) *out = (hr_tex_id)c->textures.size() - 1;
    return HR_OK;
}

int hr_texture_destroy(hr_ctx *c, hr_tex_id id)
{
    ENTER(c);
    if (id < 0 || id >= (int)c->textures.size() || !c->textures[id].alive) FAIL(c, HR_ERR_INVALID, "bad texture id");
    QUIESCE(c);
    hipFree(c->textures[id].dpx), hipFree(c->textures[id].dmips);
    c->textures[id] = Texture();
    if (c->envTex == id) c->envTex = -2, c->envW = c->envH = 0;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_material_set(hr_ctx *c, int32_t id, const hr_material *m)
{
    ENTER(c);
    if (id < 0 || id > (1 << 20) || !m) FAIL(c, HR_ERR_INVALID, "bad material id");
    if ((int)c->materials.size() <= id) {
        hr_material none{};
        none.type = -1;
        c->materials.resize(id + 1, none);
    }
    c->materials[id] = *m;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_lights_set(hr_ctx *c, const hr_lights *l)
{
    ENTER(c);
    if (!l || l->n_directional < 0 || l->n_directional > HR_MAX_DIRECTIONAL_LIGHTS || l->n_point < 0 || l->n_point > HR_MAX_POINT_LIGHTS ||
        l->n_spot < 0 || l->n_spot > HR_MAX_SPOT_LIGHTS)
        FAIL(c, HR_ERR_INVALID, "bad light block");
    c->lights = *l;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_interactive_blocks_set(hr_ctx *c, const int32_t *coords, int32_t nx, int32_t ny)
{
    ENTER(c);
    if (!coords) {
        c->blockNx = c->blockNy = 0;
    } else {
        if (nx <= 0 || ny <= 0 || nx * ny > 16) FAIL(c, HR_ERR_INVALID, "block table: nx*ny must be 1..16");
        for (int i = 0; i < nx * ny; ++i)
            if (coords[2 * i] < 0 || coords[2 * i + 1] < 0) FAIL(c, HR_ERR_INVALID, "block table: negative coordinate");
        c->blockNx = nx, c->blockNy = ny;
        std::memcpy(c->blockCoords, coords, sizeof(int32_t) * 2 * (size_t)(nx * ny));
    }
    c->sceneDirty = true;
    return HR_OK;
}

// ----------------------------------------------------------------------------------- sample tables
static int setTable(hr_ctx *c, float2 **dst, const float *src, size_t n)
{
    QUIESCE(c);
    hipFree(*dst);
    *dst = nullptr;
    HIP_TRY(c, hipMalloc(dst, n * sizeof(float2)));
    if (src) HIP_TRY(c, hipMemcpy(*dst, src, n * sizeof(float2), hipMemcpyHostToDevice));
    return HR_OK;
}

int hr_sequences_set(hr_ctx *c, const float *seq, const float *ap, int32_t nSeq, int32_t len)
{
    ENTER(c);
    if (!seq || !ap || nSeq <= 0 || nSeq > 255 || len <= 0) FAIL(c, HR_ERR_INVALID, "bad sequence table");
    int rc = setTable(c, &c->dSeq, seq, (size_t)nSeq * len);
    if (rc) return rc;
    rc = setTable(c, &c->dAperture, ap, (size_t)nSeq * len);
    if (rc) return rc;
    c->nSeq = nSeq, c->seqLen = len;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_seq_offsets_set(hr_ctx *c, const float *off, int32_t n)
{
    ENTER(c);
    if (!off || n <= 0) FAIL(c, HR_ERR_INVALID, "bad offsets table");
    int rc = setTable(c, &c->dSeqOffsets, off, (size_t)n);
    if (rc) return rc;
    c->nSeqOffsets = n;
    c->sceneDirty = true;
    return HR_OK;
}

// radialSobol's disk mapping (Random.h:272-287).  The reference evaluates it with the C library's sqrtf / cosf / sinf, whose
// last bit is libm-specific; it is therefore done on the host, with the same library the application itself would use, on the
// device-generated Sobol points (which are bit-exact): the aperture tables then equal the reference's bit for bit.
// (16 x maxRenderPasses points at initialisation time.)
static void radialOnHost(float2 *p, size_t count)
{
    const float two_pi = 6.28318530717958647692f;
    for (size_t i = 0; i < count; ++i) {
        const float s = p[i].x, t = p[i].y;
        const float sqrt_t = sqrtf(t);
        const float two_pi_s = two_pi * s;
        float x = sqrt_t * cosf(two_pi_s);
        float y = sqrt_t * sinf(two_pi_s);
        x = (x + 1.0f) * 0.5f;
        y = (y + 1.0f) * 0.5f;
        p[i] = make_float2(x, y);
    }
}

int hr_qmc_generate(hr_ctx *c, int32_t mode, uint32_t seqIndex, uint32_t count, int32_t radial, float *out)
{
    ENTER(c);
    if (mode != HR_SAMPLE_SOBOL && mode != HR_SAMPLE_HALTON && mode != HR_SAMPLE_HAMMERSLEY)
        FAIL(c, HR_ERR_UNSUPPORTED, "sample mode has no device generator (host tables only)");
    if (radial && mode != HR_SAMPLE_SOBOL) FAIL(c, HR_ERR_INVALID, "radial is defined for Sobol only");
    if (count == 0 || !out) FAIL(c, HR_ERR_INVALID, "bad count / output");
    float2 *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)count * sizeof(float2)));
    launchQmc(c->stream, mode, seqIndex, count, d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)count * sizeof(float2), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    HIP_TRY(c, e);
    if (radial) radialOnHost(reinterpret_cast<float2 *>(out), count);
    return HR_OK;
}

int hr_sequences_generate(hr_ctx *c, int32_t sampleMode, int32_t bokeh, int32_t len)
{
    ENTER(c);
    if (sampleMode != HR_SAMPLE_SOBOL && sampleMode != HR_SAMPLE_HALTON && sampleMode != HR_SAMPLE_HAMMERSLEY)
        FAIL(c, HR_ERR_UNSUPPORTED, "sample mode has no device generator: upload host tables with hr_sequences_set");
    if (bokeh != HR_BOKEH_CIRCULAR) FAIL(c, HR_ERR_UNSUPPORTED, "polygonal bokeh tables are host-generated: use hr_sequences_set");
    if (len <= 0) FAIL(c, HR_ERR_INVALID, "bad sequence length");
    const int nSeq = HR_NUM_RANDOM_SEQUENCES;
    int rc = setTable(c, &c->dSeq, nullptr, (size_t)nSeq * len);
    if (rc) return rc;
    rc = setTable(c, &c->dAperture, nullptr, (size_t)nSeq * len);
    if (rc) return rc;
    for (int s = 0; s < nSeq; ++s) { // PassGenerator.cpp:614-662
        launchQmc(c->stream, sampleMode, (uint32_t)s, (uint32_t)len, c->dSeq + (size_t)s * len);
        launchQmc(c->stream, HR_SAMPLE_SOBOL, (uint32_t)s, (uint32_t)len, c->dAperture + (size_t)s * len);
    }
    HIP_TRY(c, hipGetLastError());
    { // the aperture tables: Sobol points from the device, disk mapping on the host (see radialOnHost)
        std::vector<float2> ap((size_t)nSeq * len);
        HIP_TRY(c, hipMemcpyAsync(ap.data(), c->dAperture, ap.size() * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        radialOnHost(ap.data(), ap.size());
        HIP_TRY(c, hipMemcpyAsync(c->dAperture, ap.data(), ap.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->nSeq = nSeq, c->seqLen = len;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_seq_offsets_generate(hr_ctx *c)
{
    ENTER(c);
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    const size_t n = (size_t)c->W * c->H;
    int rc = setTable(c, &c->dSeqOffsets, nullptr, n);
    if (rc) return rc;
    launchQmc(c->stream, HR_SAMPLE_SOBOL, 0, (uint32_t)n, c->dSeqOffsets); // PassGenerator.cpp:150-159
    HIP_TRY(c, hipGetLastError());
    c->nSeqOffsets = (int)n;
    c->sceneDirty = true;
    return HR_OK;
}

int hr_multiscatter_lut_generate(hr_ctx *c, float *out, hr_tex_id *outTex)
{
    ENTER(c);
    float2 *seq = nullptr;
    float *lut = nullptr;
    HIP_TRY(c, hipMalloc(&seq, 4096 * sizeof(float2)));
    HIP_TRY(c, hipMalloc(&lut, 128 * 128 * sizeof(float)));
    launchQmc(c->stream, HR_SAMPLE_SOBOL, 0, 4096, seq); // MultiScatterUtil.cpp:102-104
    launchMultiscatterLUT(c->stream, seq, lut);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && out) e = hipMemcpy(out, lut, 128 * 128 * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(seq);
    if (e != hipSuccess) {
        hipFree(lut);
        HIP_TRY(c, e);
    }
    if (outTex) { // loadMultiscatterTexture: LINEAR + CLAMP_TO_EDGE sampler (TextureLoader.cpp:36-41)
        Texture t;
        t.dpx = lut;
        t.desc = TexDesc{lut, 128, 128, 1, HR_WRAP_CLAMP_TO_EDGE, HR_WRAP_CLAMP_TO_EDGE, HR_FILTER_LINEAR, HR_TEX_F32, 0, nullptr, 0.0f, 0};
        t.alive = true;
        c->textures.push_back(t);
        c->sceneDirty = true;
        *outTex = (hr_tex_id)c->textures.size() - 1;
    } else {
        hipFree(lut);
    }
    return HR_OK;
}

// ------------------------------------------------------------------------------------------ pass
static int uploadScene(hr_ctx *c)
{
    if (!c->sceneDirty) return HR_OK;
    QUIESCE(c);
    if (c->dMaterialsCap < c->materials.size() || !c->dMaterials) {
        hipFree(c->dMaterials);
        c->dMaterialsCap = c->materials.size() + 16;
        HIP_TRY(c, hipMalloc(&c->dMaterials, c->dMaterialsCap * sizeof(hr_material)));
    }
    if (!c->materials.empty())
        HIP_TRY(c, hipMemcpy(c->dMaterials, c->materials.data(), c->materials.size() * sizeof(hr_material), hipMemcpyHostToDevice));
    if (c->dTexturesCap < c->textures.size() || !c->dTextures) {
        hipFree(c->dTextures);
        c->dTexturesCap = c->textures.size() + 16;
        HIP_TRY(c, hipMalloc(&c->dTextures, c->dTexturesCap * sizeof(TexDesc)));
    }
    std::vector<TexDesc> td(c->textures.size());
    for (size_t i = 0; i < td.size(); ++i) {
        td[i] = c->textures[i].desc;
        if (!c->textures[i].alive) td[i].px = nullptr;
    }
    if (!td.empty()) {
        HIP_TRY(c, hipMemcpy(c->dTextures, td.data(), td.size() * sizeof(TexDesc), hipMemcpyHostToDevice));
        launchTexLodScale(c->stream, c->dTextures, (int)td.size()); // TexDesc::lodScale, in the device's (= the oracle's) arithmetic
    }
    SceneDev &s = c->hScene;
    s.materials = c->dMaterials, s.nMaterials = (int)c->materials.size();
    s.textures = c->dTextures, s.nTextures = (int)c->textures.size();
    s.lights = c->lights;
    s.seq = c->dSeq, s.aperture = c->dAperture, s.seqOffsets = c->dSeqOffsets;
    s.nSeq = c->nSeq, s.seqLen = c->seqLen, s.nSeqOffsets = c->nSeqOffsets;
    s.envRowCdf = c->dEnvRowCdf, s.envColCdf = c->dEnvColCdf, s.envProb = c->dEnvProb;
    s.envRowGuide = c->dEnvRowGuide, s.envColGuide = c->dEnvColGuide;
    s.envW = c->envW, s.envH = c->envH, s.envMeanLum = c->envMeanLum;
    s.blockNx = c->blockNx, s.blockNy = c->blockNy;
    std::memcpy(s.blockCoords, c->blockCoords, sizeof(s.blockCoords));
    s.texDensity = c->texDensityStale ? nullptr : c->dTexDensity;
    HIP_TRY(c, hipMemcpy(c->dScene, &s, sizeof(SceneDev), hipMemcpyHostToDevice));
    // rays can outlive maxRayDepth only by passing through single-sided / alpha-masked surfaces
    c->hasPassthrough = false, c->hasGlass = false;
    for (const hr_material &m : c->materials) {
        if (m.type == HR_MAT_PBR && (!(m.flags & HR_MF_DOUBLE_SIDED) || (m.flags & HR_MF_ALPHA_MASK))) c->hasPassthrough = true;
        if (m.type == HR_MAT_GLASS) c->hasGlass = true;
    }
    c->sceneDirty = false;
    return HR_OK;
}

// HR_ESTIMATOR_ENV_MIS: (re)build the importance table of the current environment map on the device
static int ensureEnvTable(hr_ctx *c)
{
    const int id = c->lights.env_texture;
    const bool have = c->lights.env_enabled && id >= 0 && id < (int)c->textures.size() && c->textures[id].alive;
    if (!have) {
        if (c->envW != 0) {
            QUIESCE(c);
            c->envW = c->envH = 0, c->envTex = -2;
            c->sceneDirty = true;
        }
        return HR_OK;
    }
    const TexDesc &t = c->textures[id].desc;
    if (c->envTex == id && c->envW == t.w && c->envH == t.h) return HR_OK;
    if (t.w > 65535 || t.h > 65535) FAIL(c, HR_ERR_UNSUPPORTED, "environment map too large for the importance table (65535 texels per side)");
    QUIESCE(c);
    hipFree(c->dEnvRowCdf), hipFree(c->dEnvColCdf), hipFree(c->dEnvProb), hipFree(c->dEnvRowGuide), hipFree(c->dEnvColGuide);
    c->dEnvRowCdf = c->dEnvColCdf = c->dEnvProb = nullptr, c->envW = c->envH = 0, c->envTex = -2;
    c->dEnvRowGuide = c->dEnvColGuide = nullptr;
    const size_t n = (size_t)t.w * t.h;
    float *lum = nullptr, *dil = nullptr;
    uint32_t *wq = nullptr, *maxBits = nullptr;
    unsigned long long *rowSum = nullptr;
    hipError_t e = hipMalloc(&c->dEnvRowCdf, sizeof(float) * ((size_t)t.h + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvColCdf, sizeof(float) * (size_t)t.h * ((size_t)t.w + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvProb, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&c->dEnvRowGuide, sizeof(uint16_t) * (kEnvRowGuide + 1));
    if (e == hipSuccess) e = hipMalloc(&c->dEnvColGuide, sizeof(uint16_t) * (size_t)t.h * (kEnvColGuide + 1));
    if (e == hipSuccess) e = hipMalloc(&lum, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&dil, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&wq, sizeof(uint32_t) * n);
    if (e == hipSuccess) e = hipMalloc(&rowSum, sizeof(unsigned long long) * ((size_t)t.h + 1));
    if (e == hipSuccess) e = hipMalloc(&maxBits, 16);
    if (e == hipSuccess) {
        float *dMean = reinterpret_cast<float *>(maxBits) + 1;
        launchEnvTable(c->stream, t, lum, dil, wq, rowSum, rowSum + t.h, maxBits, c->dEnvRowCdf, c->dEnvColCdf, c->dEnvProb, dMean);
        launchEnvGuides(c->stream, c->dEnvRowCdf, c->dEnvColCdf, t.w, t.h, c->dEnvRowGuide, c->dEnvColGuide);
        e = hipMemcpyAsync(&c->envMeanLum, dMean, sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    hipFree(lum), hipFree(dil), hipFree(wq), hipFree(rowSum), hipFree(maxBits);
    HIP_TRY(c, e);
    c->envW = t.w, c->envH = t.h, c->envTex = id;
    c->sceneDirty = true;
    return HR_OK;
}

// HR_TEXTURE_LOD_CONE: build what the mode needs and is missing — the mip chains of the textures and the per-triangle level offset
static int ensureTextureLod(hr_ctx *c)
{
    bool quiesced = false;
    for (Texture &t : c->textures) {
        if (!t.alive || t.desc.nLevels != 0) continue;
        if (!quiesced) {
            QUIESCE(c);
            quiesced = true;
        }
        TexDesc &d = t.desc;
        int levels = 1;
        size_t elems = 0;
        for (int w = d.w, h = d.h; (w > 1 || h > 1) && d.filter != HR_FILTER_NEAREST; ++levels) {
            w = w / 2 < 1 ? 1 : w / 2, h = h / 2 < 1 ? 1 : h / 2;
            elems += (size_t)w * h * d.c;
        }
        if (levels > 1) {
            HIP_TRY(c, hipMalloc(&t.dmips, elems * sizeof(float)));
            launchMipChain(c->stream, d, levels, t.dmips);
        }
        d.nLevels = levels, d.mips = t.dmips;
        c->sceneDirty = true;
    }
    if (c->texDensityStale && c->tree.tris) {
        if (!quiesced) {
            QUIESCE(c);
            quiesced = true;
        }
        const size_t nTris = c->treeTris;
        if (c->texDensityCap < nTris || !c->dTexDensity) {
            hipFree(c->dTexDensity);
            c->dTexDensity = nullptr, c->texDensityCap = 0;
            HIP_TRY(c, hipMalloc(&c->dTexDensity, sizeof(float) * (nTris ? nTris : 1)));
            c->texDensityCap = nTris;
        }
        const uint32_t slots = c->tree.triSlots ? c->tree.triSlots : (uint32_t)nTris;
        launchTexDensity(c->stream, c->tree.tris, slots, c->attrs, c->dTexDensity);
        c->texDensityStale = false;
        c->sceneDirty = true;
    }
    if (quiesced) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HR_OK;
}

int hr_clear(hr_ctx *c)
{
    ENTER(c);
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    int rc = drainPipeline(c);
    if (rc) return rc;
    if (c->hOverflow && c->hOverflow[0]) { // a dropped-rays report: the frame starts afresh and so does the report, once nothing that could repeat it is running
        QUIESCE(c);
        c->hOverflow[0] = c->hOverflow[1] = c->hOverflow[2] = c->hOverflow[3] = 0u;
    }
    HIP_TRY(c, hipMemsetAsync(c->fb(), 0, (size_t)c->W * c->H * 4 * sizeof(float), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dStats, 0, sizeof(Stats) * kStatSlots, c->stream));
    c->resolvedAtClear = c->nextResolveOrder;
    c->snapshotEpoch++;
    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "hr_clear %p: ray-memory growths so far %llu, waits %llu (%.2f ms)\n", (void *)c, c->dbgGrowths, c->dbgWaits, (double)c->dbgWaitNs * 1e-6);
    c->drainTimes();
    for (int k = 0; k < HR_KERNEL_COUNT; ++k) c->kernelMs[k] = 0.0f, c->kernelLaunches[k] = 0;
    return HR_OK;
}

static int allocSlot(hr_ctx *c, hr_ctx::PassSlot &ps)
{
    const size_t fbBytes = (size_t)c->W * c->H * 4 * sizeof(float);
    // (with HR_ESTIMATOR_ALL_LIGHTS the sample's further partial sums lie right behind the first: k_trace indexes one buffer)
    const hipError_t e = hipMalloc(&ps.passbuf, fbBytes * (c->allLightsUsed ? 4 : 1));
    if (e != hipSuccess) { // say what ran out: a pass slot is the unit the pipeline's memory grows in
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        c->err = "pass slot " + std::to_string(c->nSlotsAllocated + 1) + " (" + std::to_string(fbBytes >> 20) + " MiB pass buffer at " + std::to_string(c->W) + "x" +
                 std::to_string(c->H) + "): " + hipGetErrorString(e) + "; " + std::to_string(freeB >> 20) + " MiB of device memory free";
        return HR_ERR_DEVICE;
    }
    if (c->allLightsUsed) ps.passbufB = ps.passbuf + (size_t)c->W * c->H * 4;
    ps.ctr = c->dCounters + (&ps - c->slots);
    HIP_TRY(c, hipEventCreateWithFlags(&ps.evFinal, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&ps.evResolved, hipEventDisableTiming));
    ps.allocated = true;
    c->nSlotsAllocated++;
    return HR_OK;
}

// ---- the groups' ray memory (hr_ctx::Group::arena / scratch)
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
static size_t rayQueueBytes(size_t cap) { return 4 * align256(cap * 16); }
static size_t shadowQueueBytes(size_t cap) { return 3 * align256(cap * 16); }
static RayQueue carveRayQueue(char *&p, size_t cap)
{
    RayQueue q;
    const size_t n = align256(cap * 16);
    q.A = (float4 *)p, q.B = (float4 *)(p + n), q.C = (float4 *)(p + 2 * n), q.D = (int4 *)(p + 3 * n);
    p += 4 * n;
    return q;
}
static ShadowQueue carveShadowQueue(char *&p, size_t cap)
{
    ShadowQueue q;
    const size_t n = align256(cap * 16);
    q.A = (float4 *)p, q.B = (float4 *)(p + n), q.C = (float4 *)(p + 2 * n);
    p += 3 * n;
    return q;
}
// a region that is too small is replaced once everything the group has enqueued is done (what it held is dead by then: a step's
// scratch dies with the step, and arena[t & 1] holds the rays step t - 2 emitted, which step t - 1 consumed)
static int ensureRegion(hr_ctx *c, hr_ctx::Group &G, hr_ctx::Group::Region &r, size_t need, const char *what)
{
    if (need <= r.cap) return HR_OK;
    c->dbgGrowths++, c->dbgGrowBytes += need;
    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "  grow %s: need %.1f MiB, had %.1f MiB (step %llu)\n", what, (double)need / 1048576.0, (double)r.cap / 1048576.0, G.stepCounter);
    HIP_TRY(c, hipStreamSynchronize(G.stream));
    const size_t hadCap = r.cap;
    hipFree(r.base);
    r.base = nullptr, r.cap = 0;
    // a third of headroom: counts vary from pass to pass, and while the pipeline fills (the first depth + 2 steps of a render) every
    // step carries one more generation of passes — for the benchmark soup the steady state needs 27 % more than the step that
    // triggered the last growth (profiles/r4m_mem.txt); a step that needs more regrows once more
    size_t want = need + need / 3;
    if (hadCap && !c->memBudget && want < hadCap + hadCap / 2) want = hadCap + hadCap / 2; // (a region that has to grow again grows by half at least: few events; under a memory budget only by what is needed)
    want = (want + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
    hipError_t e = hipMalloc((void **)&r.base, want);
    size_t got = want;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        got = align256(need);
        e = hipMalloc((void **)&r.base, got);
    }
    if (e != hipSuccess) {
        size_t freeB = 0, totalB = 0;
        hipMemGetInfo(&freeB, &totalB);
        c->err = std::string("ray memory (") + what + ", " + std::to_string(need >> 20) + " MiB for one macro step at " + std::to_string(c->W) + "x" + std::to_string(c->H) +
                 "): " + hipGetErrorString(e) + "; " + std::to_string(freeB >> 20) + " MiB of device memory free";
        return HR_ERR_DEVICE;
    }
    r.cap = got;
    return HR_OK;
}

// passes that hold a slot: in flight, or finished and waiting for their turn to resolve
static int occupiedSlots(const hr_ctx *c, int group)
{
    int n = 0;
    for (const hr_ctx::PassSlot &ps : c->slots) n += ((ps.active || ps.finished) && (group < 0 || ps.group == group)) ? 1 : 0;
    return n;
}

static int slotLimit(const hr_ctx *c);
static int activePasses(const hr_ctx *c)
{
    int n = 0;
    for (const hr_ctx::PassSlot &ps : c->slots) n += ps.active ? 1 : 0;
    return n;
}

// Finished passes are added to the frame on the caller's stream, strictly in pass order (float addition order is
// part of the arithmetic contract), whatever order the groups finished them in.
static int resolveReady(hr_ctx *c)
{
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    const LaunchCfg cfg = c->cfg(c->stream);
    for (;;) {
        // collect the passes whose turn it is (up to kMaxBatch) and add them with one launch
        PassBufList bufs{};
        hr_ctx::PassSlot *ready[kMaxBatch];
        while (bufs.n < kMaxBatch) {
            hr_ctx::PassSlot *next = nullptr;
            const unsigned long long want = c->nextResolveOrder + (unsigned long long)bufs.n;
            for (hr_ctx::PassSlot &ps : c->slots)
                if ((ps.active || ps.finished) && ps.order == want) next = &ps;
            if (!next || !next->finished) break;
            ready[bufs.n] = next;
            bufs.bufB[bufs.n] = next->pp.estimator == HR_ESTIMATOR_ALL_LIGHTS ? next->passbufB : nullptr;
            bufs.buf[bufs.n++] = next->passbuf;
        }
        if (bufs.n == 0) {
            // nothing requested is unfinished any more: the age of "the oldest waiting request" starts afresh with the next request
            // (stamped only in hr_render_pass, it used to survive every pass that completed the normal way, so that 4 ms after the
            // first request EVERY progressive read-back that found the streams idle drained a partly filled batch)
            if (c->pendingInject.empty() && occupiedSlots(c) == 0) c->oldestWaitingNs = 0;
            return HR_OK;
        }
        for (int k = 0; k < bufs.n; ++k) {
            bool seen = false;
            for (int j = 0; j < k; ++j) seen = seen || ready[j]->finalEv == ready[k]->finalEv;
            if (!seen) HIP_TRY(c, hipStreamWaitEvent(c->stream, ready[k]->finalEv, 0));
        }
        c->timeBegin(HR_KERNEL_RESOLVE, c->stream);
        launchResolve(cfg, fr, bufs);
        c->timeEnd(c->stream);
        HIP_TRY(c, hipEventRecord(ready[bufs.n - 1]->evResolved, c->stream));
        for (int k = 0; k < bufs.n; ++k) {
            ready[k]->resolvedEv = ready[bufs.n - 1]->evResolved;
            ready[k]->finished = false, ready[k]->everResolved = true;
            ready[k]->resolvedAt = c->nextResolveOrder;
            c->nextResolveOrder++;
        }
    }
}

// wait for the queue lengths step (want - 1) of this group reports when its k_trace starts (Group::hCounts)
static int waitCounts(hr_ctx *c, hr_ctx::Group &G, int ring, unsigned long long want)
{
    const auto t0 = std::chrono::steady_clock::now();
    c->dbgWaits++;
    for (unsigned spins = 0;; ++spins) {
        if (G.hSeq[ring] == want) {
            if (spins) c->dbgWaitSpun++, c->dbgWaitNs += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            break;
        }
        if ((spins & 255u) == 255u) {
            const hipError_t q = hipStreamQuery(G.stream);
            if (q == hipSuccess) { // everything enqueued has run: the report must have arrived
                if (G.hSeq[ring] == want) break;
                FAIL(c, HR_ERR_DEVICE, "internal: a step's queue lengths never arrived");
            }
            if (q != hipErrorNotReady) HIP_TRY(c, q);
            std::this_thread::yield();
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return HR_OK;
}

// One macro step of pipeline group g: (raygen of the injected passes) -> trace of every in-flight pass of the group ->
// shade; passes whose last stage this was become `finished`.
static const int kProbeEvery = 64; // injecting steps between two probes of the packet selector
static int stagesOf(const hr_ctx *c, const hr_pass_params &pp);
static int packetLog2(const hr_ctx *c);
static bool packetsInUse(const hr_ctx *c);
static int macroStep(hr_ctx *c, int g, int nInject)
{
    static const bool dbgT = getenv("HR_DEBUG_STEPTIMES") != nullptr;
    auto nowUs = [] { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3; };
    const double tA = dbgT ? nowUs() : 0.0;
    double tB = 0, tC = 0, tD = 0, tE = 0;
    hr_ctx::Group &G = c->groups[g];
    const LaunchCfg cfg = c->cfg(G.stream);
    FrameDev fr = c->frame;
    fr.fb = c->fb();
    if (G.needUserSync) { // state set up on the caller's stream (scene, tables, cleared buffers) must be visible
        HIP_TRY(c, hipEventRecord(G.evUser, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(G.stream, G.evUser, 0));
        G.needUserSync = false;
    }
    int injectedSlots[kMaxSegs];
    hipEvent_t waited[kMaxSegs];
    int nInjected = 0;
    for (int k = 0; k < nInject; ++k) {
        const hr_pass_params pp = c->pendingInject.front();
        c->pendingInject.pop_front();
        // Reuse the free slot whose pass was resolved longest ago: the injection waits for that resolve, and a slot freed
        // by the step just enqueued would chain this group's step behind the other group's (no overlap).  A slot resolved
        // only recently is passed over for fresh memory while the slot budget allows.
        int slot = -1, fresh = -1;
        for (int i = 0; i < kMaxSlots; ++i) {
            const hr_ctx::PassSlot &cand = c->slots[i];
            if (cand.active || cand.finished) continue;
            if (!cand.allocated) {
                if (fresh < 0) fresh = i;
            } else if (slot < 0 || cand.resolvedAt < c->slots[slot].resolvedAt) {
                slot = i;
            }
        }
        const unsigned long long recent = 2ull * (unsigned long long)c->nGroups * (unsigned long long)(nInject > 0 ? nInject : 1);
        if (fresh >= 0 && c->nSlotsAllocated < slotLimit(c) && (slot < 0 || (c->nGroups > 1 && c->nextResolveOrder - c->slots[slot].resolvedAt < recent)))
            slot = fresh;
        if (slot < 0) slot = fresh;
        if (slot < 0) FAIL(c, HR_ERR_INVALID, "internal: no free pass slot");
        hr_ctx::PassSlot &ps = c->slots[slot];
        if (!ps.allocated) {
            int rc = allocSlot(c, ps);
            if (rc) return rc;
        }
        if (ps.everResolved) { // the pass buffer is free again once the launch that resolved it has run
            bool seen = false;
            for (int j = 0; j < nInjected; ++j) seen = seen || waited[j] == ps.resolvedEv;
            if (!seen) HIP_TRY(c, hipStreamWaitEvent(G.stream, ps.resolvedEv, 0));
        }
        waited[nInjected] = ps.everResolved ? ps.resolvedEv : nullptr;
        ps.active = true, ps.finished = false, ps.group = g, ps.step = 0, ps.nIter = pp.max_ray_depth + 1, ps.pp = pp;
        ps.qcur = RayQueue{}, ps.scur = ShadowQueue{}, ps.capCur = 0, ps.sCapCur = 0;
        ps.order = c->injected++;
        injectedSlots[nInjected++] = slot;
    }
    for (int j0 = 0; j0 < nInjected; j0 += kMaxBatch) { // the injected passes' counters back to zero, one launch
        CounterList cl{};
        for (int j = j0; j < nInjected && cl.n < kMaxBatch; ++j) cl.ctr[cl.n++] = c->slots[injectedSlots[j]].ctr;
        launchZeroCounters(cfg, cl);
    }
    // Pass-through rays (back faces of single-sided materials, alpha masks: physicallyBased.rlsl:70-108) are not bounded by
    // maxRayDepth, so in such scenes a pass runs until its closest-hit queue is empty.  The queue lengths come from the snapshot
    // taken two macro steps ago (see Group::hQCount): a slot about to run stage `st` then knows the lengths of stages <= st - 1;
    // if stage st - 1 had no rays, it emitted nothing and the pass was complete with the steps already enqueued.
    if (c->hasPassthrough) {
        const unsigned long long N = G.stepCounter;
        if (N >= 2 && G.statusUsed[(N - 2) % kTableRing]) {
            const int ring = (int)((N - 2) % kTableRing);
            HIP_TRY(c, hipEventSynchronize(G.statusEv[ring]));
            hr_ctx::PassSlot *endedEarly[kMaxSlots];
            int nEndedEarly = 0;
            const uint32_t *snap = G.hQCount + (size_t)ring * kMaxSlots * kMaxBounceSlots;
            for (int i = 0; i < kMaxSlots; ++i) {
                hr_ctx::PassSlot &ps = c->slots[i];
                if (!ps.active || ps.group != g || G.statusOrder[ring][i] != ps.order + 1ull) continue;
                const int st = ps.step;
                const bool empty = st >= 2 && snap[(size_t)i * kMaxBounceSlots + ((st - 1) % kMaxBounceSlots)] == 0u;
                if (empty) {
                    ps.active = false, ps.finished = true;
                    endedEarly[nEndedEarly++] = &ps;
                }
            }
            hr_ctx::PassSlot *owner = nullptr; // the newest of them: it is resolved last, so its event outlives the others' waits
            for (int k = 0; k < nEndedEarly; ++k)
                if (!owner || endedEarly[k]->order > owner->order) owner = endedEarly[k];
            if (owner) HIP_TRY(c, hipEventRecord(owner->evFinal, G.stream));
            for (int k = 0; k < nEndedEarly; ++k) endedEarly[k]->finalEv = owner->evFinal;
        }
    }
    if (dbgT) tB = nowUs();
    // table of the group's in-flight passes, oldest first
    int order[kMaxSlots], n = 0;
    for (int i = 0; i < kMaxSlots; ++i)
        if (c->slots[i].active && c->slots[i].group == g) order[n++] = i;
    for (int a = 1; a < n; ++a)
        for (int b = a; b > 0 && c->slots[order[b]].order < c->slots[order[b - 1]].order; --b) std::swap(order[b], order[b - 1]);
    if (n == 0) return resolveReady(c);
    if (n > kMaxSegs) FAIL(c, HR_ERR_INVALID, "internal: too many passes in one group");
    const unsigned long long stepIdx = G.stepCounter++;
    const int ring = (int)(stepIdx % kTableRing);
    if (G.tableUsed[ring]) HIP_TRY(c, hipEventSynchronize(G.tableCopied[ring])); // staging entry free again (4 steps old)
    // ---- ray memory of this step (Group::arena): every queue sized by an upper bound of what can arrive in it
    const uint32_t P = c->tuneOverflowTest == 1 ? c->queueCapacity / 8u + 1u : (c->queueCapacity ? c->queueCapacity : 1u); // (ovf=1, TEST ONLY: camera rays do not fit)
    const size_t kS = c->allLightsUsed ? 4 : 1;
    uint32_t boundIn[kMaxSegs];
    {
        bool wanted = false;
        for (int k = 0; k < n; ++k) wanted = wanted || c->slots[order[k]].step > 0;
        int idxOfSlot[kMaxSlots];
        const int prev = (int)((stepIdx + kTableRing - 1) % kTableRing);
        if (wanted && stepIdx > 0) {
            int rc = waitCounts(c, G, prev, stepIdx); // (step stepIdx - 1 wrote stepIdx: its number + 1)
            if (rc) return rc;
            if (g == 0 && c->probePending && stepIdx > c->probeStep) { // (steps from the probe's own on report the totals) complete once every wave of the probe has counted itself
                const unsigned long long pk = G.hProbe[4 * prev], ry = G.hProbe[4 * prev + 1], done = G.hProbe[4 * prev + 2], nr = G.hProbe[4 * prev + 3];
                if (done - c->probeSeen[2] >= c->probeWaves) {
                    const unsigned long long dPk = pk - c->probeSeen[0], dRy = ry - c->probeSeen[1], dNr = nr - c->probeSeen[3];
                    c->probePending = false, c->probeSeen[0] = pk, c->probeSeen[1] = ry, c->probeSeen[2] = done, c->probeSeen[3] = nr;
                    if (dRy > 0) {
                        c->lastUnion = (double)dPk / (double)dRy;
                        c->packetsOn = c->lastUnion * 100.0 < (double)c->tunePacketUnion;
                        c->lastOwnPerRay = dNr ? (double)dRy / (double)dNr : 0.0;
                    }
                    if (getenv("HR_DEBUG_PIPE")) fprintf(stderr, "packet probe of step %llu (seen at step %llu): union %.3f, %.1f child boxes entered per ray -> packets %s\n", c->probeStep, stepIdx, c->lastUnion, c->lastOwnPerRay, c->packetsOn ? "on" : "off");
                }
            }
            for (int i = 0; i < kMaxSlots; ++i) idxOfSlot[i] = -1;
            for (int j = 0; j < G.countN[prev]; ++j) idxOfSlot[G.countSlot[prev][j]] = j;
        }
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            uint32_t b = P;
            if (ps.step > 0) {
                b = ps.capCur; // (what its queue can hold is a bound too: used when the pass was not in the previous step's table)
                const int j = (wanted && stepIdx > 0) ? idxOfSlot[order[k]] : -1;
                if (j >= 0 && G.countOrder[prev][j] == ps.order + 1ull) {
                    const uint32_t seen = G.hCounts[(size_t)prev * kMaxSegs + j]; // length of its closest-hit queue one stage ago
                    b = seen < b ? seen : b;
                    if (ps.step == 1 && seen < P) c->lastCameraCount = seen; // (camera rays that passed the root cull: what a packet kernel traces per pass)
                }
            }
            // TEST ONLY (HR_TUNE="ovf=": tests/test_gpu_parity.py forces every kind of overflow once): half of what the bound should be
            if (c->tuneOverflowTest == 2 && ps.step == 1) b = b / 2u + 1u;
            boundIn[k] = b;
        }
    }
    if (dbgT) tC = nowUs();
    size_t needArena = 0, needScratch = 0;
    for (int k = 0; k < n; ++k) {
        const hr_ctx::PassSlot &ps = c->slots[order[k]];
        const bool closest = c->hasPassthrough || ps.step < ps.nIter;
        if (ps.step == 0) needScratch += rayQueueBytes(P);
        if (closest) {
            needScratch += align256((size_t)boundIn[k] * hitRecordSize()) + align256((size_t)boundIn[k] * 4);
            needArena += rayQueueBytes(boundIn[k]) + shadowQueueBytes((size_t)boundIn[k] * kS);
        }
    }
    if (c->memBudget) { // what this step carves per pass and stage (budgetBytesPerPass)
        double sumA[kMaxBounceSlots] = {0}, sumS[kMaxBounceSlots] = {0};
        int cnt[kMaxBounceSlots] = {0};
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            const int st = ps.step < kMaxBounceSlots ? ps.step : kMaxBounceSlots - 1;
            const bool closest = c->hasPassthrough || ps.step < ps.nIter;
            cnt[st]++;
            if (ps.step == 0) sumS[st] += (double)rayQueueBytes(P);
            if (closest) {
                sumS[st] += (double)(align256((size_t)boundIn[k] * hitRecordSize()) + align256((size_t)boundIn[k] * 4));
                sumA[st] += (double)(rayQueueBytes(boundIn[k]) + shadowQueueBytes((size_t)boundIn[k] * kS));
            }
        }
        for (int st = 0; st < kMaxBounceSlots; ++st)
            if (cnt[st]) {
                const double a = sumA[st] / cnt[st], sc = sumS[st] / cnt[st];
                c->stageArenaSeen[st] = (c->stageSeen[st] && c->stageArenaSeen[st] > a) ? c->stageArenaSeen[st] : a;
                c->stageScratchSeen[st] = (c->stageSeen[st] && c->stageScratchSeen[st] > sc) ? c->stageScratchSeen[st] : sc;
                c->stageSeen[st] = true;
            }
    }
    hr_ctx::Group::Region &arena = G.arena[stepIdx & 1ull];
    {
        G.arenaHighWater = needArena > G.arenaHighWater ? needArena : G.arenaHighWater;
        int rc = ensureRegion(c, G, arena, G.arenaHighWater, "rays emitted by a step");
        if (rc == HR_OK) rc = ensureRegion(c, G, G.scratch, needScratch, "camera rays and hit records of a step");
        if (rc) {
            // out of device memory: nothing of this step has been enqueued except the counters' reset.  The passes it was to inject go
            // back to the head of the request queue (their slots are free again), so that a later call — after the caller has released
            // memory — injects them properly instead of tracing queues no k_raygen ever filled.
            for (int j = nInjected - 1; j >= 0; --j) {
                hr_ctx::PassSlot &ps = c->slots[injectedSlots[j]];
                c->pendingInject.push_front(ps.pp);
                ps.active = false;
                c->injected--;
            }
            G.stepCounter--;
            return rc;
        }
    }
    char *pArena = arena.base, *pScratch = G.scratch.base;
    StepTable &tbl = G.hTables[ring];
    std::memset(tbl.heads, 0, sizeof(tbl.heads));
    std::memset(tbl.clkStart, 0xFF, sizeof(tbl.clkStart)), std::memset(tbl.clkEnd, 0, sizeof(tbl.clkEnd));
    tbl.headsLog2 = (uint32_t)(c->tuneHeads < 0 ? 0 : (c->tuneHeads > 6 ? 6 : c->tuneHeads));
    tbl.nSeg = n;
    tbl.refillLanes = c->tuneRefill, tbl.triPhaseLanes = c->tuneTri;
    tbl.fetchMax = c->tuneFetchMax > 0 ? c->tuneFetchMax : 1, tbl.fetchMin = c->tuneFetchMin > 0 ? c->tuneFetchMin : 1;
    tbl.staticPerWave = c->tuneStaticDeal, tbl.hasGlass = c->hasGlass ? 1 : 0;
    tbl.primaryFromSeg = n, tbl.fetchMaxPrimary = ((c->tuneFetchPrimary > 0 ? c->tuneFetchPrimary : 1) & 0xFFFF) | ((c->tuneFetchGate & 0xFFFF) << 16); // (primaryFromSeg is set below, once the injected passes' places in the table are known)
    int injectedSegs[kMaxSegs];
    int nInjectedSegs = 0;
    for (int k = 0; k < n; ++k) {
        hr_ctx::PassSlot &ps = c->slots[order[k]];
        SegDev &sg = tbl.seg[k];
        const int st = ps.step;
        const bool closest = c->hasPassthrough || st < ps.nIter;
        sg.qin = st == 0 ? carveRayQueue(pScratch, P) : ps.qcur; // (a new pass's camera rays live for this step only)
        sg.sqIn = ps.scur;                                        // (nothing to trace there in a pass's first step: sCountIn is the zero word)
        sg.qinCap = st == 0 ? P : ps.capCur, sg.sInCap = st == 0 ? 0u : ps.sCapCur, sg.sOutCap = 0u;
        sg.qout = RayQueue{}, sg.sqOut = ShadowQueue{}, sg.hits = nullptr, sg.hitIdx = nullptr;
        if (closest) {
            sg.hits = (HitRec *)pScratch, pScratch += align256((size_t)boundIn[k] * hitRecordSize());
            sg.hitIdx = (uint32_t *)pScratch, pScratch += align256((size_t)boundIn[k] * 4);
            sg.qout = carveRayQueue(pArena, boundIn[k]);
            sg.sqOut = carveShadowQueue(pArena, (size_t)boundIn[k] * kS);
            ps.qcur = sg.qout, ps.scur = sg.sqOut, ps.capCur = boundIn[k];
            sg.sOutCap = (uint32_t)((size_t)boundIn[k] * kS);
            if (c->tuneOverflowTest == 3 && st == 0) sg.sOutCap = sg.sOutCap / 8u + 1u; // TEST ONLY: the first hits' occlusion rays do not fit
            ps.sCapCur = sg.sOutCap;
        }
        sg.passbuf = ps.passbuf;
        sg.passbufB = ps.pp.estimator == HR_ESTIMATOR_ALL_LIGHTS ? ps.passbufB : nullptr;
        // The per-stage counters are a ring: a chain of pass-through rays (stacked single-sided sheets seen from behind, alpha holes:
        // physicallyBased.rlsl:70-108 re-emits without a depth bound) can outlive any fixed number of stages, so from stage
        // kMaxBounceSlots - 1 on the entries this step appends to are cleared first (their previous use lies a whole ring back).
        const int R = kMaxBounceSlots;
        if (st + 1 >= R) {
            HIP_TRY(c, hipMemsetAsync(&ps.ctr->qCount[(st + 1) % R], 0, sizeof(uint32_t), G.stream));
            if (st >= R) {
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->sCount[st % R], 0, sizeof(uint32_t), G.stream));
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->pCount[st % R], 0, sizeof(uint32_t), G.stream));
                HIP_TRY(c, hipMemsetAsync(&ps.ctr->gCount[st % R], 0, sizeof(uint32_t), G.stream));
            }
        }
        sg.qCountIn = &ps.ctr->qCount[st % R];
        sg.sCountIn = st > 0 ? &ps.ctr->sCount[(st - 1) % R] : c->dZero;
        sg.qCountOut = &ps.ctr->qCount[(st + 1) % R];
        sg.sCountOut = &ps.ctr->sCount[st % R];
        sg.pCount = &ps.ctr->pCount[st % R], sg.gCount = &ps.ctr->gCount[st % R];
        sg.hitCap = closest ? boundIn[k] : 0u, sg.packets = 0; // (capacity of hits, the hit list and qout: what was carved above)
        sg.pp = ps.pp;
        sg.closestEnabled = closest ? 1 : 0;
        G.countSlot[ring][k] = order[k], G.countOrder[ring][k] = ps.order + 1ull;
        for (int j = 0; j < nInjected; ++j)
            if (order[k] == injectedSlots[j]) injectedSegs[nInjectedSegs++] = k;
    }
    if (nInjectedSegs > 0) tbl.primaryFromSeg = injectedSegs[0]; // (the table is in pass order: the passes injected now are its last entries)
    // packet selector (above): do the injected passes' camera rays travel as packets (k_raygen_packets), beside k_trace or in front of it, and does this step carry a probe?
    int probeSeg = -1;
    bool packetsNow = packetsInUse(c) && nInjectedSegs > 0 && tbl.seg[injectedSegs[0]].pp.interactive_mode == 0; // (interactive sub-passes of one sample share no pixels)
    if (c->tunePackets == 2 && g == 0 && nInjectedSegs > 0 && !c->probePending && tbl.seg[injectedSegs[0]].pp.interactive_mode == 0) {
        const hr_pass_params &pp = tbl.seg[injectedSegs[0]].pp;
        float cam[21] = {pp.fov_tan, pp.aspect_ratio, pp.focus_distance, pp.aperture_radius};
        std::memcpy(cam + 4, pp.view_matrix, sizeof(pp.view_matrix));
        cam[20] = (float)pp.interactive_mode;
        // another camera sees another part of the tree: probe again, but not more often than every eighth injecting step (a camera in motion)
        if (std::memcmp(cam, c->probeCamera, sizeof(cam)) != 0 && c->probeCountdown > 0 && c->probeCountdown <= kProbeEvery - 8) c->probeCountdown = 0;
        if (c->probeCountdown <= 0) {
            probeSeg = injectedSegs[0];
            std::memcpy(c->probeCamera, cam, sizeof(cam));
        } else {
            c->probeCountdown--;
        }
    }
    const bool corunNow = packetsNow && (c->tuneCorun == 2 || (c->tuneCorun == 1 && c->lastOwnPerRay >= (double)c->tuneCorunMin));
    for (int j = 0; j < nInjectedSegs; ++j)
        if (packetsNow) tbl.seg[injectedSegs[j]].packets = corunNow ? 2 : 1;
    tbl.hostCameraCount = corunNow ? G.dCounts + (size_t)kTableRing * kMaxSegs : nullptr;
    tbl.probe = c->dProbe, tbl.hostProbe = (g == 0 && (c->probePending || probeSeg >= 0)) ? G.dProbeHost + 4 * ring : nullptr; // (reported only while a probe is awaited)
    G.countN[ring] = n;
    tbl.hostCounts = G.dCounts + (size_t)ring * kMaxSegs, tbl.hostSeq = G.dSeq + ring, tbl.seqValue = stepIdx + 1ull;
    tbl.stepLog = c->dStepLog, tbl.nInjectedNow = (uint32_t)nInjected, tbl.group = (uint32_t)g;
    tbl.hostOverflow = c->dOverflowHost;
    StepTable *dTbl = G.dTables + ring;
    const size_t tblBytes = offsetof(StepTable, seg) + (size_t)n * sizeof(SegDev);
    if (dbgT) tD = nowUs();
    if (c->tuneTableKernel)
        launchFetchTable(G.stream, G.dTablesHost + ring, dTbl, (tblBytes + 15) & ~(size_t)15);
    else
        HIP_TRY(c, hipMemcpyAsync(dTbl, &tbl, tblBytes, hipMemcpyHostToDevice, G.stream));
    if (dbgT) tE = nowUs();
    HIP_TRY(c, hipEventRecord(G.tableCopied[ring], G.stream));
    G.tableUsed[ring] = true;
    if (c->pending.size() > 8192) c->drainTimes();
    bool timing = false; // (the kernels of a step are enqueued back to back: n + 1 timing events for n kernels)
    bool forked = false;
    for (int j0 = 0; j0 < nInjectedSegs;) { // one launch for the passes injected this step
        // (as packets: ray generation and the camera rays' traversal in one launch per group of 16, 8, 4, 2, 1 passes — the bucket
        // HR_KERNEL_RAYGEN then holds both, HR_KERNEL_TRACE and the step's device clock stay k_trace's own)
        int take = 1;
        if (packetsNow)
            while (2 * take <= nInjectedSegs - j0 && 2 * take <= kMaxBatch) take *= 2;
        else
            take = nInjectedSegs - j0 < kMaxBatch ? nInjectedSegs - j0 : kMaxBatch;
        SegList segs{};
        for (int j = j0; j < j0 + take; ++j) segs.seg[segs.n++] = injectedSegs[j];
        j0 += take;
        bool uniformParams = packetsNow;
        for (int j = 1; j < segs.n && uniformParams; ++j) { // (the usual batch: one camera, one set of options, consecutive sample indices)
            hr_pass_params a = tbl.seg[segs.seg[0]].pp, b = tbl.seg[segs.seg[j]].pp;
            a.sample_index = b.sample_index = 0;
            uniformParams = std::memcmp(&a, &b, sizeof(a)) == 0;
        }
        if (corunNow) { // beside k_trace: fork after the table copy, join before the shading kernels
            LaunchCfg cb = cfg;
            cb.stream = G.streamB;
            if (!forked) {
                HIP_TRY(c, hipEventRecord(G.evFork, G.stream));
                HIP_TRY(c, hipStreamWaitEvent(G.streamB, G.evFork, 0));
                forked = true;
            }
            launchRaygenPackets(cb, c->dScene, c->nodes, c->tris, dTbl, segs, fr, c->dStats, uniformParams);
            continue;
        }
        if (timing)
            c->timeNext(HR_KERNEL_RAYGEN, G.stream);
        else
            c->timeBegin(HR_KERNEL_RAYGEN, G.stream);
        timing = true;
        if (packetsNow)
            launchRaygenPackets(cfg, c->dScene, c->nodes, c->tris, dTbl, segs, fr, c->dStats, uniformParams);
        else
            launchRaygen(cfg, c->dScene, dTbl, segs, fr, c->dStats);
    }
    if (c->tuneShadowProbe) { // measurement only: the coherence of the occlusion rays this step's k_trace is about to trace
        if (!c->dShadowProbe) {
            HIP_TRY(c, hipMalloc(&c->dShadowProbe, 64));
            HIP_TRY(c, hipMemsetAsync(c->dShadowProbe, 0, 64, G.stream));
        }
        SegList sl{};
        uint32_t most = 0;
        for (int k = 0; k < n; ++k) {
            const hr_ctx::PassSlot &ps = c->slots[order[k]];
            if (ps.step >= 1 && (c->tuneShadowProbe == 2 || ps.step == 1) && sl.n < kMaxBatch) sl.seg[sl.n++] = k, most = tbl.seg[k].sInCap > most ? tbl.seg[k].sInCap : most;
        }
        launchShadowProbe(G.stream, c->dScene, c->nodes, c->tris, dTbl, sl, most, c->dShadowProbe);
    }
    if (forked) HIP_TRY(c, hipEventRecord(G.evJoin, G.streamB));
    if (timing)
        c->timeNext(HR_KERNEL_TRACE, G.stream);
    else
        c->timeBegin(HR_KERNEL_TRACE, G.stream);
    if (probeSeg >= 0) {
        // (probeSeen holds the totals of the report the previous decision was taken on: probes never overlap, that probe was complete)
        HIP_TRY(c, hipEventRecord(c->evProbeA, G.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->probeStream, c->evProbeA, 0));
        c->probeWaves = (unsigned long long)launchPacketProbe(c->probeStream, c->dScene, c->nodes, c->tris, tbl.seg[probeSeg].pp, packetLog2(c), fr, c->dProbe);
        HIP_TRY(c, hipEventRecord(c->evProbeB, c->probeStream));
        c->probeGuard = true, c->probePending = true, c->probeStep = stepIdx, c->probeCountdown = kProbeEvery;
    }
    {
        LaunchCfg ct = cfg;
        if (forked) {
            // camera rays of this step (what the passes injected before sent through the root cull, or half the pixels while unknown)
            // against the rays k_trace carries (two per entry of the closest-hit queues' bounds: the ray and its occlusion ray)
            double others = 0.0;
            for (int k = 0; k < n; ++k)
                if (c->slots[order[k]].step > 0) others += 2.0 * (double)((c->slots[order[k]].step == 1 && c->lastCameraCount && boundIn[k] > c->lastCameraCount) ? c->lastCameraCount : boundIn[k]);
            const uint32_t late = ((volatile uint32_t *)G.hCounts)[(size_t)kTableRing * kMaxSegs]; // (k_shade_sort's hint: camera rays per pass behind the root cull)
            if (late) c->lastCameraCount = late;
            const double cam = (double)nInjectedSegs * (double)(c->lastCameraCount ? c->lastCameraCount : P / 2u);
            int blocks = cam > 0.2 * others ? 3 : 4;
            if (c->tuneCorunBlocks > 0) blocks = c->tuneCorunBlocks;
            if (blocks < ct.traceBlocksPerCU) ct.traceBlocksPerCU = blocks;
        }
        launchTrace(ct, c->dScene, c->tree.leafKeys, c->tree.nodes32, c->tris, dTbl, c->dStats);
    }
    if (forked) { // (the bucket HR_KERNEL_TRACE stays k_trace's own launch; what the packet kernel beside it runs longer is booked as ray generation)
        c->timeNext(HR_KERNEL_RAYGEN, G.stream);
        HIP_TRY(c, hipStreamWaitEvent(G.stream, G.evJoin, 0));
    }
    c->timeNext(HR_KERNEL_SHADE, G.stream);
    launchShade(cfg, c->dScene, dTbl, c->dStats);
    c->timeEnd(G.stream);
    hr_ctx::PassSlot *ended[kMaxSegs];
    int nEnded = 0;
    for (int k = 0; k < n; ++k) {
        hr_ctx::PassSlot &ps = c->slots[order[k]];
        if (!c->hasPassthrough && ps.step >= ps.nIter) {
            ps.active = false, ps.finished = true;
            ended[nEnded++] = &ps;
        } else {
            ps.step++;
        }
    }
    if (nEnded > 0) HIP_TRY(c, hipEventRecord(ended[nEnded - 1]->evFinal, G.stream)); // one event for the passes whose last stage this step was
    for (int k = 0; k < nEnded; ++k) ended[k]->finalEv = ended[nEnded - 1]->evFinal;
    if (c->hasPassthrough) { // snapshot of the queue lengths after this step, read two steps from now
        uint32_t *dst = G.hQCount + (size_t)ring * kMaxSlots * kMaxBounceSlots;
        HIP_TRY(c, hipMemcpy2DAsync(dst, sizeof(uint32_t) * kMaxBounceSlots, &c->dCounters[0].qCount[0], sizeof(Counters),
                                    sizeof(uint32_t) * kMaxBounceSlots, kMaxSlots, hipMemcpyDeviceToHost, G.stream));
        HIP_TRY(c, hipEventRecord(G.statusEv[ring], G.stream));
        G.statusUsed[ring] = true;
        for (int i = 0; i < kMaxSlots; ++i)
            G.statusOrder[ring][i] = (c->slots[i].active && c->slots[i].group == g) ? c->slots[i].order + 1ull : 0ull;
    }
    HIP_TRY(c, hipGetLastError());
    if (dbgT) fprintf(stderr, "step %llu (inject %d): begin %.1f us | injected +%.1f | counts known +%.1f | table built +%.1f | copy enqueued +%.1f | launched +%.1f\n", stepIdx, nInject, tA, tB - tA, tC - tB, tD - tC, tE - tD, nowUs() - tE);
    return resolveReady(c);
}

// Stages a pass occupies in the pipeline (depth+1 shaded stages + the last occlusion stage; in pass-through scenes two more
// until the host has seen that its queue ran empty — longer only for rays that really pass through surfaces).
static int stagesOf(const hr_ctx *c, const hr_pass_params &pp) { return pp.max_ray_depth + 2 + (c->hasPassthrough ? 2 : 0); }

// Advance the group that holds the oldest in-flight pass by one macro step (keeps passes finishing in order).
static int stepOldest(hr_ctx *c)
{
    const hr_ctx::PassSlot *oldest = nullptr;
    for (const hr_ctx::PassSlot &ps : c->slots)
        if (ps.active && (!oldest || ps.order < oldest->order)) oldest = &ps;
    if (!oldest) return resolveReady(c);
    return macroStep(c, oldest->group, 0);
}

static int slotLimit(const hr_ctx *c)
{
    int limit = c->maxSlots < c->tuneDepth ? c->maxSlots : c->tuneDepth; // passes in flight, all groups
    return limit < 1 ? 1 : (limit > kMaxSlots ? kMaxSlots : limit);
}

// Inject n pending passes into the next group (round robin), first making room for them.
static int injectBatch(hr_ctx *c, int n, int perGroupLimit)
{
    const int g = c->nextGroup;
    c->nextGroup = (g + 1) % c->nGroups;
    int guard = 0;
    while ((occupiedSlots(c, g) + n > perGroupLimit || occupiedSlots(c) + n > slotLimit(c)) && occupiedSlots(c) > 0) {
        int rc = stepOldest(c);
        if (rc) return rc;
        if (++guard > 64 * kMaxBounceSlots) FAIL(c, HR_ERR_DEVICE, "internal: pass pipeline did not make room");
    }
    return macroStep(c, g, n);
}

// Passes injected together when their camera rays travel as packets: a wave holds 2^k passes of 64 >> k pixels (hr_render.hip:
// k_raygen_packets), so the batch is the power of two next to the usual one (12 -> 16, 3 -> 4, 5 -> 4), sixteen per launch at most.
static int packetBatch(const hr_ctx *c)
{
    const int b = c->injectBatch < 1 ? 1 : c->injectBatch;
    int up = 1;
    while (up < b) up <<= 1;
    return (4 * b >= 3 * up) ? up : up / 2;
}
static int packetLog2(const hr_ctx *c)
{
    int k = 0;
    while ((2 << k) <= packetBatch(c) && k < 4) ++k;
    return k;
}
static bool packetsInUse(const hr_ctx *c) { return c->tunePackets == 1 || (c->tunePackets == 2 && c->packetsOn); }

// hr_ctx_desc::memory_budget: how many passes per step fit.  A pass of the batch holds, over the `stages` steps of its life, a pass buffer
// (S + 2 of them per batch pass are kept: the pipeline's depth and the resolve lag), its camera rays and hit records (scratch), and
// what each of its closest-hit stages emits (arena: two halves, each with a third of headroom).  A stage that has not been seen yet
// counts as long as it can possibly get (one ray per owned pixel: the guarantee); a stage that has, by the largest per-pass average a
// step carved for it, plus a tenth.  All stages of a batch are in flight at once (one generation per stage), so the sum over the
// stages is what one more pass per step costs.
static double budgetBytesPerPass(const hr_ctx *c, int stages)
{
    const double P = (double)(c->queueCapacity ? c->queueCapacity : 1u), kS = c->allLightsUsed ? 4.0 : 1.0;
    const double fb = (double)c->W * c->H * 16.0 * (c->allLightsUsed ? 4.0 : 1.0);
    double arena = 0.0, scratch = 0.0;
    for (int st = 0; st + 1 < stages && st < kMaxBounceSlots; ++st) { // (the last stage traces occlusion rays only)
        arena += c->stageSeen[st] ? 1.1 * c->stageArenaSeen[st] : P * (64.0 + 48.0 * kS);
        scratch += c->stageSeen[st] ? 1.1 * c->stageScratchSeen[st] : P * 20.0 + (st == 0 ? P * 64.0 : 0.0);
    }
    return (double)c->nGroups * ((double)(stages + 2) * fb + (4.0 / 3.0) * (2.0 * arena + scratch));
}
static int budgetBatch(const hr_ctx *c, int stages)
{
    if (!c->memBudget) return 1 << 20;
    const double fit = (double)c->memBudget / budgetBytesPerPass(c, stages);
    return fit < 1.0 ? 1 : (fit > 1e6 ? 1 << 20 : (int)fit);
}

static int batchFor(const hr_ctx *c, int stages)
{
    int batch = packetsInUse(c) ? packetBatch(c) : c->injectBatch;
    const int fit = budgetBatch(c, stages);
    if (batch > fit) {
        batch = fit;
        if (packetsInUse(c)) // (a packet holds a power of two of passes)
            while (batch & (batch - 1)) batch &= batch - 1;
    }
    int perGroup = slotLimit(c) / c->nGroups;
    if (perGroup > kMaxSegs) perGroup = kMaxSegs;
    if (batch * stages > perGroup) batch = perGroup / stages;
    return batch < 1 ? 1 : batch;
}

static int drainPipeline(hr_ctx *c)
{
    while (!c->pendingInject.empty()) {
        const int stages = stagesOf(c, c->pendingInject.front());
        const int batch = batchFor(c, stages);
        int n = (int)c->pendingInject.size() < batch ? (int)c->pendingInject.size() : batch;
        // the last, partly filled batch of a run on several pipeline groups is dealt out over the groups (each group's dependent
        // chain of stages then carries a share of it, and the chains overlap on the device) instead of going to one of them whole
        if (c->nGroups > 1 && (int)c->pendingInject.size() <= batch) {
            const int idleGroups = c->nGroups - (c->nextGroup % c->nGroups);
            const int share = ((int)c->pendingInject.size() + idleGroups - 1) / (idleGroups > 0 ? idleGroups : 1);
            n = share < 1 ? 1 : share;
        }
        // With the camera rays as packets a step injects WHOLE launches of kMaxBatch passes where it can: a remainder goes in one step later,
        // where its (smaller, less coherent) packets run beside the k_trace that carries the first launches' first bounce instead of
        // lengthening the step that has nothing beside it.  A 1/8 shard's 20 passes as 16, then 4: 0.304 -> 0.294 ms/step; as 12 + 8,
        // 10 + 10, 8 + 8 + 4 (each of them smaller packets all round): 0.307 - 0.323 (profiles/r5g_burst_pmin.txt).
        if (packetsInUse(c) && n > kMaxBatch && n % kMaxBatch) n -= n % kMaxBatch;
        int perGroup = batch * stages;
        int rc = injectBatch(c, n, perGroup);
        if (rc) return rc;
    }
    int guard = 0;
    while (activePasses(c) > 0) {
        int rc = stepOldest(c);
        if (rc) return rc;
        if (++guard > 64 * kMaxBounceSlots) FAIL(c, HR_ERR_DEVICE, "internal: pass pipeline did not drain");
    }
    int rc = resolveReady(c);
    if (rc) return rc;
    if (occupiedSlots(c) > 0) FAIL(c, HR_ERR_DEVICE, "internal: finished passes left unresolved");
    c->oldestWaitingNs = 0;
    if (c->probeGuard) { // (a probe nobody has waited for: whatever follows on the caller's stream — frees after a synchronise included — comes after it)
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evProbeB, 0));
        c->probeGuard = false;
    }
    // whatever the caller does next on its stream (clear, scene edits, new tables) has to be seen by the groups
    for (int g = 0; g < kMaxGroups; ++g) c->groups[g].needUserSync = true;
    return HR_OK;
}

int hr_render_pass(hr_ctx *c, const hr_pass_params *pp)
{
    ENTER(c);
    if (!pp) FAIL(c, HR_ERR_INVALID, "null params");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    if (!c->committed) FAIL(c, HR_ERR_INVALID, "scene not committed");
    if (c->nSeq <= 0 || c->nSeqOffsets <= 0) FAIL(c, HR_ERR_INVALID, "sample tables not set");
    if (pp->max_ray_depth < 0 || pp->max_ray_depth + 2 >= kMaxBounceSlots - 8) FAIL(c, HR_ERR_INVALID, "max_ray_depth out of range");
    if (pp->interactive_mode && (pp->block_size[0] <= 0 || pp->block_size[1] <= 0)) FAIL(c, HR_ERR_INVALID, "bad block size");
    int rc = HR_OK;
    if (pp->estimator == HR_ESTIMATOR_ENV_MIS || pp->estimator == HR_ESTIMATOR_ALL_LIGHTS) {
        rc = ensureEnvTable(c);
        if (rc) return rc;
        if (pp->estimator == HR_ESTIMATOR_ALL_LIGHTS && !c->allLightsUsed) {
            // pass slots grow (a second occlusion ray per path, a second partial sum per pass): the existing ones are released and
            // re-allocated below with the new sizes
            rc = drainPipeline(c);
            if (rc) return rc;
            QUIESCE(c);
            const uint32_t keepCap = c->queueCapacity;
            freeQueues(c);
            c->queueCapacity = keepCap;
            c->allLightsUsed = true;
            slotBudget(c);
        }
    } else if (pp->estimator != HR_ESTIMATOR_REFERENCE) {
        FAIL(c, HR_ERR_INVALID, "unknown estimator");
    }
    if (pp->texture_lod == HR_TEXTURE_LOD_CONE) {
        rc = ensureTextureLod(c);
        if (rc) return rc;
        c->textureLodUsed = true;
    } else if (pp->texture_lod != HR_TEXTURE_LOD_BASE) {
        FAIL(c, HR_ERR_INVALID, "unknown texture_lod mode");
    }
    rc = uploadScene(c); // drains the pipeline first when the scene constants changed
    if (rc) return rc;
    if (c->frame.nOwnedTiles == 0) return HR_OK;
    // only passes of equal depth overlap (keeps the groups in lockstep; order is enforced by resolveReady regardless)
    if (pp->max_ray_depth != c->lastDepth && (occupiedSlots(c) > 0 || !c->pendingInject.empty())) {
        rc = drainPipeline(c);
        if (rc) return rc;
    }
    c->lastDepth = pp->max_ray_depth;
    if (c->memBudget && (double)c->memBudget < budgetBytesPerPass(c, stagesOf(c, *pp))) {
        c->err = "hr_ctx_desc.memory_budget (" + std::to_string(c->memBudget >> 20) + " MiB) is less than one pass per pipeline step needs at " + std::to_string(c->W) + "x" +
                 std::to_string(c->H) + ", depth " + std::to_string(pp->max_ray_depth) + ": " + std::to_string((unsigned long long)budgetBytesPerPass(c, stagesOf(c, *pp)) >> 20) + " MiB";
        return HR_ERR_INVALID;
    }
    {
        // All pass slots this depth needs are allocated up front, on the first pass (hipMalloc synchronises the device and
        // takes ~0.1 ms per buffer: allocating slot by slot as the pipeline filled stalled the first 20-odd passes of a render)
        const int stagesNow = stagesOf(c, *pp);
        const int batchNow = batchFor(c, stagesNow);
        int want = c->nGroups * batchNow * stagesNow + 2 * c->nGroups * batchNow;
        if (want > slotLimit(c)) want = slotLimit(c);
        for (int i = 0; i < kMaxSlots && c->nSlotsAllocated < want; ++i)
            if (!c->slots[i].allocated) {
                rc = allocSlot(c, c->slots[i]);
                if (rc) return rc;
            }
    }
    c->pendingInject.push_back(*pp);
    if (c->oldestWaitingNs == 0)
        c->oldestWaitingNs = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    // a macro step is launched once enough passes are waiting to fill it; each group holds batch x stages passes
    const int stages = stagesOf(c, *pp);
    const int batch = batchFor(c, stages);
    if ((int)c->pendingInject.size() < batch) return HR_OK;
    return injectBatch(c, batch, batch * stages);
}

int hr_frame_pass_batch(hr_ctx *c, int32_t max_ray_depth, int32_t *batch)
{
    ENTER(c);
    if (!batch || max_ray_depth < 0) FAIL(c, HR_ERR_INVALID, "bad arguments");
    if (c->W <= 0) FAIL(c, HR_ERR_INVALID, "no frame");
    hr_pass_params pp{};
    pp.max_ray_depth = max_ray_depth;
    *batch = batchFor(c, stagesOf(c, pp));
    return HR_OK;
}

int hr_flush(hr_ctx *c)
{
    ENTER(c);
    const int rc = drainPipeline(c);
    return rc ? rc : overflowCheck(c); // (no wait here: what the kernels have reported so far)
}

int hr_get_stats(hr_ctx *c, hr_pass_stats *out)
{
    ENTER(c);
    if (!out) FAIL(c, HR_ERR_INVALID, "null output");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    std::vector<Stats> parts(kStatSlots);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats) * kStatSlots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    {
        const int rc = overflowCheck(c);
        if (rc) return rc;
    }
    Stats s{};
    for (const Stats &p : parts) {
        s.paths += p.paths, s.raysClosest += p.raysClosest, s.raysAny += p.raysAny, s.shadedHits += p.shadedHits;
        s.accumulates += p.accumulates, s.nodeVisits += p.nodeVisits, s.triTests += p.triTests;
        s.nodeVisitsAny += p.nodeVisitsAny, s.triTestsAny += p.triTestsAny;
    }
    std::memset(out, 0, sizeof(*out));
    out->paths = s.paths, out->rays_closest = s.raysClosest, out->rays_any = s.raysAny, out->shaded_hits = s.shadedHits;
    out->accumulates = s.accumulates, out->node_visits = s.nodeVisits, out->tri_tests = s.triTests;
    out->node_visits_any = s.nodeVisitsAny, out->tri_tests_any = s.triTestsAny;
    return HR_OK;
}

int hr_get_kernel_times(hr_ctx *c, hr_kernel_times *out)
{
    ENTER(c);
    if (!out) FAIL(c, HR_ERR_INVALID, "null output");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    c->drainTimes();
    for (int k = 0; k < HR_KERNEL_COUNT; ++k) out->ms[k] = c->kernelMs[k], out->launches[k] = c->kernelLaunches[k];
    std::vector<Stats> parts(kStatSlots);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats) * kStatSlots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long ticks = 0, launches = 0;
    for (const Stats &p : parts) ticks += p.traceTicks, launches += p.traceLaunches;
    out->trace_clock_ms = (float)((double)ticks * 1e-5); // 100 MHz: 10 ns per tick
    out->trace_clock_launches = (uint32_t)launches;
    out->camera_packets = packetsInUse(c) ? (uint32_t)packetBatch(c) : 0u;
    out->packet_union = (float)c->lastUnion;
    return HR_OK;
}

int hr_get_step_log(hr_ctx *c, hr_step_record *out, int32_t capacity, int32_t *n_records)
{
    ENTER(c);
    if (!out || !n_records || capacity <= 0) FAIL(c, HR_ERR_INVALID, "bad arguments");
    {
        int rc = drainPipeline(c);
        if (rc) return rc;
    }
    std::vector<Stats> parts(1);
    std::vector<unsigned long long> log(3 * (size_t)kStepLogCap);
    HIP_TRY(c, hipMemcpyAsync(parts.data(), c->dStats, sizeof(Stats), hipMemcpyDeviceToHost, c->stream)); // (k_shade_sort's first thread counts in the first copy)
    HIP_TRY(c, hipMemcpyAsync(log.data(), c->dStepLog, log.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned long long total = parts[0].traceLaunches;
    const unsigned long long first = total > (unsigned long long)kStepLogCap ? total - (unsigned long long)kStepLogCap : 0ull;
    // Records are appended when a step's k_trace has ENDED (k_shade_sort writes them), so with several pipeline groups they arrive out of
    // start order: they are handed out sorted by start, each with its group.
    std::vector<const unsigned long long *> recs;
    for (unsigned long long i = first; i < total; ++i) recs.push_back(&log[3 * (size_t)(i % (unsigned long long)kStepLogCap)]);
    std::stable_sort(recs.begin(), recs.end(), [](const unsigned long long *a, const unsigned long long *b) { return a[0] < b[0]; });
    int32_t n = 0;
    for (const unsigned long long *rec : recs) {
        if (n >= capacity) break;
        out[n].start_ms = (double)(rec[0] - recs[0][0]) * 1e-5; // 100 MHz device clock
        out[n].trace_ms = (float)((double)(rec[1] - rec[0]) * 1e-5);
        out[n].passes_in_flight = (int32_t)(rec[2] & 0xFFFFull), out[n].group = (int32_t)((rec[2] >> 16) & 0xFFull), out[n].passes_injected = (int32_t)(rec[2] >> 32);
        ++n;
    }
    *n_records = n;
    return HR_OK;
}

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
