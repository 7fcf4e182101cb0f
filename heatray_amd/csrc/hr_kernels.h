// hr_kernels.h — host-visible declarations of the kernel launchers (hr_render.hip, hr_build.hip).
#pragma once

#include "hr_types.h"

#include <cstddef>

namespace hr {

struct FrameDev {
    float *fb; // RGBA32F accumulation buffer, row 0 = bottom
    int32_t W, H;
    int32_t rank, world, tile, tilesX, tilesY, nOwnedTiles;
};

struct HitRec; // hr_trace.h

// Device counters are kept in kStatSlots copies (a workgroup adds to slot blockIdx.x % kStatSlots) so that the adds of a
// launch spread over many addresses instead of serialising on one; hr_get_stats sums the copies.
static const int kStatSlots = 4096;
struct Stats {
    unsigned long long paths, raysClosest, raysAny, shadedHits, accumulates, nodeVisits, triTests, nodeVisitsAny, triTestsAny;
    unsigned long long traceTicks, traceLaunches; // k_trace by the 100 MHz device clock: first workgroup's start to last workgroup's end, summed over launches
};

struct LaunchCfg {
    hipStream_t stream;
    int numCUs;
    int traceBlocksPerCU;
    int shadeBlocksPerCU;
    bool collectStats;
    bool textureLod; // some pass has asked for HR_TEXTURE_LOD_CONE: launch the shading kernel that carries the trilinear sampler
    bool allLights;  // some pass has asked for HR_ESTIMATOR_ALL_LIGHTS: the shading kernel that can emit two occlusion rays per vertex
    bool hasGlass;   // some material of the scene is glass: the glass shading kernel is launched too
    int packetSwizzle = 0; // k_raygen_packets deals whole 32x32 tiles to the XCDs instead of consecutive 16-pixel patches (HR_TUNE pswz)
};

// One in-flight pass as seen by the kernels of one macro step.
struct SegDev {
    RayQueue qin;        // closest-hit rays traced and shaded this step
    RayQueue qout;       // closest-hit rays emitted for the next step
    ShadowQueue sqIn;    // occlusion rays traced this step (emitted by the previous step's shade)
    ShadowQueue sqOut;   // occlusion rays this step's shade emits (traced by the next step)
    HitRec *hits;        // closest-hit records of qin
    float *passbuf;      // RGBA32F sample of this pass (full-frame indexing)
    float *passbufB;     // HR_ESTIMATOR_ALL_LIGHTS: the sample's second partial sum (analytic-light contributions), or null
    uint32_t *qCountIn;  // number of rays in qin
    uint32_t *sCountIn;  // number of occlusion rays to trace this step
    uint32_t *qCountOut; // counter shade appends qout with
    uint32_t *sCountOut; // counter shade appends sq with
    uint32_t *hitIdx;    // hit list of this step: indices into qin of the rays that hit a PBR material from the front, a glass material from the back
    uint32_t *pCount;    // entries at the front of hitIdx (k_shade_sort appends, k_shade_hit<.., 0> reads)
    uint32_t *gCount;    // entries at the back of hitIdx
    uint32_t hitCap;     // capacity of hitIdx (= of the ray queues)
    uint32_t packets;    // the entry's closest-hit queue (camera rays) is filled AND traced by k_raygen_packets this step: k_trace leaves it out; 2: that kernel runs beside k_trace (the queue's length is not final when k_trace starts)
    hr_pass_params pp;   // per-pass uniforms
    int32_t closestEnabled; // 0 in a pass's last step (only its occlusion rays remain)
    // Capacities of the queues beside hitCap (which also bounds hits / hitIdx / qout): every append compares its slot with them and every
    // reader clamps the counter it reads — a bound that turns out wrong drops rays and raises StepTable::hostOverflow instead of
    // writing past an arena (hr_render.hip: queueOverflow)
    uint32_t qinCap;  // rays qin can hold
    uint32_t sInCap;  // occlusion rays sqIn can hold
    uint32_t sOutCap; // occlusion rays sqOut can hold
};

#ifndef HR_MAX_SEGS
#define HR_MAX_SEGS 320
#endif
static const int kMaxSegs = HR_MAX_SEGS;
static const int kClkSlots = 16;
static const int kStepLogCap = 4096; // macro steps the step log keeps (a ring)
static const int kTraceHeadsMax = 64; // in-flight passes of one group: (passes injected per macro step) x (stages per pass)
struct StepTable {
    // read-mostly header: every wave of every kernel of the step reads it once
    int32_t nSeg;
    int32_t refillLanes;   // refill a wave from the work pool once this many lanes are idle
    int32_t triPhaseLanes; // run the triangle phase once this many lanes are blocked on a postponed leaf
    int32_t fetchMax;      // work items a wave reserves per global atomic while plenty of work is left ...
    int32_t fetchMin;      // ... shrinking to this near the end of the pool (guided self-scheduling: short tail)
    int32_t staticPerWave; // launches of at most this many rays per resident wave are dealt out statically (k_trace)
    int32_t hasGlass;      // some material of the scene is glass (else the glass hit list is never appended to)
    int32_t primaryFromSeg;  // passes [primaryFromSeg, nSeg) were injected this step: their closest-hit queues hold camera rays
    int32_t fetchMaxPrimary; // chunk size of the work fetch inside that (coherent) part of the index space: low 16 bits; high 16 bits: how many such chunks per resident wave the part must hold for it to be used
    uint32_t headsLog2;      // 2^headsLog2 ranges are in use (HR_TUNE heads=)
    // k_trace's first workgroup reports the closest-hit queue length of every entry to pinned HOST memory when it starts, then the
    // step's number (hr_core.hip, Group::hCounts): how the host sizes the next steps' queues without a packet on the stream
    uint32_t *hostCounts;
    unsigned long long *hostSeq;
    unsigned long long seqValue;
    // log of the k_trace launches since the last hr_clear (hr_get_step_log): k_shade_sort appends (first start, last end) by the device clock
    unsigned long long *stepLog; // kStepLogCap x {start tick, end tick, passes in the table | group << 16 | passes injected << 32}
    uint32_t nInjectedNow;
    uint32_t group; // pipeline group the step belongs to (goes into its log record)
    // k_packet_probe adds (children a packet entered x its rays, child boxes the rays themselves entered, 1 per finished wave) to
    // probe[0..2]; k_trace's first workgroup copies the totals to pinned host memory with the queue lengths (hr_core.hip: the packet selector)
    unsigned long long *probe;
    unsigned long long *hostProbe;
    // Sticky overflow report (pinned host memory, four words: queue kind, the step's number, table entry, count seen): set by the first
    // append or read that finds a queue longer than its capacity; hr_flush / hr_readback / hr_synchronize turn it into HR_ERR_DEVICE
    uint32_t *hostOverflow;
    uint32_t *hostCameraCount; // pinned: camera rays per pass behind the root cull, as the step's shading kernel last saw them in queues a packet kernel filled (a hint for the host's scheduling, no ordering)
    // The work cursors of k_trace: the index space of a launch is cut into 2^headsLog2 equal ranges (32 by default, at most
    // kTraceHeadsMax), each with a cursor on a cache line of its own — none of them shares its 128-byte line with the header above
    // or with seg[] below, which every wave reads while the cursors are hammered by atomics.  ONE cursor serialises at ~12 ns per atomic:
    // a launch of 25 M camera rays in 64-ray chunks needs 390 k of them, 4.7 ms of a 10 ms launch.  A workgroup starts on range
    // (blockIdx mod the number of ranges) — workgroups are dealt to the 8 XCDs round robin, so with 32 ranges an XCD starts on four of
    // them — and moves on to the next range when its own is used up.  Zeroed on the host with every table upload.
    alignas(128) uint32_t heads[kTraceHeadsMax * 32];
    // k_trace times itself with the device's constant 100 MHz clock: every workgroup folds its start and end into one of kClkSlots
    // (min, max) pairs, and the step's next kernel (k_shade_sort) adds (max end - min start) to the Stats counters.  No event packets
    // on the stream (a record costs a dependent chain of small launches 4-8 us each).  Initialised with every table upload.
    alignas(128) unsigned long long clkStart[kClkSlots];
    unsigned long long clkEnd[kClkSlots];
    alignas(128) SegDev seg[kMaxSegs];
};
static_assert(offsetof(StepTable, heads) % 128 == 0 && offsetof(StepTable, seg) % 128 == 0, "k_trace's work cursors own their cache lines");

// ---- hr_render.hip
// Several passes per launch (a small shard injects / resolves a batch of passes per macro step: one launch instead of 8 + 8)
static const int kMaxBatch = 16;
struct SegList {
    int32_t n;
    int32_t seg[kMaxBatch]; // indices into StepTable::seg, one per blockIdx.y
};
struct PassBufList {
    int32_t n;
    const float *buf[kMaxBatch]; // pass samples, added to the frame in this order
    const float *bufB[kMaxBatch]; // second partial sum of a pass sample (HR_ESTIMATOR_ALL_LIGHTS), or null
};
// the per-stage counters of the passes injected by one macro step, cleared with ONE launch (a hipMemsetAsync per pass is a fill kernel
// and a kernel boundary each: 12 of them delayed a step's ray generation by ~0.1 ms)
struct CounterList {
    int32_t n;
    Counters *ctr[kMaxBatch];
};
void launchZeroCounters(const LaunchCfg &cfg, const CounterList &list);
void launchFetchTable(hipStream_t stream, const void *hostMapped, void *dst, size_t bytes);
void launchRaygen(const LaunchCfg &cfg, const SceneDev *S, const StepTable *tbl, const SegList &segs, const FrameDev &fr, Stats *stats);
void launchResolve(const LaunchCfg &cfg, const FrameDev &fr, const PassBufList &bufs);
void launchPackOwned(const LaunchCfg &cfg, const FrameDev &fr, const float *frame, float *packed, int unpack, float *full);
void launchDisplay(const LaunchCfg &cfg, const FrameDev &fr, const hr_display_params &P, int format, void *out);
void launchTrace(const LaunchCfg &cfg, const SceneDev *S, const int *leafKeys, const Node32 *nodes32, const Tri *tris, StepTable *tbl, Stats *stats);
void launchRaygenPackets(const LaunchCfg &cfg, const SceneDev *S, const Node4 *nodes, const Tri *tris, const StepTable *tbl, const SegList &segs,
                         const FrameDev &fr, Stats *stats, bool uniformParams); // segs.n: a power of two; uniformParams: the passes differ in sample_index only
int launchPacketProbe(hipStream_t stream, const SceneDev *S, const Node4 *nodes, const Tri *tris, const hr_pass_params &pp, int passesLog2, const FrameDev &fr,
                      unsigned long long *probe);
void launchShadowProbe(hipStream_t stream, const SceneDev *S, const Node4 *nodes, const Tri *tris, const StepTable *tbl, const SegList &segs, uint32_t maxRays,
                       unsigned long long *probe); // measurement only (HR_TUNE sprobe=)
void launchShade(const LaunchCfg &cfg, const SceneDev *S, const StepTable *tbl, Stats *stats);
void launchDebugTrace(const LaunchCfg &cfg, const SceneDev *S, int n, const float *o, const float *d, const float *tmax, const int *skip,
                      int anyHit, hr_hit *out);
size_t hitRecordSize();

// ---- hr_build.hip
// Per-geometry descriptor for the assemble kernel; all pointers are device pointers.  Attributes are addressed with a stride
// (in floats), exactly as the caller handed them over (rlVertexAttribBuffer's stride, Mesh.cpp:104-132): nothing is
// de-interleaved on the host.
struct GeomDev {
    const float *pos, *nrm, *uv, *tan, *bit, *col; // may be null (uv..col)
    int32_t posStride, nrmStride, uvStride, tanStride, bitStride, colStride; // floats between consecutive vertices
    const uint32_t *idx;
    uint32_t triOffset; // first global triangle (prim id) of this geometry
    uint32_t nTris;
    int32_t strip;
    uint32_t flags;    // TF_FRONT_CW | TF_NON_OCCLUDER | TF_HAS_*
    uint32_t material;
    float world[16];
};

static const int kBoundSlots = 64; // copies of the scene bounds the assemble kernel reduces into (6 ordered uints each)
struct Box6 {
    float lo[3], hi[3];
};

// Scene constants derived from the bounds on the device (so that a refit needs no host round trip before its kernels):
// diag = |hi - lo| with the contract's operation order, pad = 1e-5 diag (leaf boxes), eps = 1e-4 diag (ray epsilon)
struct SceneConsts {
    float lo[3], hi[3];
    float diag, pad, eps;
    float areaSum; // sum of the node boxes' surface areas after the last build / refit (tree-quality heuristic only)
    float triAreaSum; // sum of the triangles' areas (invariant under rigid motion, scales with the scene: what areaSum is compared with)
    uint32_t pad1;
};

static const int kMaxLevels = 64;
struct BuildResult {
    Node4 *nodes;
    Node32 *nodes32;      // the same nodes in 32 bytes (k_trace), made by encodeNodes32 after every build / refit
    int *leafKeys;        // Node4::c.w of every node, compact (k_trace finds a leaf child's triangle through it)
    Tri *tris;            // leaf order
    Box6 *nodeBox;        // float box of every node (a refit passes child boxes upwards through it)
    uint32_t *slotOfPrim; // prim id -> position in `tris`
    int32_t nNodes, rootLeafCount;
    int32_t levels; // levels of 4-wide inner nodes (the traversal stack holds at most 3 entries per level)
    uint32_t triSlots;    // entries of `tris` (32-byte nodes: 4 per node, some unused)
    uint32_t levelStart[kMaxLevels + 1]; // nodes of level L are [levelStart[L], levelStart[L + 1]) (breadth-first allocation)
    int32_t builder;            // which binary tree was collapsed: 0 the radix tree over Morton codes (LBVH), 1 PLOC
    float costRadix, costPloc;  // summed surface area of the 4-wide nodes over the root's, per candidate (0: not built)
};
struct BuildOptions {
    int ploc;       // 0: radix tree only; 1: build both, keep the cheaper collapse; 2: PLOC whenever it fits the traversal stack
    int plocRadius; // search radius of the nearest-neighbour step
    int maxLevels;  // levels of 4-wide nodes the traversal stack can hold (a PLOC tree deeper than that is not used)
};

// scene bounds as float-ordered uints: lo.xyz, hi.xyz.  slotOfPrim != null: triangles go to their leaf slot (refit), else prim order.
void launchAssemble(hipStream_t st, const GeomDev *geoms, int nGeoms, uint32_t nTris, Tri *tris, const uint32_t *slotOfPrim, TriAttr *attrs,
                    TriAttrExt *ext, uint32_t *boundsOrdered);
// bounds (ordered uints) -> SceneConsts, and the ray epsilon straight into the device scene block
void launchSceneConsts(hipStream_t st, const uint32_t *boundsOrdered, SceneConsts *out, SceneDev *scene);
// Full LBVH build from assembled triangles (prim order); allocates scratch internally; returns device arrays (hipMalloc).
int buildLBVH(hipStream_t st, const Tri *trisPrimOrder, uint32_t nTris, const float lo[3], const float hi[3], float pad,
              const SceneConsts *deviceConsts, BuildResult *out, const BuildOptions &opt);
// Refit: the tree keeps its topology; every node's child boxes are recomputed bottom-up from the triangles in `tree.tris`
// (already moved by launchAssemble) and re-quantised.  Level by level, no host synchronisation.
void refitLBVH(hipStream_t st, const BuildResult &tree, uint32_t nTris, SceneConsts *consts);
// Node32 of every node from its Node4 (children, counts), the node boxes and the triangles; also writes the grid into *scene (device) when given.
// The grid is a function of the scene bounds in *consts: gridOf() gives the host the same numbers.
void encodeNodes32(hipStream_t st, const BuildResult &tree, const SceneConsts *consts, SceneDev *scene);
void gridOf(const SceneConsts &k, float gridLo[3], float gridCell[3], uint32_t gridCellExp[3]);
// order-independent 64-bit digest of a device buffer, added to *out (device)
void launchMipChain(hipStream_t st, const TexDesc &t, int nLevels, float *mips);
void launchTexLodScale(hipStream_t st, TexDesc *table, int n);
void launchTexDensity(hipStream_t st, const Tri *leafTris, uint32_t nSlots, const TriAttr *attrs, float *out);
void launchHashWords(hipStream_t st, const void *words, size_t nWords, unsigned long long seed, unsigned long long *out);
void launchAreaSum(hipStream_t st, const Box6 *nodeBox, uint32_t n, SceneConsts *consts);
void launchTriAreaSum(hipStream_t st, const Tri *leafTris, uint32_t nSlots, SceneConsts *consts);

// environment importance table (HR_ESTIMATOR_ENV_MIS): scratch and outputs are caller-owned device arrays
void launchEnvGuides(hipStream_t st, const float *rowCdf, const float *colCdf, int w, int h, uint16_t *rowGuide, uint16_t *colGuide);
void launchEnvTable(hipStream_t st, const TexDesc &tex, float *lum, float *dil, uint32_t *wq, unsigned long long *rowSum, unsigned long long *total,
                    uint32_t *maxBits, float *rowCdf, float *colCdf, float *prob, float *meanLum);

void launchQmc(hipStream_t st, int mode, uint32_t sequenceIndex, uint32_t count, float2 *out);
// the sequential generators (hr_tables.h): nSeq tables of `count` points, table s at out + s * stride.  edges == 0: util::uniformRandomFloats
// with seed seed0 + s, else util::randomPolygonal over the polygon (vx, vy)[edges]; launchBlueNoise: util::blueNoise of sequence seq0 + s,
// cand = scratch of nSeq * (count - 1) * 30 points
void launchMtTables(hipStream_t st, uint32_t seed0, int nSeq, uint32_t count, uint32_t edges, const float *vx, const float *vy, float2 *out, size_t stride);
void launchBlueNoise(hipStream_t st, int32_t seq0, int nSeq, uint32_t count, float2 *out, size_t stride, float2 *cand);
void launchMultiscatterLUT(hipStream_t st, const float2 *sobol4096, float *out128x128);

} // namespace hr
