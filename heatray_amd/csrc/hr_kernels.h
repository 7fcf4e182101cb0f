// hr_kernels.h — host-visible declarations of the kernel launchers (hr_render.hip, hr_build.hip).
#pragma once

#include "hr_types.h"

namespace hr {

struct FrameDev {
    float *fb; // RGBA32F accumulation buffer, row 0 = bottom
    int32_t W, H;
    int32_t rank, world, tile, tilesX, tilesY, nOwnedTiles;
};

struct Stats {
    unsigned long long paths, raysClosest, raysAny, shadedHits, accumulates, nodeVisits, triTests, nodeVisitsAny, triTestsAny;
};

struct LaunchCfg {
    hipStream_t stream;
    int numCUs;
    int traceBlocksPerCU;
    int shadeBlocksPerCU;
    bool collectStats;
};

// ---- hr_render.hip
void launchRaygen(const LaunchCfg &cfg, const SceneDev *S, const hr_pass_params &pp, const FrameDev &fr, RayQueue q, Counters *ctr, Stats *stats);
void launchTraceClosest(const LaunchCfg &cfg, const SceneDev *S, RayQueue q, void *hits, Counters *ctr, Stats *stats, int slot);
void launchTraceShadow(const LaunchCfg &cfg, const SceneDev *S, ShadowQueue sq, float *fb, Counters *ctr, Stats *stats, int slot);
void launchShade(const LaunchCfg &cfg, const SceneDev *S, const hr_pass_params &pp, float *fb, RayQueue qin, const void *hits, RayQueue qout,
                 ShadowQueue sq, Counters *ctr, Stats *stats, int slot);
void launchDebugTrace(const LaunchCfg &cfg, const SceneDev *S, int n, const float *o, const float *d, const float *tmax, const int *skip,
                      int anyHit, hr_hit *out);
size_t hitRecordSize();

// ---- hr_build.hip
// Per-geometry descriptor for the assemble kernel; all pointers are device pointers.
struct GeomDev {
    const float *pos, *nrm, *uv, *tan, *bit, *col; // tightly packed, may be null (uv..col)
    const uint32_t *idx;
    uint32_t triOffset; // first global triangle (prim id) of this geometry
    uint32_t nTris;
    int32_t strip;
    uint32_t flags;    // TF_FRONT_CW | TF_NON_OCCLUDER | TF_HAS_*
    uint32_t material;
    float world[16];
};

struct BuildResult {
    Node *nodes;
    Tri *tris;
    int32_t nNodes, rootLeafCount;
};

// scene bounds as float-ordered uints: lo.xyz, hi.xyz
void launchAssemble(hipStream_t st, const GeomDev *geoms, int nGeoms, uint32_t nTris, Tri *trisPrimOrder, TriAttr *attrs, TriAttrExt *ext,
                    uint32_t *boundsOrdered);
// Full LBVH build from assembled triangles; allocates scratch internally; returns device arrays (hipMalloc).
int buildLBVH(hipStream_t st, const Tri *trisPrimOrder, uint32_t nTris, const float lo[3], const float hi[3], float pad, BuildResult *out);

void launchQmc(hipStream_t st, int mode, uint32_t sequenceIndex, uint32_t count, int radial, float2 *out);
void launchMultiscatterLUT(hipStream_t st, const float2 *sobol4096, float *out128x128);

} // namespace hr
