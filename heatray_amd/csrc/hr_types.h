// hr_types.h — device-resident data layouts of libhrcore (DESIGN.md §Data layout in HBM).
#pragma once

#include "../../include/hrcore.h"
#include "hr_math.h"

namespace hr {

// ---- acceleration structure -------------------------------------------------------------------
// 4-wide BVH node, child boxes quantised to 8 bits per plane against the node's own box: one 64-byte,
// 64-byte-aligned record (never straddles a cache line) of which THREE dwordx4 loads are used.
//   a = (origin.x, origin.y, origin.z, meta)    meta = ex | ey << 8 | ez << 16 | nInner << 24 | nValid << 27,
//                                               scale_k = as_float(e_k << 23)
//   b = (qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4])   byte j of each dword belongs to child j
//   c = (qhi.y[4], qhi.z[4], innerBase, leafKey)
//   d = unused
// child box = origin + q * scale, with qlo rounded down and qhi rounded up (conservative).
// Children 0 .. nInner-1 are the inner nodes innerBase + j: the children of one node are allocated together, so
// siblings — which a ray tends to visit together — share 128-byte cache lines.  Children nInner .. nValid-1 are single
// triangles, also stored together, in descending order of the slot: the reference of child j is ~(triangle index) = leafKey + j.
// A reference >= 0 is a node index; < 0 is a leaf ~(first | (count-1) << 28) (count > 1 only for a root leaf).
struct alignas(64) Node4 {
    float4 a;
    uint4 b, c, d;
};


// The same 4-wide node in 32 bytes = TWO dwordx4 loads, for k_trace (the texture addresser — one lane per cycle per load instruction
// whatever its width — is what binds that kernel on long walks; the packet kernel reads Node4 through the scalar cache and keeps it):
//   p = (qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4])   q = (qhi.y[4], qhi.z[4], w6, w7)      byte j of a plane dword belongs to child j
//   w6 = innerBase | nInner << 25 | ex << 28            w7 = gx | gy << 8 | gz << 16 | ey << 24 | ez << 28
// The frame is origin + q * scale with origin_k = gridLo_k + g_k * cell_k on a scene-wide 256^3 grid of power-of-two cells
// (SceneDev::grid*) and scale_k = cell_k * 2^(e_k - 8), an exponent per axis (flat nodes of a mesh keep tight planes across their thin
// side): 8-bit planes as in Node4 (the cheap v_cvt_f32_ubyteN decode, which also unpacks the grid bytes), a frame of 2^e_k cells
// per axis.  A slot without a child holds the inverted planes lo = 255, hi = 0 and can never be entered (no child count needed).
// Children and their order are Node4's; a leaf child's triangle is found through Node4::c.w of the same node when the leaf is tested
// (its reference on the stack is ~(4 * node + 3 - slot)).  innerBase has 25 bits: 33 M nodes, ~95 M triangles (checked at commit).
struct alignas(32) Node32 {
    uint4 p, q;
};

// World-space triangle in BVH leaf order, 48 bytes = three dwordx4 loads:
//   p = (v0.x v0.y v0.z e1.x)  q = (e1.y e1.z e2.x e2.y)  r = (e2.z, prim id, flags, -)
struct alignas(16) Tri {
    float4 p, q, r;
};

// Shading attributes per triangle in SUBMISSION order (prim id), 64 bytes:
//   n0 n1 n2 (world normals, 9 floats) uv0 uv1 uv2 (6 floats) matflags (material id | TF_* << 24)
struct alignas(64) TriAttr {
    float n[9];
    float uv[6];
    uint32_t matflags;
};
// Optional per-triangle tangent / bitangent / colour varyings (only allocated when a mesh has them)
struct alignas(16) TriAttrExt {
    float tan[9], bit[9], col[9];
    float pad;
};

enum : uint32_t { TF_FRONT_CW = 1u, TF_NON_OCCLUDER = 2u, TF_HAS_UV = 4u, TF_HAS_TANGENTS = 8u, TF_HAS_COLORS = 16u };
static const uint32_t kMatMask = 0x00FFFFFFu;

// ---- textures ---------------------------------------------------------------------------------
struct TexDesc {
    const void *px; // w*h*c floats or bytes (dtype), row 0 = bottom
    int32_t w, h, c;
    int32_t wrapS, wrapT, filter;
    int32_t dtype;  // HR_TEX_F32, or HR_TEX_U8: bytes stay bytes in HBM and are normalised on fetch as float(byte) / 255.0f
    int32_t nLevels; // 0: no mip chain built; otherwise levels 0 .. nLevels-1 exist (HR_TEXTURE_LOD_CONE, hr_texture.h)
    const float *mips; // levels 1 .. nLevels-1, always f32, c channels, level l is max(1, w >> l) x max(1, h >> l), stored one after the other
    float lodScale;  // 0.5 * log2(w * h): texels per unit uv length, as a level offset
    int32_t pad;
};

// ---- ray queues (structure of arrays, one float4 / int4 stream per field group) -----------------
//   A = (origin.xyz, tmax)   B = (dir.xyz, extraT)   C = (weight.xyz, pixel bits)
//   D = (meta, sequenceIndexOffset, srcPrim, -)
//   meta = sequenceID | depth << 8 | missKind << 24 | missIdx << 27
struct RayQueue {
    float4 *A, *B, *C;
    int4 *D;
};
// Occlusion (NEE) rays: A = (origin.xyz, tmax)  B = (dir.xyz, srcPrim bits)
//   C = (value.rgb, pixel bits): the clamped radiance the light shader adds when the ray is unoccluded
struct ShadowQueue {
    float4 *A, *B, *C;
};
// closest-hit record: prim | frontCCW << 31 (prim == 0x7FFFFFFF: miss), t, u, v
static const uint32_t kMissPrim = 0x7FFFFFFFu;

enum MissKind { MISS_NONE = 0, MISS_ENV = 1, MISS_DIR = 2, MISS_POINT = 3, MISS_SPOT = 4 };

static const int kEnvRowGuide = 256, kEnvColGuide = 64; // buckets of the guide tables (powers of two: x * K is exact)

// ---- per-scene constant block (device copy) ---------------------------------------------------
struct SceneDev {
    const Node4 *nodes;
    const Node32 *nodes32; // the same tree for k_trace (hr_build.hip: encodeNodes32)
    const int *leafKeys;   // Node4::c.w of every node, compact: the way from a Node32 leaf child to its triangle
    const Tri *tris;
    const TriAttr *attrs;
    const TriAttrExt *attrsExt; // may be null
    const hr_material *materials;
    const TexDesc *textures;
    int32_t nTris, nNodes, rootLeafCount, nMaterials, nTextures;
    float rayEps;
    float gridLo[3], gridCell[3]; // frame grid of the 32-byte nodes: origin = gridLo + g * gridCell (power-of-two cells, 256 per axis cover the scene)
    uint32_t gridCellExp[3];      // biased exponent of gridCell
    float hitPad; // half the leaf padding: a hit point lies inside its triangle's bounding box grown by this much (hr_trace.h: hitInTriBox)
    hr_lights lights;
    // sample tables
    const float2 *seq, *aperture, *seqOffsets;
    int32_t nSeq, seqLen, nSeqOffsets;
    // importance table of the environment map (HR_ESTIMATOR_ENV_MIS): P(row < j), P(col < i | row j), P(texel); envW == 0: none
    const float *envRowCdf, *envColCdf, *envProb;
    const uint16_t *envRowGuide, *envColGuide; // guide tables of the two inverse-CDF searches (hr_build.hip::k_env_guides)
    int32_t envW, envH;
    float envMeanLum; // solid-angle-weighted mean luminosity of the map (light-pick weight of the MIS estimator)
    // interactive-mode block table (hr_interactive_blocks_set); blockNx == 0: the unshuffled list
    int32_t blockNx, blockNy;
    int32_t blockCoords[32];
    // HR_TEXTURE_LOD_CONE: per triangle (prim id) 0.5 * log2(uv area / world area), or null
    const float *texDensity;
};

// ---- counters ---------------------------------------------------------------------------------
static const int kMaxBounceSlots = 72;
struct Counters {
    uint32_t qCount[kMaxBounceSlots]; // rays in the closest-hit queue of iteration i
    uint32_t sCount[kMaxBounceSlots]; // rays in the occlusion queue produced by iteration i
    uint32_t pCount[kMaxBounceSlots]; // closest-hit rays of iteration i that hit a PBR material (entries of the pass's hit list, from the front)
    uint32_t gCount[kMaxBounceSlots]; // ... that hit a glass material (entries from the back of the same list)
};

} // namespace hr
