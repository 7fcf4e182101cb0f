// hr_types.h — device-resident data layouts of libhrcore (DESIGN.md §Data layout in HBM).
#pragma once

#include "../../include/hrcore.h"
#include "hr_math.h"

namespace hr {

// ---- acceleration structure -------------------------------------------------------------------
#ifndef HR_NODE32
#define HR_NODE32 0 // the 64-byte node below is the default; 1 selects the 32-byte experiment (DESIGN.md §2 "32-byte nodes")
#endif
#if HR_NODE32
// EXPERIMENT (kept buildable, parity-green, not the default): a 4-wide BVH node in 32 bytes = TWO dwordx4 loads instead of three.
// A CU's texture addresser accepts per-lane loads at about one lane per cycle per instruction whatever the width
// (tools/calib_tcp.hip), and k_trace keeps it 87 % busy, so fewer load instructions per node do unload it (-21 % TA busy cycles)
// — but unpacking 7-bit planes and the grid frame costs 23 % more VALU instructions, which were at 73 % busy already: the kernel
// becomes VALU-bound and ends 1.5 % slower (profiles/r2f_node32_vs_node64.txt).
//   p = (lo.x, lo.y, lo.z, hi.x)  q = (hi.y, hi.z, w6, w7): six plane dwords; bits 7c..7c+6 of a plane dword = child c's plane,
//   quantised to 7 bits (lo rounded down, hi rounded up: conservative); a child slot without a child holds lo = 127, hi = 0 and can
//   never be hit.  The node's frame is origin + q * scale with
//     origin_k = gridLo_k + g_k * cell_k   g_k: 14-bit coordinate on a scene-wide power-of-two grid (SceneDev::grid*)
//     scale_k  = cell_k * 2^(r_k - 7)      r_k: 4 bits
//   w6 = g.x | g.y << 14 | r.x << 28;  w7 = childBase | r.y << 28;  the top nibbles of the plane dwords hold g.z (lo.x, lo.y, lo.z:
//   4 bits each, hi.x: 2 bits, above them the number of children - 1), r.z (hi.y) and the number of inner children (hi.z).
// Children 0 .. nInner-1 are the nodes childBase + j (allocated together); the others are single triangles stored at
// tris[4 * node + 3 - j]: the reference of child j is `base + j` for both kinds (base = childBase or ~(4 * node + 3)).
// HR_NODE32 == 2: the same 32-byte record holding a THREE-wide node with 8-bit planes (byte c of a plane dword = child c: the
// cheap v_cvt_f32_ubyteN decode of the 64-byte node again); the top bytes of the plane dwords carry g.z (lo.x: low 8 bits,
// lo.y: high 6 bits, above them the number of children - 1) and r.z | nInner << 4 (lo.z); w6, w7 as above.  A node's triangles live
// at tris[3 * node + 2 - j].  Two loads per visit like the 4-wide 32-byte node, and fewer VALU instructions per visit than the
// 64-byte node (18 conversions, a 3-key sorting network) — against ~1.25 x the visits of a 4-wide tree.
struct alignas(32) Node4 {
    uint4 p, q;
};
static const int kGridBits = 14;
static const int kNodeWidth = (HR_NODE32 == 2) ? 3 : 4; // children per node, triangle slots per node
static const int kPlaneMax = (HR_NODE32 == 2) ? 255 : 127;
#else
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

#endif

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
//   M = merge word of a ray whose subtrees have been handed to other waves (k_trace's steal pool): groups of lanes still
//       traversing parts of it | occluded << 31.  Written only for such rays; never initialised for the others.
struct ShadowQueue {
    float4 *A, *B, *C;
    uint32_t *M;
};
// closest-hit record: prim | frontCCW << 31 (prim == 0x7FFFFFFF: miss), t, u, v
static const uint32_t kMissPrim = 0x7FFFFFFFu;

enum MissKind { MISS_NONE = 0, MISS_ENV = 1, MISS_DIR = 2, MISS_POINT = 3, MISS_SPOT = 4 };

static const int kEnvRowGuide = 256, kEnvColGuide = 64; // buckets of the guide tables (powers of two: x * K is exact)

// ---- per-scene constant block (device copy) ---------------------------------------------------
struct SceneDev {
    const Node4 *nodes;
    const Tri *tris;
    const TriAttr *attrs;
    const TriAttrExt *attrsExt; // may be null
    const hr_material *materials;
    const TexDesc *textures;
    int32_t nTris, nNodes, rootLeafCount, nMaterials, nTextures;
    float rayEps;
    hr_lights lights;
    // sample tables
    const float2 *seq, *aperture, *seqOffsets;
    int32_t nSeq, seqLen, nSeqOffsets;
    // frame grid of the 32-byte nodes: origin of the grid, cell size (a power of two per axis), biased exponent of cell * 2^-7
    float gridLo[3], gridCell[3];
    int32_t gridExpM7[3];
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
    // prim id -> position in `tris` (k_shade_sort recomputes the barycentrics of a hit whose ray was traversed by several waves)
    const uint32_t *slotOfPrim;
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
