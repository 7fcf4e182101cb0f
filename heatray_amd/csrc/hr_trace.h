// hr_trace.h — BVH traversal + Möller–Trumbore for one ray per lane (wave64).
//
// Replaces what OpenRL does inside rlRenderFrame() for every ray
// (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386).
// The hit is defined by the triangle test alone — Möller–Trumbore AND the hit point inside the triangle's
// half-padded bounding box (hitInTriBox below), (t, prim id) lexicographic minimum over t in (tmin, tmax) —
// so it does not depend on the acceleration structure; the slab test only has to be conservative (boxes
// are padded at build time).  While-while traversal: all lanes descend
// inner nodes until each holds a leaf, then all test triangles.  The deferred-child stack lives in
// LDS, one column per lane ([entry][lane] => bank = lane, conflict-free), with a private overflow
// region sized for the deepest tree the builder can produce (kStackLDS + kStackOvf >= 3 x kMaxTreeLevels;
// hr_scene_commit additionally checks the built tree's level count against it).
#pragma once

#include "hr_texture.h"
#include "hr_types.h"

namespace hr {

#ifndef HR_STACK_LDS
#define HR_STACK_LDS 16
#endif
static const int kStackLDS = HR_STACK_LDS; // entries per lane kept in LDS
// A 4-wide node pushes up to 3 entries and the ray descends one level, so a path through L inner levels holds at most 3 L
// entries.  The collapse opens the child with the largest area, so an unopened sibling sits only ONE binary level deeper:
// along such a path the 4-wide depth equals the binary depth, and that is bounded by the key length of the radix tree:
// 30 Morton bits + 28 index bits (n < 2^28) = 58 levels.
#ifndef HR_TREE_LEVELS
#define HR_TREE_LEVELS 58
#endif
static const int kMaxTreeLevels = HR_TREE_LEVELS;
static const int kStackOvf = 3 * kMaxTreeLevels + 2 - HR_STACK_LDS; // private overflow area (scratch; touched by 0.1 % of node steps on c3)
static const int kSentinel = 0x7FFFFFFF;
static const int kRefillLanes = 24; // refill a wave from the work pool once this many lanes are idle
static const int kTriPhaseLanes = 20; // run the triangle phase once this many lanes are blocked on a postponed leaf

struct HitRec {
    uint32_t prim; // kMissPrim when nothing was hit; bit 31: counter-clockwise front face seen by the ray
    float t, u, v;
};

// Per-ray constants of the slab test: 1/d and o/d.
struct RayK {
    float idx, idy, idz, oix, oiy, oiz;
};

#define HR_PUSH(v)                          \
    do {                                    \
        if (sp < kStackLDS)                 \
            stackLane[sp * 64] = (v);       \
        else                                \
            ovf[sp - kStackLDS] = (v);      \
        ++sp;                               \
    } while (0)
#define HR_POP()                                                                   \
    do {                                                                           \
        if (sp == 0)                                                               \
            cur = kSentinel;                                                       \
        else {                                                                     \
            --sp;                                                                  \
            /* two typed reads, not one read through a selected pointer: that one would be a FLAT load (both waitcnt counters, TA) */ \
            cur = stackLane[(sp < kStackLDS ? sp : kStackLDS - 1) * 64];           \
            if (sp >= kStackLDS) cur = ovf[sp - kStackLDS];                        \
        }                                                                          \
    } while (0)

HRD uint32_t byteOf(uint32_t x, int c) { return (x >> (8 * c)) & 0xFFu; } // -> v_cvt_f32_ubyteN when converted to float
HRD void cswap(uint32_t &a, uint32_t &b)
{
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo, b = hi;
}

// One step at the 4-wide node `cur`: slab-test the four quantised child boxes, continue with the nearest child that is hit and push the
// others farthest first (so the nearer one pops first); pop when nothing is hit.  On the 32-byte node (hr_types.h: Node32): TWO loads per
// visit.  The frame's origin is a point of the scene's 256^3 grid, gridLo + g * cell, and the scale cell * 2^(e - 8) with an exponent per
// axis; the grid is folded into the ray's constants once per ray (rayFrame32: cell / d and (o - gridLo) / d), so that a visit decodes
//   t = q * (2^(e - 8) * cell/d) + (g * cell/d - (o - gridLo)/d)
// with 24 v_cvt_f32_ubyteN + 24 FMAs for the planes (as on the 64-byte node the packet kernel reads), three more conversions for the grid
// bytes and a few integer instructions for the packed exponents.  A leaf child's reference is ~(4 * node + 3 - slot): the triangle's index
// is ~(leafKeys[node] + slot), read when the leaf is tested.
HRD RayK rayFrame32(const SceneDev &S, v3 o, float idx, float idy, float idz)
{
    return RayK{S.gridCell[0] * idx, S.gridCell[1] * idy, S.gridCell[2] * idz, (o.x - S.gridLo[0]) * idx, (o.y - S.gridLo[1]) * idy, (o.z - S.gridLo[2]) * idz};
}
HRD void nodeStep32(const Node32 *__restrict__ nodes, int &cur, int &sp, int *stackLane, int *ovf, const RayK &rk, float tmin, float tlim)
{
    // (a 32-bit byte offset from the wave-uniform base: one shift, and the load takes base + offset itself; node indices are < 2^25)
    const Node32 &n = *reinterpret_cast<const Node32 *>(reinterpret_cast<const char *>(nodes) + (size_t)((uint32_t)cur << 5));
    const uint4 qp = n.p, qq = n.q;
    const uint32_t w6 = qq.z, w7 = qq.w;
    const uint32_t nInner = (w6 >> 25) & 7u;
    const int innerBase = (int)(w6 & 0x01FFFFFFu), leafKey = ~(int)(((uint32_t)cur << 2) | 3u);
    // 2^(e - 8) per axis: the biased exponent e + 119 goes straight into the float's exponent field
    const float bx = rk.idx * __uint_as_float(((w6 >> 28) << 23) + (119u << 23));
    const float by = rk.idy * __uint_as_float((((w7 >> 24) & 15u) << 23) + (119u << 23));
    const float bz = rk.idz * __uint_as_float(((w7 >> 28) << 23) + (119u << 23));
    const float ax = __builtin_fmaf((float)byteOf(w7, 0), rk.idx, -rk.oix), ay = __builtin_fmaf((float)byteOf(w7, 1), rk.idy, -rk.oiy),
                az = __builtin_fmaf((float)byteOf(w7, 2), rk.idz, -rk.oiz);
    // the sign of the direction decides which quantised plane is the entry and which the exit plane of each slab
    // (chosen once per node on whole dwords: byte j belongs to child j)
    const uint32_t nX = rk.idx < 0.0f ? qp.w : qp.x, fX = rk.idx < 0.0f ? qp.x : qp.w;
    const uint32_t nY = rk.idy < 0.0f ? qq.x : qp.y, fY = rk.idy < 0.0f ? qp.y : qq.x;
    const uint32_t nZ = rk.idz < 0.0f ? qq.y : qp.z, fZ = rk.idz < 0.0f ? qp.z : qq.y;
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tnx = __builtin_fmaf((float)byteOf(nX, c), bx, ax), tfx = __builtin_fmaf((float)byteOf(fX, c), bx, ax);
        const float tny = __builtin_fmaf((float)byteOf(nY, c), by, ay), tfy = __builtin_fmaf((float)byteOf(fY, c), by, ay);
        const float tnz = __builtin_fmaf((float)byteOf(nZ, c), bz, az), tfz = __builtin_fmaf((float)byteOf(fZ, c), bz, az);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        // entry distance (>= tmin >= 0, so its bits order like the value) with the child slot in the two low bits
        key[c] = (tn <= tf) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu; // (a slot without a child has inverted planes)
    }
    // sorting network: ascending entry distance, misses (0xFFFFFFFF) last
    cswap(key[0], key[1]), cswap(key[2], key[3]), cswap(key[0], key[2]), cswap(key[1], key[3]), cswap(key[1], key[2]);
    int ref[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sl = (int)(key[j] & 3u);
        ref[j] = (((uint32_t)sl < nInner) ? innerBase : leafKey) + sl; // (a miss computes a value nobody uses)
    }
    if (sp <= kStackLDS - 3) {
        // common case, branch-free: store the three farther children farthest first and advance the stack pointer only
        // past the ones that were hit (hits are a prefix of the sorted order, so a skipped slot is simply overwritten)
        stackLane[sp * 64] = ref[3];
        sp += (key[3] != 0xFFFFFFFFu) ? 1 : 0;
        stackLane[sp * 64] = ref[2];
        sp += (key[2] != 0xFFFFFFFFu) ? 1 : 0;
        stackLane[sp * 64] = ref[1];
        sp += (key[1] != 0xFFFFFFFFu) ? 1 : 0;
        const bool any = key[0] != 0xFFFFFFFFu;
        const bool empty = !any && sp == 0;
        sp -= (!any && sp > 0) ? 1 : 0;
        const int popped = stackLane[(sp < kStackLDS - 1 ? sp : kStackLDS - 1) * 64]; // stays inside the LDS part (value unused when sp == kStackLDS)
        cur = any ? ref[0] : (empty ? kSentinel : popped);
    } else {
        // deep stack: entries beyond kStackLDS live in the private overflow area
#pragma unroll
        for (int j = 3; j >= 1; --j)
            if (key[j] != 0xFFFFFFFFu) HR_PUSH(ref[j]);
        if (key[0] != 0xFFFFFFFFu) {
            cur = ref[0];
        } else {
            HR_POP();
        }
    }
}

// THE HIT TEST'S SECOND HALF: a Möller–Trumbore candidate (u, v, t in range) is a hit only if its hit point
// o + t d lies inside the triangle's own bounding box grown by `h` = half the leaf padding.  float32 Möller–Trumbore alone accepts,
// about once in 10^9 rays, a ray that passes a SLIVER triangle at a distance (the determinant is a difference of nearly equal
// products), and whether a traversal ever tests that triangle depends on the boxes of its tree.  With this rule it does not: every
// conservative tree's leaf box contains the triangle's box grown by the WHOLE padding, so a hit point inside the half-grown box is
// inside every ancestor's box with half a padding (5e-6 |diagonal|, ~50 ulp of a coordinate) to spare for the rounding of the slab
// tests — every traversal of every tree tests the triangle and finds the same answer; and a candidate whose point lies outside is a
// hit for nobody, brute force included.  Same operations, same order as oracle_bvh.cpp::hitInTriBox.
HRD bool hitInTriBoxAxis(float v0, float e1, float e2, float o, float d, float t, float h) // one coordinate of it
{
    const float p1 = v0 + e1, p2 = v0 + e2;
    const float lo = fmin_(fmin_(v0, p1), p2), hi = fmax_(fmax_(v0, p1), p2);
    const float P = o + t * d;
    return P >= lo - h && P <= hi + h;
}
HRD bool hitInTriBox(v3 v0, v3 e1, v3 e2, v3 o, v3 d, float t, float h)
{
    return hitInTriBoxAxis(v0.x, e1.x, e2.x, o.x, d.x, t, h) && hitInTriBoxAxis(v0.y, e1.y, e2.y, o.y, d.y, t, h) &&
           hitInTriBoxAxis(v0.z, e1.z, e2.z, o.z, d.z, t, h);
}

// Per-ray constants of the slab test: 1 / d and the ray origin over d
HRD RayK rayFrame(v3 o, float idx, float idy, float idz) { return RayK{idx, idy, idz, o.x * idx, o.y * idy, o.z * idz}; }

// Does the ray miss the frame box of the root node — [origin, origin + 255 * scale] per axis, which contains every child's quantised box?
// Same arithmetic as nodeStep4 with the plane bytes 0 and 255: t = q * (scale / d) + (origin / d - o / d).  For any child plane
// 0 <= q <= 255 the entry distance is >= this box's and the exit distance <= this box's (fma is monotone in q for a fixed slope), so
// `true` here implies that nodeStep4 at the root finds no child: the ray's closest hit is a miss.  (NaNs compare false: not culled.)
HRD bool rootMissed(const Node4 *__restrict__ nodes, v3 o, v3 d, float tmin, float tlim);

// Reciprocal for the slab test only (never for the hit): no infinities / NaNs enter the box test.
HRD float safeInv(float d)
{
    const float lim = 1e-20f;
    if (abs_(d) < lim) d = (d < 0.0f) ? -lim : lim;
    return 1.0f / d;
}

HRD bool rootMissed(const Node4 *__restrict__ nodes, v3 o, v3 d, float tmin, float tlim)
{
    const float4 a = nodes[0].a;
    const uint32_t meta = __float_as_uint(a.w);
    const float idx = safeInv(d.x), idy = safeInv(d.y), idz = safeInv(d.z);
    const RayK rk = rayFrame(o, idx, idy, idz);
    const float bx = __uint_as_float((meta & 0xFFu) << 23) * rk.idx;
    const float by = __uint_as_float(((meta >> 8) & 0xFFu) << 23) * rk.idy;
    const float bz = __uint_as_float(((meta >> 16) & 0xFFu) << 23) * rk.idz;
    const float ax = __builtin_fmaf(a.x, rk.idx, -rk.oix), ay = __builtin_fmaf(a.y, rk.idy, -rk.oiy), az = __builtin_fmaf(a.z, rk.idz, -rk.oiz);
    const float nX = rk.idx < 0.0f ? 255.0f : 0.0f, fX = rk.idx < 0.0f ? 0.0f : 255.0f;
    const float nY = rk.idy < 0.0f ? 255.0f : 0.0f, fY = rk.idy < 0.0f ? 0.0f : 255.0f;
    const float nZ = rk.idz < 0.0f ? 255.0f : 0.0f, fZ = rk.idz < 0.0f ? 0.0f : 255.0f;
    const float tnx = __builtin_fmaf(nX, bx, ax), tfx = __builtin_fmaf(fX, bx, ax);
    const float tny = __builtin_fmaf(nY, by, ay), tfy = __builtin_fmaf(fY, by, ay);
    const float tnz = __builtin_fmaf(nZ, bz, az), tfz = __builtin_fmaf(fZ, bz, az);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
    const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
    return tn > tf;
}

// physicallyBased.rlsl:57-91 seen by an occlusion ray on a non-occluder (alpha-masked) primitive
HRD bool alphaPasses(const SceneDev &S, uint32_t prim, float u, float v)
{
    const auto &a = G(S.attrs)[prim];
    const uint32_t mid = a.matflags & kMatMask;
    if (mid >= (uint32_t)S.nMaterials) return false;
    const auto &m = G(S.materials)[mid];
    if (m.type != HR_MAT_PBR || !(m.flags & HR_MF_ALPHA_MASK)) return false;
    float alpha = 1.0f;
    if ((m.flags & HR_MF_HAS_BASE_COLOR_TEXTURE) && m.base_color_texture >= 0 && m.base_color_texture < S.nTextures &&
        G(S.textures)[m.base_color_texture].px) {
        float w = 1.0f - u - v;
        float tu = a.uv[0] * w + a.uv[2] * u + a.uv[4] * v;
        float tv = a.uv[1] * w + a.uv[3] * u + a.uv[5] * v;
        alpha = sampleTexture(G(S.textures)[m.base_color_texture], tu, tv).w;
    }
    return alpha < 1.0f;
}

template <bool ANY, bool STATS>
HRD void traverse(const SceneDev &S, v3 o, v3 d, float tmin, float tmax, uint32_t skipPrim, int *stackLane /* LDS, stride 64 */,
                  HitRec &best, uint32_t &nodeVisits, uint32_t &triTests)
{
    best.prim = kMissPrim;
    best.t = tmax;
    best.u = best.v = 0.0f;
    if (S.nTris == 0) return;
    int ovf[kStackOvf];
    int sp = 0;
    int cur;
    if (S.rootLeafCount > 0)
        cur = ~(0 | ((S.rootLeafCount - 1) << 28));
    else
        cur = 0;
    const float idx = safeInv(d.x), idy = safeInv(d.y), idz = safeInv(d.z);
    float tlim = tmax; // shrinks to the closest hit so far (closest-hit rays only)
    const RayK rk = rayFrame32(S, o, idx, idy, idz);
    const bool rootLeaf = S.rootLeafCount > 0;
    while (cur != kSentinel) {
        // ---- inner nodes: descend until this lane holds a leaf (cur < 0) or runs out of work
        while (cur >= 0 && cur != kSentinel) {
            if (STATS) ++nodeVisits;
            nodeStep32(S.nodes32, cur, sp, stackLane, ovf, rk, tmin, tlim);
        }
        // ---- leaf: one triangle (a root leaf: 1..4), found through the node's leaf key (k_trace does the same)
        if (cur < 0) {
            const int enc = ~cur;
            int first = enc & 0x0FFFFFFF, count = (enc >> 28) + 1;
            if (!rootLeaf) first = ~(S.leafKeys[enc >> 2] + (3 - (enc & 3))), count = 1;
            bool done = false;
            for (int k = 0; k < count; ++k) {
                const Tri &tr = S.tris[first + k];
                const float4 tp = tr.p, tq = tr.q, trr = tr.r;
                if (STATS) ++triTests;
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
                if (!hitInTriBox(v0, e1, e2, o, d, t, S.hitPad)) continue;
                if (ANY) {
                    if ((__float_as_uint(trr.z) & TF_NON_OCCLUDER) && alphaPasses(S, prim, u, v)) continue;
                    best.prim = prim;
                    best.t = t;
                    done = true;
                    break;
                }
                const uint32_t bp = best.prim & 0x7FFFFFFFu;
                if (best.prim == kMissPrim || t < best.t || (t == best.t && prim < bp)) {
                    best.prim = prim | ((det > 0.0f) ? 0x80000000u : 0u);
                    best.t = t;
                    best.u = u;
                    best.v = v;
                    tlim = t;
                }
            }
            if (ANY && done) return;
            HR_POP();
        }
    }
}

} // namespace hr
