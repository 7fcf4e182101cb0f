// hr_trace.h — BVH traversal + Möller–Trumbore for one ray per lane (wave64).
//
// Replaces what OpenRL does inside rlRenderFrame() for every ray
// (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386).
// The hit is defined by the triangle test alone — (t, prim id) lexicographic minimum over
// t in (tmin, tmax) — so it does not depend on the acceleration structure; the slab test only has
// to be conservative (boxes are padded at build time).  While-while traversal: all lanes descend
// inner nodes until each holds a leaf, then all test triangles.  The deferred-child stack lives in
// LDS, one column per lane ([entry][lane] => bank = lane, conflict-free), with a private overflow
// region so that no tree depth can corrupt it.
#pragma once

#include "hr_texture.h"
#include "hr_types.h"

namespace hr {

static const int kStackLDS = 24;   // entries per lane kept in LDS
static const int kStackOvf = 40;   // private overflow (LBVH depth <= 30 + 28 index bits < 64)
static const int kSentinel = 0x7FFFFFFF;
static const int kRefillLanes = 24; // refill a wave from the work pool once this many lanes are idle
static const int kFetchChunk = 256; // work items a wave reserves per global atomic
static const int kTriPhaseLanes = 20; // run the triangle phase once this many lanes are blocked on a postponed leaf

struct HitRec {
    uint32_t prim; // kMissPrim when nothing was hit; bit 31: counter-clockwise front face seen by the ray
    float t, u, v;
};

// Reciprocal for the slab test only (never for the hit): no infinities / NaNs enter the box test.
HRD float safeInv(float d)
{
    const float lim = 1e-20f;
    if (abs_(d) < lim) d = (d < 0.0f) ? -lim : lim;
    return 1.0f / d;
}

// physicallyBased.rlsl:57-91 seen by an occlusion ray on a non-occluder (alpha-masked) primitive
HRD bool alphaPasses(const SceneDev &S, uint32_t prim, float u, float v)
{
    const TriAttr &a = S.attrs[prim];
    const uint32_t mid = a.matflags & kMatMask;
    if (mid >= (uint32_t)S.nMaterials) return false;
    const hr_material &m = S.materials[mid];
    if (m.type != HR_MAT_PBR || !(m.flags & HR_MF_ALPHA_MASK)) return false;
    float alpha = 1.0f;
    if ((m.flags & HR_MF_HAS_BASE_COLOR_TEXTURE) && m.base_color_texture >= 0 && m.base_color_texture < S.nTextures &&
        S.textures[m.base_color_texture].px) {
        float w = 1.0f - u - v;
        float tu = a.uv[0] * w + a.uv[2] * u + a.uv[4] * v;
        float tv = a.uv[1] * w + a.uv[3] * u + a.uv[5] * v;
        alpha = sampleTexture(S.textures[m.base_color_texture], tu, tv).w;
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
    const float oix = o.x * idx, oiy = o.y * idy, oiz = o.z * idz;
    float tlim = tmax; // shrinks to the closest hit so far (closest-hit rays only)

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
            cur = (sp < kStackLDS) ? stackLane[sp * 64] : ovf[sp - kStackLDS];     \
        }                                                                          \
    } while (0)

    while (cur != kSentinel) {
        // ---- inner nodes: descend until this lane holds a leaf (cur < 0) or runs out of work
        while (cur >= 0 && cur != kSentinel) {
            const Node &n = S.nodes[cur];
            const float4 na = n.a, nb = n.b, nc = n.c;
            const int4 nd = n.d;
            if (STATS) ++nodeVisits;
            // slab test of both children; t = plane * (1/d) - o * (1/d)
            float t0 = __builtin_fmaf(na.x, idx, -oix), t1 = __builtin_fmaf(na.w, idx, -oix);
            float t2 = __builtin_fmaf(na.y, idy, -oiy), t3 = __builtin_fmaf(nb.x, idy, -oiy);
            float t4 = __builtin_fmaf(na.z, idz, -oiz), t5 = __builtin_fmaf(nb.y, idz, -oiz);
            float tn0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0, t1), __builtin_fminf(t2, t3)),
                                        __builtin_fmaxf(__builtin_fminf(t4, t5), tmin));
            float tf0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0, t1), __builtin_fmaxf(t2, t3)),
                                        __builtin_fminf(__builtin_fmaxf(t4, t5), tlim));
            t0 = __builtin_fmaf(nb.z, idx, -oix), t1 = __builtin_fmaf(nc.y, idx, -oix);
            t2 = __builtin_fmaf(nb.w, idy, -oiy), t3 = __builtin_fmaf(nc.z, idy, -oiy);
            t4 = __builtin_fmaf(nc.x, idz, -oiz), t5 = __builtin_fmaf(nc.w, idz, -oiz);
            float tn1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0, t1), __builtin_fminf(t2, t3)),
                                        __builtin_fmaxf(__builtin_fminf(t4, t5), tmin));
            float tf1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0, t1), __builtin_fmaxf(t2, t3)),
                                        __builtin_fminf(__builtin_fmaxf(t4, t5), tlim));
            const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
            if (h0 && h1) {
                const bool firstIs0 = tn0 <= tn1;
                const int nearC = firstIs0 ? nd.x : nd.y, farC = firstIs0 ? nd.y : nd.x;
                HR_PUSH(farC);
                cur = nearC;
            } else if (h0) {
                cur = nd.x;
            } else if (h1) {
                cur = nd.y;
            } else {
                HR_POP();
            }
        }
        // ---- leaf: 1..4 triangles
        if (cur < 0) {
            const int enc = ~cur;
            const int first = enc & 0x0FFFFFFF, count = (enc >> 28) + 1;
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
#undef HR_PUSH
#undef HR_POP
}

} // namespace hr
