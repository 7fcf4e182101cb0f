// hr_trace.h — BVH traversal + Möller–Trumbore for one ray per lane (wave64).
//
// Replaces what OpenRL does inside rlRenderFrame() for every ray
// (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386).
// The hit is defined by the triangle test alone — (t, prim id) lexicographic minimum over
// t in (tmin, tmax) — so it does not depend on the acceleration structure; the slab test only has
// to be conservative (boxes are padded at build time).  While-while traversal: all lanes descend
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
#ifndef HR_LDS_NODES
#define HR_LDS_NODES 0 // nodes of the top of the tree a k_trace workgroup copies to LDS (experiment)
#endif
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

#if HR_NODE32
// Per-scene constants of the 32-byte node frames (wave-uniform: scalar registers)
struct GridK {
    float cellX, cellY, cellZ;
    int32_t expX, expY, expZ; // biased exponent of cell * 2^-7
};
HRD GridK gridOf(const SceneDev &S) { return GridK{S.gridCell[0], S.gridCell[1], S.gridCell[2], S.gridExpM7[0], S.gridExpM7[1], S.gridExpM7[2]}; }
// RayK for 32-byte nodes: oix/oiy/oiz hold (o - gridLo) / d (rayFrame below)

HRD uint32_t q7(uint32_t x, int c) { return (x >> (7 * c)) & 127u; } // -> v_bfe_u32

#if HR_NODE32 == 2
// One step at the 3-wide node `cur` (32-byte format with 8-bit planes, hr_types.h): two dwordx4 loads, slab test of three child
// boxes, continue with the nearest child that is hit and push the others farthest first; pop when nothing is hit.
HRD void nodeStep4(const Node4 *__restrict__ nodes, int &cur, int &sp, int *stackLane, int *ovf, const RayK &rk, const GridK &gk, float tmin, float tlim)
{
    const Node4 &n = nodes[cur];
    const uint4 P = n.p, Q = n.q;
    const uint32_t nInner = (P.z >> 28) & 3u;
    const int innerBase = (int)(Q.w & 0x0FFFFFFFu), leafKey = ~(3 * cur + 2);
    const uint32_t gx = Q.z & 0x3FFFu, gy = (Q.z >> 14) & 0x3FFFu;
    const uint32_t gz = (P.x >> 24) | (((P.y >> 24) & 0x3Fu) << 8);
    const float bx = __uint_as_float((uint32_t)(gk.expX + (int32_t)(Q.z >> 28)) << 23) * rk.idx;
    const float by = __uint_as_float((uint32_t)(gk.expY + (int32_t)(Q.w >> 28)) << 23) * rk.idy;
    const float bz = __uint_as_float((uint32_t)(gk.expZ + (int32_t)((P.z >> 24) & 15u)) << 23) * rk.idz;
    const float ax = __builtin_fmaf((float)gx * gk.cellX, rk.idx, -rk.oix), ay = __builtin_fmaf((float)gy * gk.cellY, rk.idy, -rk.oiy),
                az = __builtin_fmaf((float)gz * gk.cellZ, rk.idz, -rk.oiz);
    const uint32_t nX = rk.idx < 0.0f ? P.w : P.x, fX = rk.idx < 0.0f ? P.x : P.w;
    const uint32_t nY = rk.idy < 0.0f ? Q.x : P.y, fY = rk.idy < 0.0f ? P.y : Q.x;
    const uint32_t nZ = rk.idz < 0.0f ? Q.y : P.z, fZ = rk.idz < 0.0f ? P.z : Q.y;
    uint32_t key[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float tnx = __builtin_fmaf((float)byteOf(nX, c), bx, ax), tfx = __builtin_fmaf((float)byteOf(fX, c), bx, ax);
        const float tny = __builtin_fmaf((float)byteOf(nY, c), by, ay), tfy = __builtin_fmaf((float)byteOf(fY, c), by, ay);
        const float tnz = __builtin_fmaf((float)byteOf(nZ, c), bz, az), tfz = __builtin_fmaf((float)byteOf(fZ, c), bz, az);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        key[c] = (tn <= tf) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu; // (a slot without a child: inverted box)
    }
    cswap(key[0], key[1]), cswap(key[1], key[2]), cswap(key[0], key[1]);
    int ref[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int sl = (int)(key[j] & 3u);
        ref[j] = (((uint32_t)sl < nInner) ? innerBase : leafKey) + sl;
    }
    if (sp <= kStackLDS - 2) {
        stackLane[sp * 64] = ref[2];
        sp += (key[2] != 0xFFFFFFFFu) ? 1 : 0;
        stackLane[sp * 64] = ref[1];
        sp += (key[1] != 0xFFFFFFFFu) ? 1 : 0;
        const bool any = key[0] != 0xFFFFFFFFu;
        const bool empty = !any && sp == 0;
        sp -= (!any && sp > 0) ? 1 : 0;
        const int popped = stackLane[(sp < kStackLDS - 1 ? sp : kStackLDS - 1) * 64];
        cur = any ? ref[0] : (empty ? kSentinel : popped);
    } else {
#pragma unroll
        for (int j = 2; j >= 1; --j)
            if (key[j] != 0xFFFFFFFFu) HR_PUSH(ref[j]);
        if (key[0] != 0xFFFFFFFFu) {
            cur = ref[0];
        } else {
            HR_POP();
        }
    }
}
#else
// One step at the 4-wide node `cur` (32-byte format, hr_types.h): TWO dwordx4 loads, slab test of the four 7-bit child boxes,
// continue with the nearest child that is hit and push the others farthest first; pop when nothing is hit.
HRD void nodeStep4(const Node4 *__restrict__ nodes, int &cur, int &sp, int *stackLane, int *ovf, const RayK &rk, const GridK &gk, float tmin, float tlim)
{
    const Node4 &n = nodes[cur];
    const uint4 P = n.p, Q = n.q;
    const uint32_t nInner = (Q.y >> 28) & 7u;
    const int innerBase = (int)(Q.w & 0x0FFFFFFFu), leafKey = ~(4 * cur + 3);
    const uint32_t gx = Q.z & 0x3FFFu, gy = (Q.z >> 14) & 0x3FFFu;
    const uint32_t gz = (P.x >> 28) | ((P.y >> 28) << 4) | ((P.z >> 28) << 8) | (((P.w >> 28) & 3u) << 12);
    // t = (gridLo + g * cell + q * scale - o) / d = q * (scale / d) + (g * cell) / d - (o - gridLo) / d
    const float bx = __uint_as_float((uint32_t)(gk.expX + (int32_t)(Q.z >> 28)) << 23) * rk.idx;
    const float by = __uint_as_float((uint32_t)(gk.expY + (int32_t)(Q.w >> 28)) << 23) * rk.idy;
    const float bz = __uint_as_float((uint32_t)(gk.expZ + (int32_t)(Q.x >> 28)) << 23) * rk.idz;
    const float ax = __builtin_fmaf((float)gx * gk.cellX, rk.idx, -rk.oix), ay = __builtin_fmaf((float)gy * gk.cellY, rk.idy, -rk.oiy),
                az = __builtin_fmaf((float)gz * gk.cellZ, rk.idz, -rk.oiz);
    // the sign of the direction decides which plane dword is the entry and which the exit plane of each slab
    const uint32_t nX = rk.idx < 0.0f ? P.w : P.x, fX = rk.idx < 0.0f ? P.x : P.w;
    const uint32_t nY = rk.idy < 0.0f ? Q.x : P.y, fY = rk.idy < 0.0f ? P.y : Q.x;
    const uint32_t nZ = rk.idz < 0.0f ? Q.y : P.z, fZ = rk.idz < 0.0f ? P.z : Q.y;
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tnx = __builtin_fmaf((float)q7(nX, c), bx, ax), tfx = __builtin_fmaf((float)q7(fX, c), bx, ax);
        const float tny = __builtin_fmaf((float)q7(nY, c), by, ay), tfy = __builtin_fmaf((float)q7(fY, c), by, ay);
        const float tnz = __builtin_fmaf((float)q7(nZ, c), bz, az), tfz = __builtin_fmaf((float)q7(fZ, c), bz, az);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        // entry distance (>= tmin >= 0, so its bits order like the value) with the child slot in the two low bits; a slot without
        // a child holds an inverted box (entry plane beyond the exit plane on every axis): tn > tf
        key[c] = (tn <= tf) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
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
#endif // HR_NODE32 == 2

#else
// One step at the 4-wide node `cur`: slab-test the four quantised child boxes, continue with the nearest child that
// is hit and push the others farthest first (so the nearer one pops first); pop when nothing is hit.
HRD void nodeStep4(const Node4 *__restrict__ nodes, int &cur, int &sp, int *stackLane, int *ovf, const RayK &rk, float tmin, float tlim,
                   bool anyHit = false, const uint4 *topLds = nullptr, int topCount = 0)
{
    float4 a;
    uint4 qb, qc;
#if HR_LDS_NODES
    if (cur < topCount) {
        // the first nodes of the array are the top of the tree (level order): the workgroup holds a copy of their 48 used bytes in LDS,
        // read without the texture addresser
        const uint4 t0 = topLds[cur * 3], t1 = topLds[cur * 3 + 1], t2 = topLds[cur * 3 + 2];
        a = make_float4(__uint_as_float(t0.x), __uint_as_float(t0.y), __uint_as_float(t0.z), __uint_as_float(t0.w));
        qb = t1, qc = t2;
    } else
#endif
    {
        // (a 32-bit byte offset from the wave-uniform base: one shift, and the load takes base + offset itself; node indices are < 2^26)
        const Node4 &n = *reinterpret_cast<const Node4 *>(reinterpret_cast<const char *>(nodes) + (size_t)((uint32_t)cur << 6));
        a = n.a;
        qb = n.b, qc = n.c;
    }
    const uint32_t meta = __float_as_uint(a.w);
    const uint32_t nInner = (meta >> 24) & 7u, nValid = meta >> 27;
    const int innerBase = (int)qc.z, leafKey = (int)qc.w;
    // t = (origin + q * scale - o) / d = q * (scale / d) + (origin / d - o / d)
    const float bx = __uint_as_float((meta & 0xFFu) << 23) * rk.idx;
    const float by = __uint_as_float(((meta >> 8) & 0xFFu) << 23) * rk.idy;
    const float bz = __uint_as_float(((meta >> 16) & 0xFFu) << 23) * rk.idz;
    const float ax = __builtin_fmaf(a.x, rk.idx, -rk.oix), ay = __builtin_fmaf(a.y, rk.idy, -rk.oiy), az = __builtin_fmaf(a.z, rk.idz, -rk.oiz);
    // the sign of the direction decides which quantised plane is the entry and which the exit plane of each slab
    // (chosen once per node on whole dwords: byte j belongs to child j)
    const uint32_t nX = rk.idx < 0.0f ? qb.w : qb.x, fX = rk.idx < 0.0f ? qb.x : qb.w;
    const uint32_t nY = rk.idy < 0.0f ? qc.x : qb.y, fY = rk.idy < 0.0f ? qb.y : qc.x;
    const uint32_t nZ = rk.idz < 0.0f ? qc.y : qb.z, fZ = rk.idz < 0.0f ? qb.z : qc.y;
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tnx = __builtin_fmaf((float)byteOf(nX, c), bx, ax), tfx = __builtin_fmaf((float)byteOf(fX, c), bx, ax);
        const float tny = __builtin_fmaf((float)byteOf(nY, c), by, ay), tfy = __builtin_fmaf((float)byteOf(fY, c), by, ay);
        const float tnz = __builtin_fmaf((float)byteOf(nZ, c), bz, az), tfz = __builtin_fmaf((float)byteOf(fZ, c), bz, az);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        // entry distance (>= tmin >= 0, so its bits order like the value) with the child slot in the two low bits
        key[c] = (tn <= tf && (uint32_t)c < nValid) ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
    }
    // sorting network: ascending entry distance, misses (0xFFFFFFFF) last
    cswap(key[0], key[1]), cswap(key[2], key[3]), cswap(key[0], key[2]), cswap(key[1], key[3]), cswap(key[1], key[2]);
    (void)anyHit;
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

#endif

#ifndef HR_ANY_UNORDERED
#define HR_ANY_UNORDERED 0 // EXPERIMENT, measured and NOT the default: see below
#endif
#if !HR_NODE32
// One step at the 4-wide node `cur` for an OCCLUSION ray: whether something occludes it does not depend on the order the children
// are visited in, so the four entry distances need not be packed into keys and sorted; the hit child with the lowest slot is
// continued with, the others are pushed in slot order.  Parity-green (the whole GPU suite) and SLOWER: k_trace 9.85 against 9.42 ms
// per launch (c3, 128 steps: 1707 vs 1769 Mrays/s) with 227.9 instead of 225.8 VALU wave-instructions per ray
// (profiles/r3k_anyhit_unordered_ab.txt): what the sorting network costs is less than the second copy of the six-times unrolled
// step loop, the vote that selects it and the boolean bookkeeping cost in registers and instructions; front-to-back order also lets
// an occluded ray find its occluder 0.6 % of a visit earlier on average.  Kept buildable (-DHR_ANY_UNORDERED=1) for the record.
HRD void nodeStep4Any(const Node4 *__restrict__ nodes, int &cur, int &sp, int *stackLane, int *ovf, const RayK &rk, float tmin, float tlim)
{
    const Node4 &n = nodes[cur];
    const float4 a = n.a;
    const uint4 qb = n.b, qc = n.c;
    const uint32_t meta = __float_as_uint(a.w);
    const uint32_t nInner = (meta >> 24) & 7u, nValid = meta >> 27;
    const int innerBase = (int)qc.z, leafKey = (int)qc.w;
    const float bx = __uint_as_float((meta & 0xFFu) << 23) * rk.idx;
    const float by = __uint_as_float(((meta >> 8) & 0xFFu) << 23) * rk.idy;
    const float bz = __uint_as_float(((meta >> 16) & 0xFFu) << 23) * rk.idz;
    const float ax = __builtin_fmaf(a.x, rk.idx, -rk.oix), ay = __builtin_fmaf(a.y, rk.idy, -rk.oiy), az = __builtin_fmaf(a.z, rk.idz, -rk.oiz);
    const uint32_t nX = rk.idx < 0.0f ? qb.w : qb.x, fX = rk.idx < 0.0f ? qb.x : qb.w;
    const uint32_t nY = rk.idy < 0.0f ? qc.x : qb.y, fY = rk.idy < 0.0f ? qb.y : qc.x;
    const uint32_t nZ = rk.idz < 0.0f ? qc.y : qb.z, fZ = rk.idz < 0.0f ? qb.z : qc.y;
    bool hit[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tnx = __builtin_fmaf((float)byteOf(nX, c), bx, ax), tfx = __builtin_fmaf((float)byteOf(fX, c), bx, ax);
        const float tny = __builtin_fmaf((float)byteOf(nY, c), by, ay), tfy = __builtin_fmaf((float)byteOf(fY, c), by, ay);
        const float tnz = __builtin_fmaf((float)byteOf(nZ, c), bz, az), tfz = __builtin_fmaf((float)byteOf(fZ, c), bz, az);
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, tmin));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        hit[c] = tn <= tf && (uint32_t)c < nValid;
    }
    const int r0 = (0u < nInner ? innerBase : leafKey), r1 = (1u < nInner ? innerBase : leafKey) + 1, r2 = (2u < nInner ? innerBase : leafKey) + 2,
              r3 = (3u < nInner ? innerBase : leafKey) + 3;
    // the lowest hit slot continues; a higher hit slot is pushed when a lower one was hit too
    const bool push3 = hit[3] && (hit[0] || hit[1] || hit[2]), push2 = hit[2] && (hit[0] || hit[1]), push1 = hit[1] && hit[0];
    const bool any = hit[0] || hit[1] || hit[2] || hit[3];
    const int next = hit[0] ? r0 : (hit[1] ? r1 : (hit[2] ? r2 : r3));
    if (sp <= kStackLDS - 3) {
        stackLane[sp * 64] = r3;
        sp += push3 ? 1 : 0;
        stackLane[sp * 64] = r2;
        sp += push2 ? 1 : 0;
        stackLane[sp * 64] = r1;
        sp += push1 ? 1 : 0;
        const bool empty = !any && sp == 0;
        sp -= (!any && sp > 0) ? 1 : 0;
        const int popped = stackLane[(sp < kStackLDS - 1 ? sp : kStackLDS - 1) * 64];
        cur = any ? next : (empty ? kSentinel : popped);
    } else {
        if (push3) HR_PUSH(r3);
        if (push2) HR_PUSH(r2);
        if (push1) HR_PUSH(r1);
        if (any) {
            cur = next;
        } else {
            HR_POP();
        }
    }
}
#endif

// Per-ray constants of the slab test: 1 / d and the ray origin over d (relative to the frame grid's origin for 32-byte nodes)
HRD RayK rayFrame(const SceneDev &S, v3 o, float idx, float idy, float idz)
{
#if HR_NODE32
    return RayK{idx, idy, idz, (o.x - S.gridLo[0]) * idx, (o.y - S.gridLo[1]) * idy, (o.z - S.gridLo[2]) * idz};
#else
    (void)S;
    return RayK{idx, idy, idz, o.x * idx, o.y * idy, o.z * idz};
#endif
}

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
    const RayK rk = rayFrame(S, o, idx, idy, idz);
#if HR_NODE32
    const GridK gk = gridOf(S);
#endif
    while (cur != kSentinel) {
        // ---- inner nodes: descend until this lane holds a leaf (cur < 0) or runs out of work
        while (cur >= 0 && cur != kSentinel) {
            if (STATS) ++nodeVisits;
#if HR_NODE32
            nodeStep4(S.nodes, cur, sp, stackLane, ovf, rk, gk, tmin, tlim);
#else
            nodeStep4(S.nodes, cur, sp, stackLane, ovf, rk, tmin, tlim);
#endif
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
}

} // namespace hr
