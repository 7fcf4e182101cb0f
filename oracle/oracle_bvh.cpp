// oracle_bvh.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// What OpenRL does behind rlDrawElements / rlRenderFrame
// (/root/reference/Source/HeatrayRenderer/Scene/Mesh.cpp:152,
//  Source/HeatrayRenderer/PassGenerator.cpp:386): acceleration structure and ray /
// triangle intersection.  OpenRL is closed, so the algorithm is this repo's stated
// spec (SURVEY §8a row a6, §8d "binary LBVH, <= 4 tris/leaf"):
//
//   * the HIT is defined independently of the acceleration structure: Möller–Trumbore
//     in the operation order below, accepted for t in (tmin, tmax) when the hit point
//     o + t d lies inside the triangle's bounding box grown by half the leaf padding
//     (hitInTriBox), closest hit = lexicographic minimum of (t, triangle id).  A
//     brute-force loop over all triangles (ctx.brute) gives the same answer as any
//     conservative BVH — also for the phantom hits float32 Möller–Trumbore produces
//     on sliver triangles, which the box condition turns away for everybody.
//   * the BVH here is the LBVH spec the node-visit / triangle-test figures V and T of
//     the roofline model are counted on: 30-bit Morton codes of the triangle-AABB
//     centres normalised by the scene AABB, stable sort, Karras radix tree,
//     subtrees of <= 4 triangles collapsed into leaves, boxes padded by
//     1e-5 * |scene diagonal| so that the slab test is conservative w.r.t. the
//     rounding of the Möller–Trumbore hit.
//
// PARITY STATUS: "parity unpinned" — OpenRL's intersector (epsilon, tie-breaking, watertightness) is a closed binary
// with no tests in the reference; the definitions below are this repository's stated spec.
#include "oracle_internal.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace ora {

static const int kLeafMax = 4;

static inline uint32_t expandBits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static inline uint32_t quantize10(float c, float lo, float ext)
{
    float q = (ext > 0.0f) ? (c - lo) / ext : 0.0f;
    q = q * 1024.0f;
    q = fmin_(fmax_(q, 0.0f), 1023.0f);
    return (uint32_t)q;
}

struct Box {
    float lo[3], hi[3];
};
static inline void boxUnion(Box &a, const Box &b)
{
    for (int k = 0; k < 3; ++k) {
        a.lo[k] = fmin_(a.lo[k], b.lo[k]);
        a.hi[k] = fmax_(a.hi[k], b.hi[k]);
    }
}
static inline Box triBox(const Tri &t, float pad)
{
    const vec3 p1 = t.v0 + t.e1, p2 = t.v0 + t.e2;
    const vec3 lo = min3(min3(t.v0, p1), p2), hi = max3(max3(t.v0, p1), p2);
    Box b;
    b.lo[0] = lo.x - pad, b.lo[1] = lo.y - pad, b.lo[2] = lo.z - pad;
    b.hi[0] = hi.x + pad, b.hi[1] = hi.y + pad, b.hi[2] = hi.z + pad;
    return b;
}

namespace {
struct Builder {
    const std::vector<uint32_t> &keys;
    int n;
    std::vector<int> left, right, first, last; // Karras internal nodes 0..n-2
    explicit Builder(const std::vector<uint32_t> &k) : keys(k), n((int)k.size()) {}
    // longest common prefix of the (key, index) pairs; -1 outside the range
    inline int delta(int i, int j) const
    {
        if (j < 0 || j >= n) return -1;
        uint32_t a = keys[i], b = keys[j];
        if (a == b) return 32 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
        return __builtin_clz(a ^ b);
    }
    void run()
    {
        left.resize(n - 1), right.resize(n - 1), first.resize(n - 1), last.resize(n - 1);
        for (int i = 0; i < n - 1; ++i) {
            int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1;
            int dmin = delta(i, i - d);
            int lmax = 2;
            while (delta(i, i + lmax * d) > dmin) lmax *= 2;
            int l = 0;
            for (int t = lmax / 2; t >= 1; t /= 2)
                if (delta(i, i + (l + t) * d) > dmin) l += t;
            int j = i + l * d;
            int dnode = delta(i, j);
            int s = 0;
            int t = l;
            do {
                t = (t + 1) >> 1;
                if (delta(i, i + (s + t) * d) > dnode) s += t;
            } while (t > 1);
            int gamma = i + s * d + std::min(d, 0);
            int lo = std::min(i, j), hi = std::max(i, j);
            // leaves are encoded as ~leafIndex
            left[i] = (lo == gamma) ? ~gamma : gamma;
            right[i] = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
            first[i] = lo, last[i] = hi;
        }
    }
};
} // namespace

void buildLBVH(const std::vector<Tri> &trisIn, Bvh &bvh)
{
    const int n = (int)trisIn.size();
    bvh.nodes.clear();
    bvh.order.resize(n);
    bvh.tris.resize(n);
    bvh.rootLeafCount = 0;
    if (n == 0) return;

    // scene bounds (unpadded) for the Morton normalisation and the pad
    vec3 lo(INFINITY), hi(-INFINITY);
    for (const Tri &t : trisIn) {
        const vec3 p1 = t.v0 + t.e1, p2 = t.v0 + t.e2;
        lo = min3(lo, min3(min3(t.v0, p1), p2));
        hi = max3(hi, max3(max3(t.v0, p1), p2));
    }
    bvh.lo[0] = lo.x, bvh.lo[1] = lo.y, bvh.lo[2] = lo.z;
    bvh.hi[0] = hi.x, bvh.hi[1] = hi.y, bvh.hi[2] = hi.z;
    const vec3 ext = hi - lo;
    const float pad = 1e-5f * length(ext);

    std::vector<uint32_t> code(n);
    for (int i = 0; i < n; ++i) {
        const Tri &t = trisIn[i];
        const vec3 p1 = t.v0 + t.e1, p2 = t.v0 + t.e2;
        const vec3 bl = min3(min3(t.v0, p1), p2), bh = max3(max3(t.v0, p1), p2);
        const vec3 c = (bl + bh) * 0.5f;
        code[i] = (expandBits10(quantize10(c.x, lo.x, ext.x)) << 2) | (expandBits10(quantize10(c.y, lo.y, ext.y)) << 1) |
                  expandBits10(quantize10(c.z, lo.z, ext.z));
    }
    std::iota(bvh.order.begin(), bvh.order.end(), 0u);
    std::stable_sort(bvh.order.begin(), bvh.order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    std::vector<uint32_t> keys(n);
    std::vector<Box> boxes(n);
    for (int i = 0; i < n; ++i) {
        keys[i] = code[bvh.order[i]];
        bvh.tris[i] = trisIn[bvh.order[i]];
        boxes[i] = triBox(bvh.tris[i], pad);
    }
    if (n <= kLeafMax) {
        bvh.rootLeafCount = n;
        return;
    }
    Builder b(keys);
    b.run();

    // Emit collapsed nodes in DFS pre-order.  ref >= 0: Karras internal node, ref < 0: leaf ~index.
    struct Range {
        int first, last;
    };
    auto rangeOf = [&](int ref) { return ref < 0 ? Range{~ref, ~ref} : Range{b.first[ref], b.last[ref]}; };
    auto rangeBox = [&](Range r) {
        Box bx = boxes[r.first];
        for (int i = r.first + 1; i <= r.last; ++i) boxUnion(bx, boxes[i]);
        return bx;
    };
    struct Item {
        int karras;
        int outIndex;
    };
    std::vector<Item> stack;
    bvh.nodes.emplace_back();
    stack.push_back({0, 0});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        const int refs[2] = {b.left[it.karras], b.right[it.karras]};
        for (int c = 0; c < 2; ++c) {
            Range r = rangeOf(refs[c]);
            Box bx = rangeBox(r);
            BvhNode &node = bvh.nodes[it.outIndex];
            for (int k = 0; k < 3; ++k) node.lo[c][k] = bx.lo[k], node.hi[c][k] = bx.hi[k];
            const int count = r.last - r.first + 1;
            if (count <= kLeafMax) {
                bvh.nodes[it.outIndex].child[c] = ~(r.first | ((count - 1) << 28));
            } else {
                const int idx = (int)bvh.nodes.size();
                bvh.nodes[it.outIndex].child[c] = idx;
                bvh.nodes.emplace_back();
                stack.push_back({refs[c], idx});
            }
        }
    }
}

// Möller–Trumbore, operation order is part of the spec.
static inline bool intersectTri(const Tri &tr, vec3 o, vec3 d, float &t, float &u, float &v)
{
    const vec3 pvec = cross(d, tr.e2);
    const float det = dot(tr.e1, pvec);
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const vec3 tvec = o - tr.v0;
    u = dot(tvec, pvec) * inv;
    if (!(u >= 0.0f) || u > 1.0f) return false;
    const vec3 qvec = cross(tvec, tr.e1);
    v = dot(d, qvec) * inv;
    if (!(v >= 0.0f) || u + v > 1.0f) return false;
    t = dot(tr.e2, qvec) * inv;
    return true;
}

// The hit test's second half: a Möller–Trumbore candidate is a hit only if its hit point o + t d lies inside the
// triangle's own bounding box grown by half the leaf padding.  float32 Möller–Trumbore alone accepts, about once in 10^9 rays, a ray
// that passes a sliver triangle at a distance, and whether a traversal ever TESTS that triangle depends on the boxes of its tree;
// a hit point inside the half-padded box lies inside every box a conservative tree puts around the triangle (those contain the box
// grown by the whole padding), with half a padding to spare for the rounding of the slab tests: every traversal of every tree — and
// the brute-force loop — gives the same answer.
static inline bool hitInTriBoxAxis(float v0, float e1, float e2, float o, float d, float t, float h)
{
    const float p1 = v0 + e1, p2 = v0 + e2;
    const float lo = fmin_(fmin_(v0, p1), p2), hi = fmax_(fmax_(v0, p1), p2);
    const float P = o + t * d;
    return P >= lo - h && P <= hi + h;
}
static inline bool hitInTriBox(const Tri &tr, vec3 o, vec3 d, float t, float h)
{
    return hitInTriBoxAxis(tr.v0.x, tr.e1.x, tr.e2.x, o.x, d.x, t, h) && hitInTriBoxAxis(tr.v0.y, tr.e1.y, tr.e2.y, o.y, d.y, t, h) &&
           hitInTriBoxAxis(tr.v0.z, tr.e1.z, tr.e2.z, o.z, d.z, t, h);
}

// Reciprocal direction for the slab test only: components with |d| < 1e-20 are replaced
// so that no infinity / NaN enters the box test (the hit itself never uses this).
static inline float safeInv(float d)
{
    const float lim = 1e-20f;
    if (fabsf(d) < lim) d = (d < 0.0f) ? -lim : lim;
    return 1.0f / d;
}

static inline bool slab(const float *lo, const float *hi, vec3 o, vec3 id, float tmin, float tmax, float &tnear)
{
    float t0 = (lo[0] - o.x) * id.x, t1 = (hi[0] - o.x) * id.x;
    float tn = fmin_(t0, t1), tf = fmax_(t0, t1);
    t0 = (lo[1] - o.y) * id.y, t1 = (hi[1] - o.y) * id.y;
    tn = fmax_(tn, fmin_(t0, t1)), tf = fmin_(tf, fmax_(t0, t1));
    t0 = (lo[2] - o.z) * id.z, t1 = (hi[2] - o.z) * id.z;
    tn = fmax_(tn, fmin_(t0, t1)), tf = fmin_(tf, fmax_(t0, t1));
    tnear = tn;
    return tn <= tf && tf >= tmin && tn <= tmax;
}

template <bool ANY>
static bool traverse(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, TraceCounters *tc, Hit &best)
{
    const Bvh &bvh = ctx.bvh;
    const int n = (int)bvh.tris.size();
    best.prim = -1;
    best.t = tmax;
    if (n == 0) return false;
    auto testRange = [&](int first, int count) -> bool {
        for (int i = first; i < first + count; ++i) {
            const int prim = (int)bvh.order[i];
            if (tc) tc->triTests++;
            if (prim == skip) continue;
            float t, u, v;
            if (!intersectTri(bvh.tris[i], o, d, t, u, v)) continue;
            if (!(t > tmin) || !(t < tmax)) continue;
            if (!hitInTriBox(bvh.tris[i], o, d, t, ctx.hitPad)) continue;
            if (ANY) {
                if ((ctx.attrs[prim].flags & TF_NON_OCCLUDER) && alphaPasses(ctx, prim, u, v)) continue;
                best.prim = prim, best.t = t, best.u = u, best.v = v;
                return true;
            }
            if (best.prim < 0 || t < best.t || (t == best.t && prim < best.prim)) best.prim = prim, best.t = t, best.u = u, best.v = v;
        }
        return false;
    };
    if (bvh.rootLeafCount > 0) {
        testRange(0, bvh.rootLeafCount);
        return best.prim >= 0;
    }
    const vec3 id(safeInv(d.x), safeInv(d.y), safeInv(d.z));
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const int ni = stack[--sp];
        if (ni < 0) {
            const int enc = ~ni;
            if (testRange(enc & 0x0FFFFFFF, (enc >> 28) + 1) && ANY) return true;
            continue;
        }
        const BvhNode &node = bvh.nodes[ni];
        if (tc) tc->nodeVisits++;
        const float lim = ANY ? tmax : ((best.prim >= 0) ? best.t : tmax);
        float tn0, tn1;
        const bool h0 = slab(node.lo[0], node.hi[0], o, id, tmin, lim, tn0);
        const bool h1 = slab(node.lo[1], node.hi[1], o, id, tmin, lim, tn1);
        if (h0 && h1) {
            const bool firstIs0 = tn0 <= tn1;
            stack[sp++] = node.child[firstIs0 ? 1 : 0];
            stack[sp++] = node.child[firstIs0 ? 0 : 1];
        } else if (h0) {
            stack[sp++] = node.child[0];
        } else if (h1) {
            stack[sp++] = node.child[1];
        }
    }
    return best.prim >= 0;
}

template <bool ANY> static bool bruteForce(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, Hit &best)
{
    best.prim = -1;
    best.t = tmax;
    const int n = (int)ctx.tris.size();
    for (int prim = 0; prim < n; ++prim) {
        if (prim == skip) continue;
        float t, u, v;
        if (!intersectTri(ctx.tris[prim], o, d, t, u, v)) continue;
        if (!(t > tmin) || !(t < tmax)) continue;
        if (!hitInTriBox(ctx.tris[prim], o, d, t, ctx.hitPad)) continue;
        if (ANY) {
            if ((ctx.attrs[prim].flags & TF_NON_OCCLUDER) && alphaPasses(ctx, prim, u, v)) continue;
            best.prim = prim, best.t = t, best.u = u, best.v = v;
            return true;
        }
        if (best.prim < 0 || t < best.t) best.prim = prim, best.t = t, best.u = u, best.v = v; // ascending id: ties keep the lower id
    }
    return best.prim >= 0;
}

Hit traceClosest(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, TraceCounters *tc, bool brute)
{
    Hit h;
    if (brute)
        bruteForce<false>(ctx, o, d, tmin, tmax, skip, h);
    else
        traverse<false>(ctx, o, d, tmin, tmax, skip, tc, h);
    return h;
}

bool traceOccluded(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, TraceCounters *tc, bool brute)
{
    Hit h;
    return brute ? bruteForce<true>(ctx, o, d, tmin, tmax, skip, h) : traverse<true>(ctx, o, d, tmin, tmax, skip, tc, h);
}

} // namespace ora
