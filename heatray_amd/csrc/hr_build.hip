// hr_build.hip — scene assembly and LBVH construction on the GPU, plus the sample-table generators.
//
// Replaces what OpenRL does with the buffers of rlDrawElements (/root/reference/Source/HeatrayRenderer/
// Scene/Mesh.cpp:29-153, Resources/shaders/vertex.rlsl:25-43) — SURVEY §8a row a6 — and the host-side table
// generators (Source/Utility/Random.h:85-289, Source/HeatrayRenderer/Materials/MultiScatterUtil.cpp:20-139).
//
//   assemble : one thread per triangle: world transform, edge form, shading attributes, scene bounds
//   morton   : 30-bit Morton code of the triangle-AABB centre, normalised by the scene bounds
//   sort     : LSD radix sort, 4 x 8-bit digits, stable (histogram / scan / ballot-ranked scatter)
//   karras   : binary radix tree over the sorted (key, index) pairs
//   refit    : bottom-up boxes in kernel-ordered rounds (no inter-workgroup hand-off inside a launch,
//              so no reliance on cross-XCD visibility)
//   emit     : subtrees of <= 4 triangles collapse into leaves; 64-byte two-box nodes
#include "hr_kernels.h"
#include "hr_tables.h"
#include <cstring>
#include "hr_texture.h"

#include <vector>

namespace hr {


#define HR_CHECK(expr)                  \
    do {                                \
        hipError_t e_ = (expr);         \
        if (e_ != hipSuccess) return 1; \
    } while (0)

HRD uint32_t orderedFromFloat(float f)
{
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

HRD v3 fetch3(const float *a, uint32_t i, int stride) { const float *p = a + (size_t)i * stride; return v3(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------- assemble
__global__ __launch_bounds__(256) void k_assemble(const GeomDev *__restrict__ geoms, int nGeoms, uint32_t nTris, Tri *__restrict__ tris,
                                                  const uint32_t *__restrict__ slotOfPrim, TriAttr *__restrict__ attrs,
                                                  TriAttrExt *__restrict__ ext, uint32_t *__restrict__ bounds)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    v3 lo(__builtin_inff()), hi(-__builtin_inff());
    if (t < nTris) {
        int a = 0, b = nGeoms - 1; // last geometry with triOffset <= t
        while (a < b) {
            int m = (a + b + 1) >> 1;
            if (geoms[m].triOffset <= t)
                a = m;
            else
                b = m - 1;
        }
        const GeomDev &g = geoms[a];
        const uint32_t lt = t - g.triOffset;
        uint32_t i0, i1, i2;
        if (g.strip) { // GL strip ordering: odd triangles swap the first two vertices
            i0 = g.idx[lt], i1 = g.idx[lt + 1], i2 = g.idx[lt + 2];
            if (lt & 1u) {
                uint32_t s = i0;
                i0 = i1;
                i1 = s;
            }
        } else {
            i0 = g.idx[3 * lt], i1 = g.idx[3 * lt + 1], i2 = g.idx[3 * lt + 2];
        }
        const uint32_t id[3] = {i0, i1, i2};
        // vertex.rlsl:27 — rl_Position = worldFromEntity * vec4(position, 1)
        const v3 p0 = xformPoint(g.world, fetch3(g.pos, i0, g.posStride)), p1 = xformPoint(g.world, fetch3(g.pos, i1, g.posStride)),
                 p2 = xformPoint(g.world, fetch3(g.pos, i2, g.posStride));
        const v3 e1 = p1 - p0, e2 = p2 - p0;
        Tri tr;
        tr.p = make_float4(p0.x, p0.y, p0.z, e1.x);
        tr.q = make_float4(e1.y, e1.z, e2.x, e2.y);
        tr.r = make_float4(e2.z, __uint_as_float(t), __uint_as_float(g.flags), 0.0f);
        tris[slotOfPrim ? slotOfPrim[t] : t] = tr;
        TriAttr at;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // vertex.rlsl:28-29 — normal = mat3(worldFromEntity) * normalAttribute
            const v3 n = xformVector(g.world, fetch3(g.nrm, id[k], g.nrmStride));
            at.n[3 * k] = n.x, at.n[3 * k + 1] = n.y, at.n[3 * k + 2] = n.z;
            at.uv[2 * k] = g.uv ? g.uv[(size_t)id[k] * g.uvStride] : 0.0f;
            at.uv[2 * k + 1] = g.uv ? g.uv[(size_t)id[k] * g.uvStride + 1] : 0.0f;
        }
        at.matflags = (g.material & kMatMask) | (g.flags << 24);
        attrs[t] = at;
        if (ext) {
            TriAttrExt e;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v3 tn(0.0f), bt(0.0f), cl(0.0f);
                if (g.tan && g.bit) { // vertex.rlsl:35-38
                    tn = xformVector(g.world, fetch3(g.tan, id[k], g.tanStride));
                    bt = xformVector(g.world, fetch3(g.bit, id[k], g.bitStride));
                }
                if (g.col) cl = fetch3(g.col, id[k], g.colStride); // vertex.rlsl:40-42
                e.tan[3 * k] = tn.x, e.tan[3 * k + 1] = tn.y, e.tan[3 * k + 2] = tn.z;
                e.bit[3 * k] = bt.x, e.bit[3 * k + 1] = bt.y, e.bit[3 * k + 2] = bt.z;
                e.col[3 * k] = cl.x, e.col[3 * k + 1] = cl.y, e.col[3 * k + 2] = cl.z;
            }
            e.pad = 0.0f;
            ext[t] = e;
        }
        // bounds over v0, v0+e1, v0+e2 (the vertices the intersector sees)
        const v3 q1 = p0 + e1, q2 = p0 + e2;
        lo = min3(min3(p0, q1), q2);
        hi = max3(max3(p0, q1), q2);
    }
    // wave reduction, then block reduction through LDS, then one atomic per workgroup and component into one of kBoundSlots copies
    // of the bounds (same-address atomics serialise: one per wave into a single copy cost 0.9 ms for a million triangles)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo.x = fmin_(lo.x, __shfl_xor(lo.x, o)), lo.y = fmin_(lo.y, __shfl_xor(lo.y, o)), lo.z = fmin_(lo.z, __shfl_xor(lo.z, o));
        hi.x = fmax_(hi.x, __shfl_xor(hi.x, o)), hi.y = fmax_(hi.y, __shfl_xor(hi.y, o)), hi.z = fmax_(hi.z, __shfl_xor(hi.z, o));
    }
    __shared__ float red[4][6];
    if ((threadIdx.x & 63) == 0) {
        float *r = red[threadIdx.x >> 6];
        r[0] = lo.x, r[1] = lo.y, r[2] = lo.z, r[3] = hi.x, r[4] = hi.y, r[5] = hi.z;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[0][k];
        for (int w = 1; w < 4; ++w) v = k < 3 ? fmin_(v, red[w][k]) : fmax_(v, red[w][k]);
        uint32_t *slot = bounds + 6 * (blockIdx.x & (kBoundSlots - 1));
        if (k < 3) {
            if (v != __builtin_inff()) atomicMin(&slot[k], orderedFromFloat(v));
        } else {
            if (v != -__builtin_inff()) atomicMax(&slot[k], orderedFromFloat(v));
        }
    }
}

__global__ void k_bounds_init(uint32_t *bounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * kBoundSlots) bounds[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
}

void launchAssemble(hipStream_t st, const GeomDev *geoms, int nGeoms, uint32_t nTris, Tri *tris, const uint32_t *slotOfPrim, TriAttr *attrs,
                    TriAttrExt *ext, uint32_t *bounds)
{
    if (nTris == 0) return;
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(6 * kBoundSlots), 0, st, bounds);
    hipLaunchKernelGGL(k_assemble, dim3((nTris + 255) / 256), dim3(256), 0, st, geoms, nGeoms, nTris, tris, slotOfPrim, attrs, ext, bounds);
}

HRD float floatFromOrderedDev(uint32_t u)
{
    const uint32_t b = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
    return __uint_as_float(b);
}
__global__ void k_scene_consts(const uint32_t *__restrict__ bounds, SceneConsts *__restrict__ out, SceneDev *__restrict__ scene)
{
    SceneConsts c;
    for (int k = 0; k < 3; ++k) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int sl = 0; sl < kBoundSlots; ++sl) {
            const uint32_t a = bounds[6 * sl + k], b = bounds[6 * sl + 3 + k];
            lo = a < lo ? a : lo, hi = b > hi ? b : hi;
        }
        c.lo[k] = floatFromOrderedDev(lo), c.hi[k] = floatFromOrderedDev(hi);
    }
    // |hi - lo| with the contract's operation order: sqrt((x*x + y*y) + z*z), correctly rounded like the host's sqrtf
    const float ex = c.hi[0] - c.lo[0], ey = c.hi[1] - c.lo[1], ez = c.hi[2] - c.lo[2];
    c.diag = sqrt_(ex * ex + ey * ey + ez * ez);
    c.pad = 1e-5f * c.diag;
    c.eps = 1e-4f * c.diag;
    c.areaSum = 0.0f, c.triAreaSum = 0.0f, c.pad1 = 0u;
    *out = c;
    if (scene) scene->rayEps = c.eps, scene->hitPad = 0.5f * c.pad;
}
void launchSceneConsts(hipStream_t st, const uint32_t *bounds, SceneConsts *out, SceneDev *scene)
{
    hipLaunchKernelGGL(k_scene_consts, dim3(1), dim3(1), 0, st, bounds, out, scene);
}

// --------------------------------------------------------------------------------------- morton
HRD uint32_t expandBits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
HRD uint32_t quantize10(float c, float lo, float ext)
{
    float q = (ext > 0.0f) ? (c - lo) / ext : 0.0f;
    q = q * 1024.0f;
    q = fmin_(fmax_(q, 0.0f), 1023.0f);
    return (uint32_t)q;
}
HRD void triBounds(const Tri &t, v3 &bl, v3 &bh)
{
    const v3 v0(t.p.x, t.p.y, t.p.z), e1(t.p.w, t.q.x, t.q.y), e2(t.q.z, t.q.w, t.r.x);
    const v3 p1 = v0 + e1, p2 = v0 + e2;
    bl = min3(min3(v0, p1), p2);
    bh = max3(max3(v0, p1), p2);
}

__global__ __launch_bounds__(256) void k_morton(const Tri *__restrict__ tris, uint32_t n, v3 lo, v3 ext, uint32_t *__restrict__ keys,
                                                uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    v3 bl, bh;
    triBounds(tris[i], bl, bh);
    const v3 c = (bl + bh) * 0.5f;
    keys[i] = (expandBits10(quantize10(c.x, lo.x, ext.x)) << 2) | (expandBits10(quantize10(c.y, lo.y, ext.y)) << 1) |
              expandBits10(quantize10(c.z, lo.z, ext.z));
    vals[i] = i;
}

// ----------------------------------------------------------------------------------- radix sort
static const int kSortTile = 2048; // keys per single-wave workgroup

__global__ __launch_bounds__(64) void k_sort_hist(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t *__restrict__ blockHist,
                                                  uint32_t nBlocks)
{
    __shared__ uint32_t hist[256];
    for (int i = threadIdx.x; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / 64; ++r) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) blockHist[(uint32_t)i * nBlocks + blockIdx.x] = hist[i];
}

// exclusive scan of `n` values by ONE workgroup of 1024 threads (n is at most a few hundred thousand)
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t *__restrict__ data, uint32_t n, uint32_t *__restrict__ total)
{
    __shared__ uint32_t waveSums[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < n) ? data[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(inc, o);
            if ((int)lane >= o) inc += t;
        }
        if (lane == 63) waveSums[wave] = inc;
        __syncthreads();
        uint32_t wavePrefix = 0;
        for (uint32_t w = 0; w < wave; ++w) wavePrefix += waveSums[w];
        const uint32_t c = carry;
        if (i < n) data[i] = c + wavePrefix + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wavePrefix + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(64) void k_sort_scatter(const uint32_t *__restrict__ keysIn, const uint32_t *__restrict__ valsIn, uint32_t n, int shift,
                                                     const uint32_t *__restrict__ blockOffsets, uint32_t nBlocks, uint32_t *__restrict__ keysOut,
                                                     uint32_t *__restrict__ valsOut)
{
    __shared__ uint32_t off[256];
    for (int i = threadIdx.x; i < 256; i += 64) off[i] = blockOffsets[(uint32_t)i * nBlocks + blockIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / 64; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? keysIn[i] : 0u, val = valid ? valsIn[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        // lanes holding the same digit (stable: rank = number of equal-digit lanes before me)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        uint32_t pos = 0;
        if (valid) pos = off[d] + (uint32_t)__popcll(same & ltMask);
        __syncthreads();
        if (valid && (same & ltMask) == 0ull) off[d] += (uint32_t)__popcll(same); // first lane of each digit group
        __syncthreads();
        if (valid) {
            keysOut[pos] = key;
            valsOut[pos] = val;
        }
    }
}

__global__ __launch_bounds__(256) void k_gather_tris(const Tri *__restrict__ trisPrim, const uint32_t *__restrict__ order, uint32_t n,
                                                     Tri *__restrict__ sorted, float pad, Box6 *__restrict__ leafBox)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Tri t = trisPrim[order[i]];
    sorted[i] = t;
    v3 bl, bh;
    triBounds(t, bl, bh);
    Box6 b;
    b.lo[0] = bl.x - pad, b.lo[1] = bl.y - pad, b.lo[2] = bl.z - pad;
    b.hi[0] = bh.x + pad, b.hi[1] = bh.y + pad, b.hi[2] = bh.z + pad;
    leafBox[i] = b;
}

// --------------------------------------------------------------------------------------- karras
HRD int deltaKey(const uint32_t *keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
    return __clz((int)(a ^ b));
}

#ifndef HR_COLLAPSE_DP
#define HR_COLLAPSE_DP 1 // collapse the binary tree to 4-wide nodes by minimal summed node area (0: open the largest child twice, rounds 1-3a)
#endif
#ifndef HR_ROTATE_SWEEPS
#define HR_ROTATE_SWEEPS 0 // tree-rotation sweeps over the binary tree before the collapse (experiment knob)
#endif
struct KNode {
    int left, right; // >= 0: internal node, < 0: leaf ~index
    int first, last; // covered range of sorted triangles
};

__global__ __launch_bounds__(256) void k_karras(const uint32_t *__restrict__ keys, int n, KNode *__restrict__ nodes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (deltaKey(keys, n, i, i + 1) - deltaKey(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = deltaKey(keys, n, i, i - d);
    int lmax = 2;
    while (deltaKey(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (deltaKey(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = deltaKey(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (deltaKey(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    KNode k;
    k.left = (lo == gamma) ? ~gamma : gamma;
    k.right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    k.first = lo, k.last = hi;
    nodes[i] = k;
}

// One bottom-up round: a node whose children were finished in an EARLIER launch gets its box.
// stamp[i] = round in which node i was finished (0 = not yet); visibility comes from the kernel boundary.
HRD float boxArea(const Box6 &b);
HRD float4 collapseCost(int left, int right, float area, const float4 *__restrict__ cost);
// ... and the costs of the collapse to a 4-wide tree (k_collapse4): cost[i] = (C1, C2, C3, C4), Ck = the least sum of the surface areas of
// the 4-wide nodes that can represent the subtree of binary node i when it may occupy up to k child slots of its 4-wide parent
// (k = 1: it is a child itself, a node of its own; k >= 2: it may be opened and its two children share the slots).  A triangle costs
// nothing: it is a child slot wherever it ends up.  Surface area = the probability that a ray visits the node (SAH).
__global__ __launch_bounds__(256) void k_refit_round(const KNode *__restrict__ nodes, int nInternal, const Box6 *__restrict__ leafBox,
                                                     Box6 *__restrict__ nodeBox, uint32_t *__restrict__ stamp, uint32_t round, float4 *__restrict__ cost)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nInternal || stamp[i] != 0u) return;
    const KNode k = nodes[i];
    const bool lr = k.left < 0 || (stamp[k.left] != 0u && stamp[k.left] < round);
    const bool rr = k.right < 0 || (stamp[k.right] != 0u && stamp[k.right] < round);
    if (!(lr && rr)) return;
    const Box6 a = k.left < 0 ? leafBox[~k.left] : nodeBox[k.left];
    const Box6 b = k.right < 0 ? leafBox[~k.right] : nodeBox[k.right];
    Box6 u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        u.lo[c] = fmin_(a.lo[c], b.lo[c]);
        u.hi[c] = fmax_(a.hi[c], b.hi[c]);
    }
    nodeBox[i] = u;
    if (cost) cost[i] = collapseCost(k.left, k.right, boxArea(u), cost);
    stamp[i] = round;
}

// Tree rotations (Kensler 2008) on the binary tree before it is collapsed: node N with children L, R may exchange R with one of
// L's children (or L with one of R's) when that shrinks the surface area of the rebuilt child — the LBVH splits by Morton prefix,
// blind to the boxes, and a rotation repairs the worst of it.  One launch handles the nodes of ONE stamp value (the round of the
// bottom-up refit that finished them: a parent's stamp is larger than its children's), bottom-up, so that no two nodes of a launch
// are parent and child and the subtrees below are final: no locks.  Hits do not depend on the tree.
__global__ __launch_bounds__(256) void k_rotate(KNode *__restrict__ nodes, int nInternal, const Box6 *__restrict__ leafBox, Box6 *__restrict__ nodeBox,
                                                const uint32_t *__restrict__ stamp, uint32_t which, uint32_t *__restrict__ applied)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nInternal || stamp[i] != which) return;
    const KNode N = nodes[i];
    auto boxOf = [&](int ref) { return ref < 0 ? leafBox[~ref] : nodeBox[ref]; };
    auto unite = [](const Box6 &a, const Box6 &b) {
        Box6 u;
#pragma unroll
        for (int c = 0; c < 3; ++c) u.lo[c] = fmin_(a.lo[c], b.lo[c]), u.hi[c] = fmax_(a.hi[c], b.hi[c]);
        return u;
    };
    // option k: 0/1 = R <-> L.left / L.right, 2/3 = L <-> R.left / R.right; gain = area of the child before - after
    float bestGain = 0.0f;
    int best = -1;
    Box6 bestBox;
    for (int side = 0; side < 2; ++side) {
        const int inner = side == 0 ? N.left : N.right, other = side == 0 ? N.right : N.left;
        if (inner < 0) continue;
        const KNode A = nodes[inner];
        const float before = boxArea(nodeBox[inner]);
        const Box6 bo = boxOf(other), b1 = boxOf(A.left), b2 = boxOf(A.right);
        const Box6 u0 = unite(bo, b2), u1 = unite(b1, bo); // `other` takes the place of A.left / of A.right
        const float g0 = before - boxArea(u0), g1 = before - boxArea(u1);
        if (g0 > bestGain) bestGain = g0, best = 2 * side, bestBox = u0;
        if (g1 > bestGain) bestGain = g1, best = 2 * side + 1, bestBox = u1;
    }
    if (best < 0) return;
    const int side = best >> 1;
    const int inner = side == 0 ? N.left : N.right, other = side == 0 ? N.right : N.left;
    KNode A = nodes[inner];
    int moved; // the grandchild that becomes N's direct child
    if (best & 1)
        moved = A.right, A.right = other;
    else
        moved = A.left, A.left = other;
    KNode M = N;
    if (side == 0)
        M.right = moved;
    else
        M.left = moved;
    nodes[inner] = A;
    nodes[i] = M;
    nodeBox[inner] = bestBox;
    atomicAdd(applied, 1u);
}

// Collapse the binary tree into 4-wide nodes, one tree level per launch.  binOf[i] is the binary (Karras)
// node that 4-wide node i stands for.  Starting from its two children, the inner child with the largest
// surface area is replaced by its own two children until four children exist (or none can be opened).
// Leaves hold ONE triangle.  The inner children of a node are appended together through one atomic reservation
// (siblings become neighbours in memory), and so are its triangles, which are copied to their final place.
HRD float boxArea(const Box6 &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// (C1, C2, C3, C4) of a binary node from its children's (see k_refit_round); a triangle child (ref < 0) costs nothing
HRD float4 collapseCost(int left, int right, float area, const float4 *__restrict__ cost)
{
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 cl = left < 0 ? z : cost[left], cr = right < 0 ? z : cost[right];
    const float h2 = cl.x + cr.x;
    const float h3 = fmin_(cl.x + cr.y, cl.y + cr.x);
    const float h4 = fmin_(cl.x + cr.z, fmin_(cl.y + cr.y, cl.z + cr.x));
    const float c1 = area + h4;
    const float c2 = fmin_(c1, h2), c3 = fmin_(c2, h3), c4 = fmin_(c3, h4);
    return make_float4(c1, c2, c3, c4);
}

// biased exponent e (1..254) of the smallest power of two s = 2^(e-127) with ext / s <= 255
HRD uint32_t quantExponent(float ext)
{
    const float f = ext * (1.0f / 255.0f);
    uint32_t bits = __float_as_uint(f);
    uint32_t e = (bits >> 23) & 0xFFu;
    if (bits & 0x007FFFFFu) e += 1;                   // round the scale up to a power of two
    if (__uint_as_float(e << 23) * 255.0f < ext) e += 1; // guard against the rounding of ext/255
    return e < 1u ? 1u : (e > 254u ? 254u : e);
}

// Quantise the child boxes of a node against the node's own box (origin + q * 2^e, 8 bits per plane, lo rounded down, hi rounded
// up) and pack the 64-byte record (hr_types.h).  Shared by the collapse and by the refit.
HRD Node4 encodeNode4(const Box6 *cb, const Box6 &nb, int nValid, int nInner, uint32_t innerBase, int leafKey)
{
    uint32_t e[3];
    float inv[3];
    for (int k = 0; k < 3; ++k) {
        e[k] = quantExponent(nb.hi[k] - nb.lo[k]);
        inv[k] = 1.0f / __uint_as_float(e[k] << 23);
    }
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        for (int k = 0; k < 3; ++k) {
            uint32_t lo8 = 255u, hi8 = 0u; // no child: inverted box (the traversal also checks the child count)
            if (c < nValid) {
                const float fl = floor_((cb[c].lo[k] - nb.lo[k]) * inv[k]);
                const float fh = __builtin_ceilf((cb[c].hi[k] - nb.lo[k]) * inv[k]);
                lo8 = (uint32_t)fmin_(fmax_(fl, 0.0f), 255.0f);
                hi8 = (uint32_t)fmin_(fmax_(fh, 0.0f), 255.0f);
            }
            qlo[k] |= lo8 << (8 * c);
            qhi[k] |= hi8 << (8 * c);
        }
    }
    Node4 nd;
    nd.a = make_float4(nb.lo[0], nb.lo[1], nb.lo[2],
                       __uint_as_float(e[0] | (e[1] << 8) | (e[2] << 16) | ((uint32_t)nInner << 24) | ((uint32_t)nValid << 27)));
    nd.b = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
    nd.c = make_uint4(qhi[1], qhi[2], innerBase, (uint32_t)leafKey);
    nd.d = make_uint4(0u, 0u, 0u, 0u);
    return nd;
}

__global__ __launch_bounds__(256) void k_collapse4(const KNode *__restrict__ knodes, const Box6 *__restrict__ leafBox,
                                                   const Box6 *__restrict__ nodeBox, int *__restrict__ binOf, uint32_t levelStart,
                                                   uint32_t levelEnd, uint32_t *__restrict__ counter, uint32_t *__restrict__ leafCounter,
                                                   const Tri *__restrict__ sorted, Tri *__restrict__ finalTris, Node4 *__restrict__ out,
                                                   Box6 *__restrict__ nodeBoxOut, const SceneConsts *__restrict__ consts, const float4 *__restrict__ cost)
{
    const uint32_t i = levelStart + blockIdx.x * 256 + threadIdx.x;
    if (i >= levelEnd) return;
    const int b = binOf[i];
    int cand[4];
    int n = 2;
    if (cost) {
        // The children that minimise the summed area of the nodes below (k_refit_round's costs): binary node b is opened over four slots;
        // a child with a budget of k slots stays one child (a node of its own, or a triangle) unless opening it over those slots is cheaper.
        auto costOf = [&](int ref, int k) -> float {
            if (ref < 0) return 0.0f;
            const float4 c = cost[ref];
            return k == 1 ? c.x : (k == 2 ? c.y : (k == 3 ? c.z : c.w));
        };
        int stackRef[4], stackK[4], sp = 0;
        n = 0;
        stackRef[sp] = b, stackK[sp] = 4, ++sp;
        bool first = true;
        while (sp > 0) {
            --sp;
            const int ref = stackRef[sp], k = stackK[sp];
            const bool open = first || (ref >= 0 && k >= 2 && costOf(ref, k) < costOf(ref, 1));
            first = false;
            if (!open) {
                cand[n++] = ref;
                continue;
            }
            const int L = knodes[ref].left, R = knodes[ref].right;
            int bestL = 1;
            float best = costOf(L, 1) + costOf(R, k - 1);
            for (int jl = 2; jl < k; ++jl) {
                const float c = costOf(L, jl) + costOf(R, k - jl);
                if (c < best) best = c, bestL = jl;
            }
            stackRef[sp] = R, stackK[sp] = k - bestL, ++sp;
            stackRef[sp] = L, stackK[sp] = bestL, ++sp;
        }
    } else
    {
    cand[0] = knodes[b].left, cand[1] = knodes[b].right;
    cand[2] = cand[3] = 0;
    for (int round = 0; round < 2; ++round) {
        int pick = -1;
        float bestArea = -1.0f;
        for (int c = 0; c < n; ++c) {
            const int ref = cand[c];
            if (ref < 0) continue; // a single triangle
            const float a = boxArea(nodeBox[ref]);
            if (a > bestArea) bestArea = a, pick = c;
        }
        if (pick < 0) break;
        const int ref = cand[pick];
        cand[pick] = knodes[ref].left;
        cand[n++] = knodes[ref].right;
    }
    }
    // inner children first, triangles after them (the order of the children inside a node is free)
    int ord[4];
    int nInner = 0;
    for (int c = 0; c < n; ++c)
        if (cand[c] >= 0) ord[nInner++] = cand[c];
    int nValid = nInner;
    for (int c = 0; c < n; ++c)
        if (cand[c] < 0) ord[nValid++] = cand[c];
    const int nLeaf = nValid - nInner;
    const uint32_t innerBase = nInner ? atomicAdd(counter, (uint32_t)nInner) : 0u;
    const uint32_t leafBase = nLeaf ? atomicAdd(leafCounter, (uint32_t)nLeaf) : 0u;
    Box6 cb[4];
    Box6 nb;
    for (int k = 0; k < 3; ++k) nb.lo[k] = __builtin_inff(), nb.hi[k] = -__builtin_inff();
    for (int c = 0; c < nValid; ++c) {
        const int ref = ord[c];
        if (ref < 0) {
            cb[c] = leafBox[~ref];
            finalTris[leafBase + (uint32_t)(nValid - 1 - c)] = sorted[~ref]; // descending with the child slot (see leafKey)
        } else {
            cb[c] = nodeBox[ref];
            binOf[innerBase + (uint32_t)c] = ref;
        }
        for (int k = 0; k < 3; ++k) nb.lo[k] = fmin_(nb.lo[k], cb[c].lo[k]), nb.hi[k] = fmax_(nb.hi[k], cb[c].hi[k]);
    }
    // child j >= nInner is triangle top - j with top = leafBase + nValid - 1: its reference ~(top - j) = ~top + j, so that a
    // child reference is `base + slot` for both kinds (base = innerBase or leafKey)
    const int leafKey = ~((int)leafBase + nValid - 1);
    out[i] = encodeNode4(cb, nb, nValid, nInner, innerBase, leafKey);
    nodeBoxOut[i] = nb;
}

// sum over the workgroup, then ONE float atomic (heuristic only: the order of the additions does not matter)
HRD void blockAreaAdd(float area, SceneConsts *consts, bool toTriangles = false)
{
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) area += __shfl_xor(area, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = area;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (part[0] + part[1]) + (part[2] + part[3]);
        if (t > 0.0f) atomicAdd(toTriangles ? &consts->triAreaSum : &consts->areaSum, t);
    }
}

// ---------------------------------------------------------------------------------------- refit
// One level of a bottom-up refit (levels are contiguous index ranges: nodes are allocated breadth-first).  A node's child
// boxes come from its triangles (leaf children; padded like at build time) and from the boxes its inner children wrote in
// the previous launch; its own box goes to nodeBox for the level above.  Same encoder as the build: conservative by construction.
__global__ __launch_bounds__(256) void k_refit4(Node4 *__restrict__ nodes, Box6 *__restrict__ nodeBox, const Tri *__restrict__ tris,
                                                uint32_t levelStart, uint32_t levelEnd, SceneConsts *__restrict__ consts)
{
    const uint32_t i = levelStart + blockIdx.x * 256 + threadIdx.x;
    float area = 0.0f;
    if (i < levelEnd) {
        const float pad = consts->pad;
        const Node4 nd = nodes[i];
        const uint32_t meta = __float_as_uint(nd.a.w);
        const int nInner = (int)((meta >> 24) & 7u), nValid = (int)(meta >> 27);
        const uint32_t innerBase = nd.c.z;
        const int leafKey = (int)nd.c.w;
        Box6 cb[4];
        Box6 nb;
        for (int k = 0; k < 3; ++k) nb.lo[k] = __builtin_inff(), nb.hi[k] = -__builtin_inff();
        for (int c = 0; c < nValid; ++c) {
            if (c < nInner) {
                cb[c] = nodeBox[innerBase + (uint32_t)c];
            } else {
                v3 bl, bh;
                triBounds(tris[~(leafKey + c)], bl, bh);
                cb[c].lo[0] = bl.x - pad, cb[c].lo[1] = bl.y - pad, cb[c].lo[2] = bl.z - pad;
                cb[c].hi[0] = bh.x + pad, cb[c].hi[1] = bh.y + pad, cb[c].hi[2] = bh.z + pad;
            }
            for (int k = 0; k < 3; ++k) nb.lo[k] = fmin_(nb.lo[k], cb[c].lo[k]), nb.hi[k] = fmax_(nb.hi[k], cb[c].hi[k]);
        }
        nodes[i] = encodeNode4(cb, nb, nValid, nInner, innerBase, leafKey);
        nodeBox[i] = nb;
        area = boxArea(nb);
    }
    blockAreaAdd(area, consts); // tree-quality heuristic: sum of node areas
}

// ---- the 32-byte nodes of k_trace (hr_types.h: Node32)
// biased exponent of the grid's cell along an axis of extent `ext`: the power of two with 256 cells > ext (so every frame origin is one of
// 256 grid points: a byte), kept inside the normal range with room for the scale exponents on both sides
HRD uint32_t gridCellExponent(float ext)
{
    const uint32_t e = (__float_as_uint(ext) >> 23) & 0xFFu; // ext < 2^(e - 126)
    const uint32_t c = e > 7u ? e - 7u : 0u;                  // 2^(c - 127) * 256 = 2^(e - 126) > ext
    return c < 24u ? 24u : (c > 230u ? 230u : c);
}
void gridOf(const SceneConsts &k, float gridLo[3], float gridCell[3], uint32_t gridCellExp[3])
{
    for (int a = 0; a < 3; ++a) {
        // (node boxes reach one leaf padding beyond the scene's bounds: the grid starts two paddings below them)
        gridLo[a] = k.lo[a] - 2.0f * k.pad;
        uint32_t bits;
        const float ext = (k.hi[a] + 2.0f * k.pad) - gridLo[a];
        std::memcpy(&bits, &ext, 4);
        const uint32_t e = (bits >> 23) & 0xFFu, c0 = e > 7u ? e - 7u : 0u, c = c0 < 24u ? 24u : (c0 > 230u ? 230u : c0);
        gridCellExp[a] = c;
        const uint32_t cb = c << 23;
        std::memcpy(&gridCell[a], &cb, 4);
    }
}
__global__ __launch_bounds__(256) void k_encode32(const Node4 *__restrict__ nodes, const Box6 *__restrict__ nodeBox, const Tri *__restrict__ tris, uint32_t nNodes,
                                                  const SceneConsts *__restrict__ consts, Node32 *__restrict__ out, int *__restrict__ leafKeys, SceneDev *__restrict__ scene)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float gLo[3], cell[3], invCell[3];
    uint32_t cexp[3];
    for (int k = 0; k < 3; ++k) {
        gLo[k] = consts->lo[k] - 2.0f * consts->pad; // (as gridOf on the host: same operations)
        cexp[k] = gridCellExponent((consts->hi[k] + 2.0f * consts->pad) - gLo[k]);
        cell[k] = __uint_as_float(cexp[k] << 23), invCell[k] = __uint_as_float((254u - cexp[k]) << 23);
    }
    if (i == 0 && scene)
        for (int k = 0; k < 3; ++k) scene->gridLo[k] = gLo[k], scene->gridCell[k] = cell[k], scene->gridCellExp[k] = cexp[k];
    if (i >= nNodes) return;
    const float pad = consts->pad;
    const Node4 nd = nodes[i];
    const uint32_t meta = __float_as_uint(nd.a.w);
    const int nInner = (int)((meta >> 24) & 7u), nValid = (int)(meta >> 27);
    const uint32_t innerBase = nd.c.z;
    const int leafKey = (int)nd.c.w;
    Box6 cb[4];
    Box6 nb;
    for (int k = 0; k < 3; ++k) nb.lo[k] = __builtin_inff(), nb.hi[k] = -__builtin_inff();
    for (int c = 0; c < nValid; ++c) {
        if (c < nInner) {
            cb[c] = nodeBox[innerBase + (uint32_t)c];
        } else {
            v3 bl, bh;
            triBounds(tris[~(leafKey + c)], bl, bh);
            cb[c].lo[0] = bl.x - pad, cb[c].lo[1] = bl.y - pad, cb[c].lo[2] = bl.z - pad;
            cb[c].hi[0] = bh.x + pad, cb[c].hi[1] = bh.y + pad, cb[c].hi[2] = bh.z + pad;
        }
        for (int k = 0; k < 3; ++k) nb.lo[k] = fmin_(nb.lo[k], cb[c].lo[k]), nb.hi[k] = fmax_(nb.hi[k], cb[c].hi[k]);
    }
    // frame: the grid point at or below the node's lower corner (the same float expression k_trace evaluates) and, per axis, the smallest
    // power-of-two number of cells from there that holds the node: hi <= origin + 255 * scale with scale = cell * 2^(e - 8)
    uint32_t g[3], e[3];
    float origin[3];
    for (int k = 0; k < 3; ++k) {
        const float f = floor_((nb.lo[k] - gLo[k]) * invCell[k]);
        g[k] = (uint32_t)fmin_(fmax_(f, 0.0f), 255.0f);
        origin[k] = __builtin_fmaf((float)g[k], cell[k], gLo[k]);
        if (origin[k] > nb.lo[k] && g[k] > 0u) g[k] -= 1u, origin[k] = __builtin_fmaf((float)g[k], cell[k], gLo[k]); // (the rounding of the sum)
        e[k] = 0u;
        while (e[k] < 15u && nb.hi[k] - origin[k] > 255.0f * __uint_as_float((cexp[k] + e[k] - 8u) << 23)) ++e[k];
    }
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        const float inv = __uint_as_float((254u - (cexp[k] + e[k] - 8u)) << 23); // 1 / scale
        for (int c = 0; c < 4; ++c) {
            uint32_t lo8 = 255u, hi8 = 0u; // no child: inverted planes, never entered
            if (c < nValid) {
                const float fl = floor_((cb[c].lo[k] - origin[k]) * inv);
                const float fh = __builtin_ceilf((cb[c].hi[k] - origin[k]) * inv);
                lo8 = (uint32_t)fmin_(fmax_(fl, 0.0f), 255.0f);
                hi8 = (uint32_t)fmin_(fmax_(fh, 0.0f), 255.0f);
            }
            qlo[k] |= lo8 << (8 * c);
            qhi[k] |= hi8 << (8 * c);
        }
    }
    Node32 o;
    o.p = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
    o.q = make_uint4(qhi[1], qhi[2], (innerBase & 0x01FFFFFFu) | ((uint32_t)nInner << 25) | (e[0] << 28), g[0] | (g[1] << 8) | (g[2] << 16) | (e[1] << 24) | (e[2] << 28));
    out[i] = o;
    leafKeys[i] = leafKey; // (the one word of Node4 k_trace still needs: a compact array that stays in L2)
}
void encodeNodes32(hipStream_t st, const BuildResult &tree, const SceneConsts *consts, SceneDev *scene)
{
    const uint32_t n = tree.nNodes > 0 ? (uint32_t)tree.nNodes : 0u;
    if (!tree.nodes32 && n) return;
    hipLaunchKernelGGL(k_encode32, dim3(n ? (n + 255) / 256 : 1), dim3(256), 0, st, tree.nodes, tree.nodeBox, tree.tris, n, consts, tree.nodes32, tree.leafKeys, scene);
}

void refitLBVH(hipStream_t st, const BuildResult &tree, uint32_t nTris, SceneConsts *consts)
{
    (void)nTris;
    for (int level = tree.levels - 1; level >= 0; --level) {
        const uint32_t a = tree.levelStart[level], b = tree.levelStart[level + 1];
        if (b > a) hipLaunchKernelGGL(k_refit4, dim3((b - a + 255) / 256), dim3(256), 0, st, tree.nodes, tree.nodeBox, tree.tris, a, b, consts);
    }
}

__global__ __launch_bounds__(256) void k_slot_of_prim(const Tri *__restrict__ leafTris, uint32_t n, uint32_t *__restrict__ slotOfPrim)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = __float_as_uint(leafTris[i].r.y);
    if (prim != 0xFFFFFFFFu) slotOfPrim[prim] = i; // (a slot a cached tree left unused carries prim id ~0)
}

__global__ __launch_bounds__(256) void k_area_sum(const Box6 *__restrict__ nodeBox, uint32_t n, SceneConsts *__restrict__ consts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    blockAreaAdd(i < n ? boxArea(nodeBox[i]) : 0.0f, consts);
}
void launchAreaSum(hipStream_t st, const Box6 *nodeBox, uint32_t n, SceneConsts *consts)
{
    if (n) hipLaunchKernelGGL(k_area_sum, dim3((n + 255) / 256), dim3(256), 0, st, nodeBox, n, consts);
}
// sum of the triangles' own areas, 0.5 |e1 x e2| (leaf-order array; unused slots of the 32-byte node formats carry prim id ~0)
__global__ __launch_bounds__(256) void k_tri_area_sum(const Tri *__restrict__ leafTris, uint32_t n, SceneConsts *__restrict__ consts)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float area = 0.0f;
    if (i < n) {
        const float4 p = leafTris[i].p, q = leafTris[i].q, r = leafTris[i].r;
        if (__float_as_uint(r.y) != 0xFFFFFFFFu) {
            const v3 e1(p.w, q.x, q.y), e2(q.z, q.w, r.x);
            const float a = 0.5f * length(cross(e1, e2));
            area = (a == a && a < 3.0e38f) ? a : 0.0f;
        }
    }
    blockAreaAdd(area, consts, true);
}
void launchTriAreaSum(hipStream_t st, const Tri *leafTris, uint32_t nSlots, SceneConsts *consts)
{
    if (nSlots) hipLaunchKernelGGL(k_tri_area_sum, dim3((nSlots + 255) / 256), dim3(256), 0, st, leafTris, nSlots, consts);
}

// ------------------------------------------------------------------------------------------- PLOC
// Parallel locally-ordered clustering (Meister & Bittner 2018) over the Morton-ordered triangle boxes: in every iteration each cluster
// looks for its nearest neighbour (smallest surface area of the merged box) among the r clusters before and behind it in the array,
// mutual nearest neighbours merge, the array is compacted.  The binary tree that results is what the radix tree of k_karras is for the
// LBVH: input of the same DP collapse to 4-wide nodes.  On meshes it is the better tree (terrain: -17 % summed node area, i.e. node
// visits per ray; tools/tree_proto.py, profiles/r4_tree_proto.txt); on the benchmark's uniform triangle fog spatial-median splits are
// within 7 % of a full-sweep SAH build and PLOC is 2-6 % WORSE than the LBVH — so buildLBVH builds both and keeps the cheaper one.
#ifndef HR_PLOC_RADIUS_MAX
#define HR_PLOC_RADIUS_MAX 32
#endif
HRD float unionArea(const Box6 &a, const Box6 &b)
{
    const float dx = fmax_(a.hi[0], b.hi[0]) - fmin_(a.lo[0], b.lo[0]), dy = fmax_(a.hi[1], b.hi[1]) - fmin_(a.lo[1], b.lo[1]),
                dz = fmax_(a.hi[2], b.hi[2]) - fmin_(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// state of the clustering on the device: [0] clusters in the array, [1] binary nodes made so far
__global__ __launch_bounds__(256) void k_ploc_nn(const Box6 *__restrict__ box, const uint32_t *__restrict__ state, int r, uint32_t *__restrict__ nn)
{
    __shared__ Box6 tile[256 + 2 * HR_PLOC_RADIUS_MAX];
    const int m = (int)state[0];
    const int base = (int)blockIdx.x * 256;
    if (base >= m) return;
    for (int t = threadIdx.x; t < 256 + 2 * r; t += 256) {
        const int g = base - r + t;
        if (g >= 0 && g < m) tile[t] = box[g];
    }
    __syncthreads();
    const int i = base + (int)threadIdx.x;
    if (i >= m) return;
    const Box6 me = tile[threadIdx.x + r];
    // the candidate with the smallest key (area, |i - j|, parity of min(i, j), min(i, j)) — a key that is the same seen from both ends
    // of a pair, so the pair with the globally smallest key is always mutual and every iteration merges at least one pair; among equal
    // areas (coincident or regularly spaced triangles) the nearest index wins and then the pair (2k, 2k + 1), so that a run of equal
    // boxes pairs up completely in one iteration instead of one pair at a time
    float best = __builtin_inff();
    int bestJ = -1;
    for (int off = -r; off <= r; ++off) {
        const int j = i + off;
        if (off == 0 || j < 0 || j >= m) continue;
        const float a = unionArea(me, tile[(int)threadIdx.x + r + off]);
        bool better = bestJ < 0 || a < best;
        if (!better && a == best) {
            const int d1 = off < 0 ? -off : off, d0 = bestJ > i ? bestJ - i : i - bestJ;
            const int lo1 = i < j ? i : j, lo0 = i < bestJ ? i : bestJ;
            better = d1 < d0 || (d1 == d0 && ((lo1 & 1) < (lo0 & 1) || ((lo1 & 1) == (lo0 & 1) && lo1 < lo0)));
        }
        if (better) best = a, bestJ = j;
    }
    nn[i] = (uint32_t)bestJ;
}

// PHASE 0: per workgroup, how many clusters stay in the array and how many pairs merge (packed: stay | merge << 32).
// PHASE 1: with those counts turned into exclusive prefixes, write the next array: a merging pair becomes a new binary node (its box,
// its collapse costs: both children exist since an earlier launch) at the place of its lower member; the higher member disappears.
template <int PHASE>
__global__ __launch_bounds__(256) void k_ploc_merge(const uint32_t *__restrict__ state, const uint32_t *__restrict__ nn, const int *__restrict__ refIn,
                                                    const Box6 *__restrict__ boxIn, unsigned long long *__restrict__ blockSums, int *__restrict__ refOut,
                                                    Box6 *__restrict__ boxOut, KNode *__restrict__ knodes, Box6 *__restrict__ nodeBox, float4 *__restrict__ cost)
{
    __shared__ uint32_t wk[4], wm[4];
    const uint32_t m = state[0];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= m) return; // (uniform)
    bool keep = false, makes = false;
    uint32_t j = 0;
    if (i < m) {
        j = nn[i];
        const bool mutual = j < m && nn[j] == i;
        keep = !(mutual && j < i);
        makes = mutual && i < j;
    }
    const unsigned long long kb = __ballot(keep), mb = __ballot(makes);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wk[wave] = (uint32_t)__popcll(kb), wm[wave] = (uint32_t)__popcll(mb);
    __syncthreads();
    if (PHASE == 0) {
        if (threadIdx.x == 0) blockSums[blockIdx.x] = (unsigned long long)(wk[0] + wk[1] + wk[2] + wk[3]) | ((unsigned long long)(wm[0] + wm[1] + wm[2] + wm[3]) << 32);
        return;
    }
    uint32_t kBefore = 0, mBefore = 0;
    for (uint32_t w = 0; w < wave; ++w) kBefore += wk[w], mBefore += wm[w];
    const unsigned long long pre = blockSums[blockIdx.x], lt = (1ull << lane) - 1ull;
    const uint32_t pos = (uint32_t)pre + kBefore + (uint32_t)__popcll(kb & lt);
    if (makes) {
        const uint32_t id = state[1] + (uint32_t)(pre >> 32) + mBefore + (uint32_t)__popcll(mb & lt);
        const Box6 a = boxIn[i], b = boxIn[j];
        Box6 u;
#pragma unroll
        for (int c = 0; c < 3; ++c) u.lo[c] = fmin_(a.lo[c], b.lo[c]), u.hi[c] = fmax_(a.hi[c], b.hi[c]);
        KNode k;
        k.left = refIn[i], k.right = refIn[j], k.first = 0, k.last = 0;
        knodes[id] = k;
        nodeBox[id] = u;
        cost[id] = collapseCost(k.left, k.right, boxArea(u), cost);
        refOut[pos] = (int)id;
        boxOut[pos] = u;
    } else if (keep) {
        refOut[pos] = refIn[i];
        boxOut[pos] = boxIn[i];
    }
}

// exclusive scan of n packed (low | high << 32) counters by ONE workgroup; then the iteration's totals update the state
__global__ __launch_bounds__(1024) void k_ploc_scan(unsigned long long *__restrict__ data, uint32_t *__restrict__ state)
{
    __shared__ unsigned long long waveSums[16];
    __shared__ unsigned long long carry;
    const uint32_t n = (state[0] + 255u) / 256u;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = (i < n) ? data[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o);
            if ((int)lane >= o) inc += t;
        }
        if (lane == 63) waveSums[wave] = inc;
        __syncthreads();
        unsigned long long wavePrefix = 0;
        for (uint32_t w = 0; w < wave; ++w) wavePrefix += waveSums[w];
        const unsigned long long c = carry;
        if (i < n) data[i] = c + wavePrefix + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wavePrefix + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        state[2] = (uint32_t)carry;         // clusters of the next array
        state[3] = (uint32_t)(carry >> 32); // nodes made by this iteration
    }
}
__global__ void k_ploc_advance(uint32_t *state)
{
    state[0] = state[2];
    state[1] += state[3];
}
__global__ __launch_bounds__(256) void k_ploc_init(const Box6 *__restrict__ leafBox, uint32_t n, int *__restrict__ ref, Box6 *__restrict__ box)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ref[i] = ~(int)i;
    box[i] = leafBox[i];
}

// The last iterations in ONE workgroup: once at most kPlocTail clusters are left they live in LDS and the remaining ~20 iterations (each
// of which was five launches and a 4-byte read-back) run without leaving the kernel.  Same nearest-neighbour rule, same pair rule, same
// order of the new nodes (array order) as the iterations above, so the tree does not depend on where the hand-over happens.
static const int kPlocTail = 1024;
__global__ __launch_bounds__(kPlocTail) void k_ploc_tail(uint32_t *__restrict__ state, const int *__restrict__ refIn, const Box6 *__restrict__ boxIn, int r,
                                                        KNode *__restrict__ knodes, Box6 *__restrict__ nodeBox, float4 *cost /* read back inside the launch: no restrict */)
{
    __shared__ Box6 sbox[2][kPlocTail];
    __shared__ int sref[2][kPlocTail];
    __shared__ int snn[kPlocTail];
    __shared__ uint32_t wk[kPlocTail / 64], wm[kPlocTail / 64];
    const int t = (int)threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    int m = (int)state[0];
    uint32_t base = state[1];
    if (t < m) sbox[0][t] = boxIn[t], sref[0][t] = refIn[t];
    __syncthreads();
    int cur = 0;
    while (m > 1) {
        // nearest neighbour (k_ploc_nn's rule)
        int bestJ = -1;
        if (t < m) {
            const Box6 me = sbox[cur][t];
            float best = __builtin_inff();
            for (int off = -r; off <= r; ++off) {
                const int j = t + off;
                if (off == 0 || j < 0 || j >= m) continue;
                const float a = unionArea(me, sbox[cur][j]);
                bool better = bestJ < 0 || a < best;
                if (!better && a == best) {
                    const int d1 = off < 0 ? -off : off, d0 = bestJ > t ? bestJ - t : t - bestJ;
                    const int lo1 = t < j ? t : j, lo0 = t < bestJ ? t : bestJ;
                    better = d1 < d0 || (d1 == d0 && ((lo1 & 1) < (lo0 & 1) || ((lo1 & 1) == (lo0 & 1) && lo1 < lo0)));
                }
                if (better) best = a, bestJ = j;
            }
            snn[t] = bestJ;
        }
        __syncthreads();
        bool keep = false, makes = false;
        int j = 0;
        if (t < m) {
            j = snn[t];
            const bool mutual = j >= 0 && j < m && snn[j] == t;
            keep = !(mutual && j < t);
            makes = mutual && t < j;
        }
        const unsigned long long kb = __ballot(keep), mb = __ballot(makes);
        if (lane == 0) wk[wave] = (uint32_t)__popcll(kb), wm[wave] = (uint32_t)__popcll(mb);
        __syncthreads();
        uint32_t kBefore = 0, mBefore = 0, kTotal = 0, mTotal = 0;
        for (uint32_t w = 0; w < (uint32_t)(kPlocTail / 64); ++w) {
            kBefore += w < wave ? wk[w] : 0u, mBefore += w < wave ? wm[w] : 0u;
            kTotal += wk[w], mTotal += wm[w];
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t pos = kBefore + (uint32_t)__popcll(kb & lt);
        if (makes) {
            const uint32_t id = base + mBefore + (uint32_t)__popcll(mb & lt);
            const Box6 a = sbox[cur][t], b = sbox[cur][j];
            Box6 u;
#pragma unroll
            for (int c = 0; c < 3; ++c) u.lo[c] = fmin_(a.lo[c], b.lo[c]), u.hi[c] = fmax_(a.hi[c], b.hi[c]);
            KNode k;
            k.left = sref[cur][t], k.right = sref[cur][j], k.first = 0, k.last = 0;
            knodes[id] = k;
            nodeBox[id] = u;
            // (a child made earlier in THIS launch was written by another thread: visible after the barrier below and the fence)
            cost[id] = collapseCost(k.left, k.right, boxArea(u), cost);
            sref[cur ^ 1][pos] = (int)id;
            sbox[cur ^ 1][pos] = u;
        } else if (keep) {
            sref[cur ^ 1][pos] = sref[cur][t];
            sbox[cur ^ 1][pos] = sbox[cur][t];
        }
        __threadfence();
        __syncthreads();
        m = (int)kTotal, base += mTotal, cur ^= 1;
        // An iteration that merged nothing would repeat itself for ever, one workgroup spinning until the GPU is reset.  The pair key
        // reads the same from both ends, so it cannot happen for ordered areas — but NaN boxes (non-finite vertices) are not ordered.
        // mTotal comes from LDS totals every thread has read alike: a uniform exit; state[1] != n - 1 then tells buildPLOC (rc 7:
        // the radix tree is kept).
        if (mTotal == 0u) break;
    }
    if (t == 0) state[0] = (uint32_t)m, state[1] = base;
}

// Binary tree over the sorted triangles by PLOC: knodes / nodeBox / cost have n - 1 entries; *rootOut is the root's index (n - 2).
// Returns 0, or non-zero when it did not finish (the caller then keeps the radix tree).
static int buildPLOC(hipStream_t st, const Box6 *leafBox, uint32_t n, int radius, KNode *knodes, Box6 *nodeBox, float4 *cost, int *rootOut)
{
    int *ref[2] = {nullptr, nullptr};
    Box6 *box[2] = {nullptr, nullptr};
    uint32_t *nn = nullptr, *state = nullptr;
    unsigned long long *sums = nullptr;
    const uint32_t nb0 = (n + 255) / 256;
    bool ok = hipMalloc(&ref[0], 4ull * n) == hipSuccess && hipMalloc(&ref[1], 4ull * n) == hipSuccess && hipMalloc(&box[0], sizeof(Box6) * (size_t)n) == hipSuccess &&
              hipMalloc(&box[1], sizeof(Box6) * (size_t)n) == hipSuccess && hipMalloc(&nn, 4ull * n) == hipSuccess && hipMalloc(&state, 16) == hipSuccess &&
              hipMalloc(&sums, 8ull * nb0) == hipSuccess;
    const uint32_t init[4] = {n, 0u, 0u, 0u};
    ok = ok && hipMemcpyAsync(state, init, 16, hipMemcpyHostToDevice, st) == hipSuccess;
    int rc = ok ? 0 : 1;
    if (ok) {
        hipLaunchKernelGGL(k_ploc_init, dim3(nb0), dim3(256), 0, st, leafBox, n, ref[0], box[0]);
        const int r = radius < 1 ? 1 : (radius > HR_PLOC_RADIUS_MAX ? HR_PLOC_RADIUS_MAX : radius);
        uint32_t m = n;
        int cur = 0;
        // every iteration merges at least one pair (k_ploc_nn), typically a third of the clusters: ~45 iterations for a million
        // triangles.  The host reads the cluster count back after each (a 4-byte copy: the launches are sized by it).
        for (uint32_t it = 0; m > (uint32_t)kPlocTail; ++it) {
            if (it >= 4096u) { // (a pathological input that merges a pair at a time: not worth it, the radix tree is kept)
                rc = 7;
                break;
            }
            const uint32_t nb = (m + 255) / 256;
            hipLaunchKernelGGL(k_ploc_nn, dim3(nb), dim3(256), 0, st, box[cur], state, r, nn);
            hipLaunchKernelGGL((k_ploc_merge<0>), dim3(nb), dim3(256), 0, st, state, nn, ref[cur], box[cur], sums, ref[cur ^ 1], box[cur ^ 1], knodes, nodeBox, cost);
            hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, st, sums, state);
            hipLaunchKernelGGL((k_ploc_merge<1>), dim3(nb), dim3(256), 0, st, state, nn, ref[cur], box[cur], sums, ref[cur ^ 1], box[cur ^ 1], knodes, nodeBox, cost);
            hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, st, state);
            uint32_t mNew = 0;
            if (hipMemcpyAsync(&mNew, state, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rc = 1;
                break;
            }
            if (mNew >= m || mNew == 0) { // (cannot happen: see k_ploc_nn)
                rc = 7;
                break;
            }
            m = mNew;
            cur ^= 1;
        }
        if (rc == 0 && m > 1) hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kPlocTail), 0, st, state, ref[cur], box[cur], r, knodes, nodeBox, cost);
        uint32_t made = 0;
        if (rc == 0 && (hipMemcpyAsync(&made, state + 1, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) rc = 1;
        if (rc == 0 && made != n - 1) rc = 7;
        *rootOut = (int)n - 2;
    }
    hipFree(ref[0]), hipFree(ref[1]), hipFree(box[0]), hipFree(box[1]), hipFree(nn), hipFree(state), hipFree(sums);
    return rc;
}

int buildLBVH(hipStream_t st, const Tri *trisPrim, uint32_t n, const float lo[3], const float hi[3], float pad, const SceneConsts *dConsts,
              BuildResult *out, const BuildOptions &opt)
{
    out->nodes = nullptr, out->nodes32 = nullptr, out->leafKeys = nullptr, out->tris = nullptr, out->nodeBox = nullptr, out->slotOfPrim = nullptr;
    out->nNodes = 0, out->rootLeafCount = 0, out->levels = 0, out->triSlots = 0;
    out->builder = 0, out->costRadix = out->costPloc = 0.0f;
    for (uint32_t &v : out->levelStart) v = 0;
    if (n == 0) return 0;
    if (n >= (1u << 28)) return 2;
    const uint32_t nBlocks = (n + kSortTile - 1) / kSortTile;
    uint32_t *keysA = nullptr, *keysB = nullptr, *valsA = nullptr, *valsB = nullptr, *blockHist = nullptr, *stamp = nullptr, *total = nullptr;
    Tri *finalTris = nullptr;
    KNode *knodes = nullptr;
    Box6 *leafBox = nullptr, *nodeBox = nullptr;
    Tri *sorted = nullptr;
    float4 *dpCost = nullptr; // costs of the collapse (k_refit_round, k_collapse4)
    HR_CHECK(hipMalloc(&keysA, 4ull * n));
    HR_CHECK(hipMalloc(&keysB, 4ull * n));
    HR_CHECK(hipMalloc(&valsA, 4ull * n));
    HR_CHECK(hipMalloc(&valsB, 4ull * n));
    HR_CHECK(hipMalloc(&blockHist, 4ull * 256 * nBlocks));
    HR_CHECK(hipMalloc(&sorted, sizeof(Tri) * (size_t)n));
    HR_CHECK(hipMalloc(&leafBox, sizeof(Box6) * (size_t)n));
    HR_CHECK(hipMalloc(&total, 8)); // [0] nodes appended, [1] triangles placed
    const v3 vlo(lo[0], lo[1], lo[2]);
    const v3 ext(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    const uint32_t g256 = (n + 255) / 256;
    hipLaunchKernelGGL(k_morton, dim3(g256), dim3(256), 0, st, trisPrim, n, vlo, ext, keysA, valsA);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(k_sort_hist, dim3(nBlocks), dim3(64), 0, st, keysA, n, shift, blockHist, nBlocks);
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, blockHist, 256u * nBlocks, (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nBlocks), dim3(64), 0, st, keysA, valsA, n, shift, blockHist, nBlocks, keysB, valsB);
        std::swap(keysA, keysB);
        std::swap(valsA, valsB);
    }
    hipLaunchKernelGGL(k_gather_tris, dim3(g256), dim3(256), 0, st, trisPrim, valsA, n, sorted, pad, leafBox);
    int rc = 0;
    if (n <= 1u) { // a single triangle: the root is a leaf
        out->rootLeafCount = (int)n;
    } else {
        const int nInternal = (int)n - 1;
        const uint32_t gi = (nInternal + 255) / 256;
        HR_CHECK(hipMalloc(&knodes, sizeof(KNode) * (size_t)nInternal));
        HR_CHECK(hipMalloc(&nodeBox, sizeof(Box6) * (size_t)nInternal));
        HR_CHECK(hipMalloc(&stamp, 4ull * nInternal));
        HR_CHECK(hipMemsetAsync(stamp, 0, 4ull * nInternal, st));
#if HR_COLLAPSE_DP
        HR_CHECK(hipMalloc(&dpCost, sizeof(float4) * (size_t)nInternal));
#endif
        hipLaunchKernelGGL(k_karras, dim3(gi), dim3(256), 0, st, keysA, (int)n, knodes);
        // tree height <= 30 key bits + 28 index bits; check the root every 16 rounds
        uint32_t rootStamp = 0;
        for (uint32_t round = 1; round <= 64 && rootStamp == 0; ++round) {
            hipLaunchKernelGGL(k_refit_round, dim3(gi), dim3(256), 0, st, knodes, nInternal, leafBox, nodeBox, stamp, round, dpCost);
            if ((round & 15u) == 0u) {
                HR_CHECK(hipMemcpyAsync(&rootStamp, stamp, 4, hipMemcpyDeviceToHost, st));
                HR_CHECK(hipStreamSynchronize(st));
            }
        }
        if (rootStamp == 0) {
            HR_CHECK(hipMemcpyAsync(&rootStamp, stamp, 4, hipMemcpyDeviceToHost, st));
            HR_CHECK(hipStreamSynchronize(st));
        }
        if (rootStamp == 0) rc = 3;
#if HR_ROTATE_SWEEPS
        // tree rotations, bottom-up by stamp; between sweeps the refit rounds renew boxes and stamps (the heights have changed)
        for (int sweep = 0; sweep < HR_ROTATE_SWEEPS && rc == 0; ++sweep) {
            HR_CHECK(hipMemsetAsync(total, 0, 4, st));
            for (uint32_t sv = 2; sv <= rootStamp; ++sv)
                hipLaunchKernelGGL(k_rotate, dim3(gi), dim3(256), 0, st, knodes, nInternal, leafBox, nodeBox, stamp, sv, total);
            HR_CHECK(hipMemsetAsync(stamp, 0, 4ull * nInternal, st));
            rootStamp = 0;
            for (uint32_t round = 1; round <= 96 && rootStamp == 0; ++round) {
                hipLaunchKernelGGL(k_refit_round, dim3(gi), dim3(256), 0, st, knodes, nInternal, leafBox, nodeBox, stamp, round, dpCost);
                if ((round & 15u) == 0u) {
                    HR_CHECK(hipMemcpyAsync(&rootStamp, stamp, 4, hipMemcpyDeviceToHost, st));
                    HR_CHECK(hipStreamSynchronize(st));
                }
            }
            if (rootStamp == 0) {
                HR_CHECK(hipMemcpyAsync(&rootStamp, stamp, 4, hipMemcpyDeviceToHost, st));
                HR_CHECK(hipStreamSynchronize(st));
            }
            if (rootStamp == 0) rc = 3;
        }
#endif
        // ---- the second candidate: the PLOC tree over the same sorted triangles; the one whose collapse costs less (summed surface area
        // of the 4-wide nodes = expected node visits of a random ray) is kept
        KNode *pKnodes = nullptr;
        Box6 *pBox = nullptr;
        float4 *pCost = nullptr;
        int pRoot = 0;
        bool usePloc = false;
        out->costRadix = out->costPloc = 0.0f;
        if (rc == 0 && dpCost && opt.ploc != 0 && n >= 4u) {
            bool ok = hipMalloc(&pKnodes, sizeof(KNode) * (size_t)nInternal) == hipSuccess && hipMalloc(&pBox, sizeof(Box6) * (size_t)nInternal) == hipSuccess &&
                      hipMalloc(&pCost, sizeof(float4) * (size_t)nInternal) == hipSuccess;
            ok = ok && buildPLOC(st, leafBox, n, opt.plocRadius, pKnodes, pBox, pCost, &pRoot) == 0;
            float4 cR{}, cP{};
            Box6 rootBox{};
            ok = ok && hipMemcpyAsync(&cR, dpCost, sizeof(float4), hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipMemcpyAsync(&cP, pCost + pRoot, sizeof(float4), hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipMemcpyAsync(&rootBox, nodeBox, sizeof(Box6), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            if (ok) {
                const float dx = rootBox.hi[0] - rootBox.lo[0], dy = rootBox.hi[1] - rootBox.lo[1], dz = rootBox.hi[2] - rootBox.lo[2];
                const float ra = dx * dy + dy * dz + dz * dx;
                out->costRadix = ra > 0.0f ? cR.x / ra : 0.0f, out->costPloc = ra > 0.0f ? cP.x / ra : 0.0f;
                // PLOC has to win clearly (5 %): the surface-area measure assumes rays distributed like random lines through the root box,
                // while real rays concentrate where the geometry is — the benchmark soup inside a room (c3d) rates PLOC 2.3 % cheaper
                // because of the room's 18 large triangles and traces 4.3 % SLOWER with it (the soup itself suits the radix tree's regular
                // cells: PLOC 4.9 % dearer, 3 % slower); a mesh gains far more than the margin (terrain: 26.5 % cheaper, 18 % fewer node
                // visits per ray, +5.5 % Mrays/s) — profiles/r4j_ploc_ab.txt, r4k_tree_costs.txt
                usePloc = opt.ploc >= 2 || cP.x < 0.95f * cR.x;
            }
            (void)hipGetLastError();
        }
        const uint32_t nMax = (uint32_t)nInternal; // every 4-wide node stands for one binary inner node
        HR_CHECK(hipMalloc(&finalTris, sizeof(Tri) * (size_t)n));
        int *binOf = nullptr;
        HR_CHECK(hipMalloc(&out->nodes, sizeof(Node4) * (size_t)nMax));
        HR_CHECK(hipMalloc(&out->nodes32, sizeof(Node32) * (size_t)nMax));
        HR_CHECK(hipMalloc(&out->leafKeys, sizeof(int) * (size_t)nMax));
        HR_CHECK(hipMalloc(&out->nodeBox, sizeof(Box6) * (size_t)nMax));
        HR_CHECK(hipMalloc(&binOf, 4ull * nMax));
        uint32_t levelStart = 0, levelEnd = 1;
        for (int attempt = 0; attempt < 2 && rc == 0; ++attempt) {
            const KNode *bk = usePloc ? pKnodes : knodes;
            const Box6 *bb = usePloc ? pBox : nodeBox;
            const float4 *bc = usePloc ? pCost : dpCost;
            const int rootBin = usePloc ? pRoot : 0;
            const uint32_t init[2] = {1u, 0u};
            HR_CHECK(hipMemcpyAsync(binOf, &rootBin, 4, hipMemcpyHostToDevice, st));
            HR_CHECK(hipMemcpyAsync(total, init, 8, hipMemcpyHostToDevice, st));
            levelStart = 0, levelEnd = 1;
            out->levels = 0;
            // (the radix tree is at most 58 levels deep — its key length; a PLOC tree has no such bound, so it is only kept when its collapse
            // fits the traversal stack; otherwise the second attempt collapses the radix tree)
            const int levelCap = usePloc ? (opt.maxLevels < 64 ? opt.maxLevels : 64) : 64;
            for (int level = 0; level < levelCap && levelEnd > levelStart; ++level) {
                const uint32_t cnt = levelEnd - levelStart;
                hipLaunchKernelGGL(k_collapse4, dim3((cnt + 255) / 256), dim3(256), 0, st, bk, leafBox, bb, binOf, levelStart, levelEnd, total,
                                   total + 1, sorted, finalTris, out->nodes, out->nodeBox, dConsts, bc);
                uint32_t newEnd = 0;
                HR_CHECK(hipMemcpyAsync(&newEnd, total, 4, hipMemcpyDeviceToHost, st));
                HR_CHECK(hipStreamSynchronize(st));
                if (newEnd > nMax) {
                    rc = 4;
                    break;
                }
                out->levelStart[level] = levelStart, out->levelStart[level + 1] = levelEnd;
                levelStart = levelEnd;
                levelEnd = newEnd;
                out->levels = level + 1;
            }
            if (rc == 0 && levelEnd > levelStart && usePloc) { // too deep for the stack: fall back to the radix tree
                usePloc = false;
                continue;
            }
            break;
        }
        out->builder = usePloc ? 1 : 0;
        hipFree(binOf);
        hipFree(pKnodes), hipFree(pBox), hipFree(pCost);
        if (levelEnd > levelStart && rc == 0) rc = 6; // deeper than the key length allows: cannot happen with n < 2^28
        out->nNodes = (int)levelEnd;
        uint32_t placed = 0;
        HR_CHECK(hipMemcpyAsync(&placed, total + 1, 4, hipMemcpyDeviceToHost, st));
        HR_CHECK(hipStreamSynchronize(st));
        if (placed != n && rc == 0) rc = 5; // every triangle is the leaf of exactly one node
    }
    out->tris = finalTris ? finalTris : sorted;
    out->triSlots = n;
    HR_CHECK(hipMalloc(&out->slotOfPrim, 4ull * n));
    hipLaunchKernelGGL(k_slot_of_prim, dim3((out->triSlots + 255) / 256), dim3(256), 0, st, out->tris, out->triSlots, out->slotOfPrim);
    HR_CHECK(hipStreamSynchronize(st));
    hipFree(keysA), hipFree(keysB), hipFree(valsA), hipFree(valsB), hipFree(blockHist), hipFree(leafBox), hipFree(total);
    if (knodes) hipFree(knodes);
    if (nodeBox) hipFree(nodeBox);
    if (stamp) hipFree(stamp);
    if (dpCost) hipFree(dpCost);
    if (finalTris) hipFree(sorted);
    if (hipGetLastError() != hipSuccess) return 1;
    return rc;
}

// ------------------------------------------------------------------------------------ content hash
// Order-independent 64-bit digest of a device buffer (words mixed with their index, then summed): the key of the tree cache.
__global__ __launch_bounds__(256) void k_hash_words(const uint32_t *__restrict__ words, size_t n, unsigned long long seed,
                                                    unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = ((unsigned long long)words[i] << 32 | (unsigned long long)(i & 0xFFFFFFFFull)) + seed + (i >> 32) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; // splitmix64 finaliser
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        acc += z ^ (z >> 31);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
void launchHashWords(hipStream_t st, const void *words, size_t nWords, unsigned long long seed, unsigned long long *out)
{
    if (!nWords) return;
    size_t blocks = (nWords + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_hash_words, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(words), nWords, seed, out);
}

// ------------------------------------------------------------------ environment importance table (HR_ESTIMATOR_ENV_MIS)
// The distribution the one-sample MIS estimator draws environment directions from (include/hrcore.h): texel weight =
// (luminosity of the brightest texel of its 3 x 3 neighbourhood + maxLuminosity / 65536) x cos(elevation of the row), quantised to
// integers so that the sums do not depend on their order — the tables are bit-identical to oracle/oracle_scene.cpp::buildEnvTable.
__global__ __launch_bounds__(256) void k_env_lum(TexDesc t, float *__restrict__ lum, uint32_t *__restrict__ maxBits)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float l = 0.0f;
    if (i < (uint32_t)(t.w * t.h)) {
        const v4 c = texel(t, (int)(i % (uint32_t)t.w), (int)(i / (uint32_t)t.w));
        l = (c.x * 0.33f + c.y * 0.59f) + c.z * 0.11f; // utility.rlsl:163-166 luminosity
        l = l > 0.0f ? l : 0.0f;
        l = l < 1e30f ? l : 1e30f; // (an infinite texel must not turn the normalisation into Inf / Inf)
        lum[i] = l;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l = fmax_(l, __shfl_xor(l, o));
    if ((threadIdx.x & 63) == 0 && l > 0.0f) atomicMax(maxBits, __float_as_uint(l)); // non-negative floats order like their bits
}
__global__ __launch_bounds__(256) void k_env_dilate(const float *__restrict__ lum, float *__restrict__ dil, int w, int h)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint32_t)(w * h)) return;
    const int i = (int)(idx % (uint32_t)w), j = (int)(idx / (uint32_t)w);
    float m = 0.0f;
    for (int dj = -1; dj <= 1; ++dj) {
        const int jj = j + dj < 0 ? 0 : (j + dj >= h ? h - 1 : j + dj);
        for (int di = -1; di <= 1; ++di) {
            const int ii = (i + di + w) % w;
            const float l = lum[(size_t)jj * w + ii];
            m = l > m ? l : m;
        }
    }
    dil[idx] = m;
}
HRD uint32_t envWeight(float lum, float floorLum, float norm, float c)
{
    const float wt = (lum + floorLum) * c;
    const float q = norm > 0.0f ? (wt / norm) * 1048576.0f : 0.0f;
    return (uint32_t)q + 1u;
}
// one workgroup per row: quantised weights and their sum
__global__ __launch_bounds__(256) void k_env_rows(const float *__restrict__ dil, int w, int h, const uint32_t *__restrict__ maxBits,
                                                  uint32_t *__restrict__ wq, unsigned long long *__restrict__ rowSum)
{
    __shared__ unsigned long long part[4];
    const int j = blockIdx.x;
    const float maxLum = __uint_as_float(*maxBits);
    const float floorLum = maxLum * (1.0f / 65536.0f), norm = maxLum + floorLum;
    const float elevation = (((float)j + 0.5f) / (float)h - 0.5f) * HR_KPI;
    const float c = cos_(elevation);
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < w; i += 256) {
        const uint32_t v = envWeight(dil[(size_t)j * w + i], floorLum, norm, c);
        wq[(size_t)j * w + i] = v;
        s += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowSum[j] = part[0] + part[1] + part[2] + part[3];
}
// marginal over the rows (h is at most a few thousand: one thread)
__global__ void k_env_marginal(const unsigned long long *__restrict__ rowSum, int w, int h, const uint32_t *__restrict__ maxBits,
                               float *__restrict__ rowCdf, unsigned long long *__restrict__ total, float *__restrict__ meanLum)
{
    unsigned long long t = 0;
    for (int j = 0; j < h; ++j) t += rowSum[j];
    unsigned long long acc = 0;
    for (int j = 0; j < h; ++j) {
        rowCdf[j] = (float)acc / (float)t;
        acc += rowSum[j];
    }
    rowCdf[h] = 1.0f;
    *total = t;
    // mean luminosity over the sphere from the same integers: sum(weight) / sum(cos), rows in order
    const float maxLum = __uint_as_float(*maxBits);
    const float norm = maxLum + maxLum * (1.0f / 65536.0f);
    float sumC = 0.0f;
    for (int j = 0; j < h; ++j) sumC = sumC + cos_((((float)j + 0.5f) / (float)h - 0.5f) * HR_KPI);
    *meanLum = (((float)t / 1048576.0f) * norm) / ((float)w * sumC);
}
// one workgroup per row: conditional CDF over the columns (chunked scan with a carry) and the texel probabilities
// The probability of a texel is what sampleEnv() really draws it with: the float CDF steps the two searches invert,
// (rowCdf[j + 1] - rowCdf[j]) x (colCdf[i + 1] - colCdf[i]) — not the exact ratio of the integer weights, which a float CDF of a map
// with a 2^20 : 1 weight range cannot resolve for dim texels (the balance heuristic needs the density samples are drawn with).
__global__ __launch_bounds__(256) void k_env_cols(const uint32_t *__restrict__ wq, const unsigned long long *__restrict__ rowSum,
                                                  const float *__restrict__ rowCdf, int w, float *__restrict__ colCdf,
                                                  float *__restrict__ prob)
{
    __shared__ unsigned long long waveSum[4];
    __shared__ unsigned long long carry;
    const int j = blockIdx.x;
    const float fr = (float)rowSum[j], pRow = rowCdf[j + 1] - rowCdf[j];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < w; base += 256) {
        const int i = base + (int)threadIdx.x;
        const unsigned long long v = i < w ? (unsigned long long)wq[(size_t)j * w + i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o);
            if ((int)lane >= o) inc += up;
        }
        if (lane == 63) waveSum[wave] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t k = 0; k < wave; ++k) before += waveSum[k];
        if (i < w) {
            const float cThis = (float)(before + inc - v) / fr, cNext = (i == w - 1) ? 1.0f : (float)(before + inc) / fr;
            colCdf[(size_t)j * (w + 1) + i] = cThis;
            prob[(size_t)j * w + i] = pRow * (cNext - cThis);
        }
        __syncthreads();
        if (threadIdx.x == 255) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) colCdf[(size_t)j * (w + 1) + w] = 1.0f;
}

// scratch: lum, dil (w*h floats each), wq (w*h uint32), rowSum (h uint64), total (uint64), maxBits (uint32) — caller-owned
void launchEnvTable(hipStream_t st, const TexDesc &tex, float *lum, float *dil, uint32_t *wq, unsigned long long *rowSum, unsigned long long *total,
                    uint32_t *maxBits, float *rowCdf, float *colCdf, float *prob, float *meanLum)
{
    const int w = tex.w, h = tex.h;
    const uint32_t n = (uint32_t)((size_t)w * (size_t)h), g = (n + 255) / 256;
    hipMemsetAsync(maxBits, 0, 4, st);
    hipLaunchKernelGGL(k_env_lum, dim3(g), dim3(256), 0, st, tex, lum, maxBits);
    hipLaunchKernelGGL(k_env_dilate, dim3(g), dim3(256), 0, st, lum, dil, w, h);
    hipLaunchKernelGGL(k_env_rows, dim3(h), dim3(256), 0, st, dil, w, h, maxBits, wq, rowSum);
    hipLaunchKernelGGL(k_env_marginal, dim3(1), dim3(1), 0, st, rowSum, w, h, maxBits, rowCdf, total, meanLum);
    hipLaunchKernelGGL(k_env_cols, dim3(h), dim3(256), 0, st, wq, rowSum, rowCdf, w, colCdf, prob);
}

// Guide tables of the two inverse-CDF searches (hr_shade.h::sampleEnv): guide[k] = the largest index whose CDF value is <= k / K.
// A search for x starts in [guide[floor(x K)], guide[floor(x K) + 1]] and returns what the search over the whole table returns
// (the oracle's plain binary search), after ~2 probes instead of 10 + 11 dependent loads.
HRD int cdfSearch(const float *cdf, int lo, int hi, float x)
{
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cdf[mid] <= x)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(64) void k_env_guides(const float *__restrict__ rowCdf, const float *__restrict__ colCdf, int w, int h,
                                                   uint16_t *__restrict__ rowGuide, uint16_t *__restrict__ colGuide)
{
    // block 0: the row guide (kEnvRowGuide + 1 entries); block j + 1: the column guide of row j (kEnvColGuide + 1 entries)
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k <= kEnvRowGuide; k += 64)
            rowGuide[k] = (uint16_t)cdfSearch(rowCdf, 0, h - 1, (float)k * (1.0f / (float)kEnvRowGuide));
    } else {
        const int j = (int)blockIdx.x - 1;
        const float *cc = colCdf + (size_t)j * (w + 1);
        for (int k = threadIdx.x; k <= kEnvColGuide; k += 64)
            colGuide[(size_t)j * (kEnvColGuide + 1) + k] = (uint16_t)cdfSearch(cc, 0, w - 1, (float)k * (1.0f / (float)kEnvColGuide));
    }
}
void launchEnvGuides(hipStream_t st, const float *rowCdf, const float *colCdf, int w, int h, uint16_t *rowGuide, uint16_t *colGuide)
{
    hipLaunchKernelGGL(k_env_guides, dim3(h + 1), dim3(64), 0, st, rowCdf, colCdf, w, h, rowGuide, colGuide);
}

// ---------------------------------------------------------------------------- mip chain, texel density (HR_TEXTURE_LOD_CONE)
// One level of the chain: dst(x, y) = ((s(2x, 2y) + s(2x+1, 2y)) + (s(2x, 2y+1) + s(2x+1, 2y+1))) * 0.25 per channel, source
// coordinates clamped to the source level (odd sizes); u8 data enters as float(byte) / 255.0f, levels >= 1 are f32.
__global__ __launch_bounds__(256) void k_mip_down(const void *__restrict__ src, int srcIsU8, int sw, int sh, int c, float *__restrict__ dst, int dw, int dh)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint32_t)(dw * dh)) return;
    const int x = (int)(i % (uint32_t)dw), y = (int)(i / (uint32_t)dw);
    const int x0 = 2 * x < sw ? 2 * x : sw - 1, x1 = 2 * x + 1 < sw ? 2 * x + 1 : sw - 1;
    const int y0 = 2 * y < sh ? 2 * y : sh - 1, y1 = 2 * y + 1 < sh ? 2 * y + 1 : sh - 1;
    for (int k = 0; k < c; ++k) {
        float a, b, cc, d;
        if (srcIsU8) {
            const uint8_t *p = reinterpret_cast<const uint8_t *>(src);
            a = (float)p[((size_t)y0 * sw + x0) * c + k] / 255.0f, b = (float)p[((size_t)y0 * sw + x1) * c + k] / 255.0f;
            cc = (float)p[((size_t)y1 * sw + x0) * c + k] / 255.0f, d = (float)p[((size_t)y1 * sw + x1) * c + k] / 255.0f;
        } else {
            const float *p = reinterpret_cast<const float *>(src);
            a = p[((size_t)y0 * sw + x0) * c + k], b = p[((size_t)y0 * sw + x1) * c + k];
            cc = p[((size_t)y1 * sw + x0) * c + k], d = p[((size_t)y1 * sw + x1) * c + k];
        }
        dst[((size_t)y * dw + x) * c + k] = ((a + b) + (cc + d)) * 0.25f;
    }
}

// levels 1 .. nLevels-1 of `t` into `mips` (laid out as hr_texture.h's mipOffset expects)
void launchMipChain(hipStream_t st, const TexDesc &t, int nLevels, float *mips)
{
    const void *src = t.px;
    int srcU8 = t.dtype == HR_TEX_U8 ? 1 : 0, sw = t.w, sh = t.h;
    size_t off = 0;
    for (int l = 1; l < nLevels; ++l) {
        const int dw = sw / 2 < 1 ? 1 : sw / 2, dh = sh / 2 < 1 ? 1 : sh / 2;
        float *dst = mips + off;
        hipLaunchKernelGGL(k_mip_down, dim3(((uint32_t)(dw * dh) + 255) / 256), dim3(256), 0, st, src, srcU8, sw, sh, t.c, dst, dw, dh);
        off += (size_t)dw * dh * t.c;
        src = dst, srcU8 = 0, sw = dw, sh = dh;
    }
}

__global__ void k_tex_lod_scale(TexDesc *table, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) table[i].lodScale = 0.5f * (log_((float)table[i].w * (float)table[i].h) * 1.4426950408889634f);
}
void launchTexLodScale(hipStream_t st, TexDesc *table, int n)
{
    if (n > 0) hipLaunchKernelGGL(k_tex_lod_scale, dim3((n + 63) / 64), dim3(64), 0, st, table, n);
}

// texDensity[prim] = 0.5 * log2(uv area / world area) of every triangle (twice-areas cancel); -1e30 where either area is zero
// (no uvs, degenerate triangle): such a triangle is textured at level 0
__global__ __launch_bounds__(256) void k_tex_density(const Tri *__restrict__ leafTris, uint32_t nSlots, const TriAttr *__restrict__ attrs,
                                                     float *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nSlots) return;
    const Tri tr = leafTris[i];
    const uint32_t prim = __float_as_uint(tr.r.y);
    if (prim == 0xFFFFFFFFu) return;
    const v3 e1(tr.p.w, tr.q.x, tr.q.y), e2(tr.q.z, tr.q.w, tr.r.x);
    const float world2 = length(cross(e1, e2));
    const TriAttr &a = attrs[prim];
    const float du1 = a.uv[2] - a.uv[0], dv1 = a.uv[3] - a.uv[1], du2 = a.uv[4] - a.uv[0], dv2 = a.uv[5] - a.uv[1];
    const float uv2 = abs_(du1 * dv2 - dv1 * du2);
    out[prim] = (world2 > 0.0f && uv2 > 0.0f) ? 0.5f * (log_(uv2 / world2) * 1.4426950408889634f) : -1e30f;
}
void launchTexDensity(hipStream_t st, const Tri *leafTris, uint32_t nSlots, const TriAttr *attrs, float *out)
{
    if (nSlots) hipLaunchKernelGGL(k_tex_density, dim3((nSlots + 255) / 256), dim3(256), 0, st, leafTris, nSlots, attrs, out);
}

// ---------------------------------------------------------------------------------------- QMC
// Random.h:26-34 (see oracle/oracle_qmc.cpp for the x86 conversion note)
HRD uint32_t toUint32(float f) { return (uint32_t)(unsigned long long)(long long)(f * 4294967296.0f); }
HRD float toNormalizedFloat(uint32_t u) { return (float)u * (1.0f / 4294967296.0f); }
HRD uint32_t burleyHash(uint32_t x) // Random.h:36-45
{
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
HRD uint32_t burleyHashCombine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); } // :47-50
HRD uint32_t nestedUniformScramble(uint32_t x, uint32_t seed) // Random.h:52-78 (bit reversal = v_bfrev_b32)
{
    x = __brev(x);
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return __brev(x);
}
HRD uint32_t sobolDim1(uint32_t index) // Random.h:236-244: v[b] = v[b-1] ^ (v[b-1] >> 1)
{
    uint32_t result = 0, v = 0x80000000u;
    for (uint32_t bit = 0; bit < 32; ++bit) {
        if ((index >> bit) & 1u) result ^= v;
        v ^= v >> 1;
    }
    return result;
}
HRD float haltonValue(uint32_t index, int base) // Random.h:192-204
{
    float result = 0.0f, f = 1.0f;
    const float denom = (float)base;
    uint32_t n = index;
    while (n > 0) {
        f = f / denom;
        result += f * (float)(n % (uint32_t)base);
        n = n / (uint32_t)base;
    }
    return result;
}
__constant__ int kHaltonBases[16][2] = {{2, 3},  {2, 5},  {2, 7},  {3, 7}, {4, 5},   {5, 7},  {5, 9},  {5, 11},
                                        {6, 11}, {5, 11}, {8, 11}, {3, 5}, {11, 15}, {2, 15}, {3, 19}, {7, 10}}; // Random.h:172-189

// owenScrambleSequence (Random.h:85-108) with sobol / halton / hammersley generators (radialSobol's disk mapping,
// Random.h:268-289, is applied on the host: hr_core.hip::radialOnHost)
__global__ __launch_bounds__(256) void k_qmc(int mode, uint32_t sequenceIndex, uint32_t count, float2 *__restrict__ out)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t seed = burleyHash(sequenceIndex + 1);
    const uint32_t index = nestedUniformScramble(i, seed);
    float sx, sy;
    if (mode == HR_SAMPLE_SOBOL) {
        sx = toNormalizedFloat(__brev(index)); // dimension 0: direction numbers 2^(31-bit)
        sy = toNormalizedFloat(sobolDim1(index));
    } else if (mode == HR_SAMPLE_HALTON) {
        sx = haltonValue(index, kHaltonBases[sequenceIndex & 15][0]);
        sy = haltonValue(index, kHaltonBases[sequenceIndex & 15][1]);
    } else {
        sx = (float)i * (1.0f / (float)count);
        sy = (float)__brev(index) * 2.3283064365386963e-10f;
    }
    float x = toNormalizedFloat(nestedUniformScramble(toUint32(sx), burleyHashCombine(seed, 0)));
    float y = toNormalizedFloat(nestedUniformScramble(toUint32(sy), burleyHashCombine(seed, 1)));
    out[i] = make_float2(x, y);
}

void launchQmc(hipStream_t st, int mode, uint32_t sequenceIndex, uint32_t count, float2 *out)
{
    if (count == 0) return;
    hipLaunchKernelGGL(k_qmc, dim3((count + 255) / 256), dim3(256), 0, st, mode, sequenceIndex, count, out);
}

// ---------------------------------------------------------------- tables of the sequential generators
// util::uniformRandomFloats (edges == 0) and util::randomPolygonal (edges = 5 / 6 / 8) — Random.h:113-130, 293-355; arithmetic in hr_tables.h.
// One workgroup per sequence: MT19937's state lives in LDS, seeded by one lane (a 623-step chain), twisted by 208 lanes in three rounds
// (word i needs words i+1 and i+397 from before the twist for i < 227 and word i-227 from after it: a round of 208 < 227 words never reads a
// word of its own round after it was written) and tempered 624 draws at a time.  uniformRandomFloats maps draw 2s / 2s+1 to sample s, in parallel;
// randomPolygonal consumes a data-dependent number of draws per sample (Lemire's rejection for the fan triangle, a rejection loop for the
// barycentrics), so one lane walks the block of draws with the four-state machine of the serial loop.  The polygon's vertices come from the
// host (cosf / sinf of the platform's libm, the same calls the reference makes: like radialSobol's disk mapping in hr_scene.inl).
struct PolyVerts { float x[8], y[8]; };
static constexpr int kMtRound = 208; // 3 x 208 = 624
__global__ __launch_bounds__(256) void k_mt_tables(uint32_t seed0, uint32_t count, uint32_t edges, PolyVerts verts, float2 *__restrict__ outBase,
                                                   size_t stride)
{
    __shared__ uint32_t st[kMtN], draws[kMtN];
    __shared__ uint32_t sDone;
    const uint32_t tid = threadIdx.x;
    float2 *out = outBase + (size_t)blockIdx.x * stride;
    if (tid == 0) {
        uint32_t v = seed0 + blockIdx.x;
        st[0] = v;
        for (uint32_t i = 1; i < (uint32_t)kMtN; ++i) st[i] = v = mtSeedNext(v, i);
        sDone = 0;
    }
    __syncthreads();
    uint32_t done = 0, phase = 0; // (lane 0's: samples finished; 0 = fan triangle, first draw, 1 = its retries, 2 = alpha, 3 = beta)
    MtIntDraw pick{edges, 0u, 0ull};
    float alpha = 0.0f;
    for (uint32_t drawBase = 0;; drawBase += (uint32_t)kMtN) {
        for (int r = 0; r < 3; ++r) {
            const uint32_t i = (uint32_t)(r * kMtRound) + tid;
            uint32_t w = 0;
            if (tid < (uint32_t)kMtRound) w = mtTwist(st[i], st[(i + 1) % kMtN], st[(i + kMtM) % kMtN]);
            __syncthreads();
            if (tid < (uint32_t)kMtRound) st[i] = w;
            __syncthreads();
        }
        for (uint32_t k = tid; k < (uint32_t)kMtN; k += 256) draws[k] = mtTemper(st[k]);
        __syncthreads();
        if (edges == 0) {
            float *o = reinterpret_cast<float *>(out);
            for (uint32_t k = tid; k < (uint32_t)kMtN; k += 256) {
                const uint64_t g = (uint64_t)drawBase + k;
                if (g < 2ull * count) o[g] = mtCanonical(draws[k]);
            }
            if ((uint64_t)drawBase + kMtN >= 2ull * count) break;
        } else {
            if (tid == 0) {
                for (int k = 0; k < kMtN && done < count; ++k) {
                    const uint32_t u = draws[k];
                    if (phase < 2) {
                        phase = pick.accept(u, phase == 0) ? 2u : 1u;
                    } else if (phase == 2) {
                        alpha = mtCanonical(u), phase = 3;
                    } else {
                        const float beta = mtCanonical(u);
                        if (alpha + beta > 1.0f) {
                            phase = 2;
                            continue;
                        }
                        const float gamma = 1.0f - (alpha + beta);
                        const int t0 = pick.value(), t1 = (t0 + 1) % (int)edges;
                        const float vx = 0.0f * alpha + verts.x[t0] * beta + verts.x[t1] * gamma; // (the fan's centre is vertex 0 of every triangle)
                        const float vy = 0.0f * alpha + verts.y[t0] * beta + verts.y[t1] * gamma;
                        out[done++] = make_float2((vx + 1.0f) * 0.5f, (vy + 1.0f) * 0.5f);
                        phase = 0;
                    }
                }
                sDone = done;
            }
            __syncthreads();
            if (sDone >= count) break;
        }
    }
}
void launchMtTables(hipStream_t st, uint32_t seed0, int nSeq, uint32_t count, uint32_t edges, const float *vx, const float *vy, float2 *out, size_t stride)
{
    if (count == 0 || nSeq <= 0) return;
    PolyVerts v{};
    for (uint32_t i = 0; i < edges && i < 8u; ++i) v.x[i] = vx[i], v.y[i] = vy[i];
    hipLaunchKernelGGL(k_mt_tables, dim3((uint32_t)nSeq), dim3(256), 0, st, seed0, count, edges, v, out, stride);
}

// util::blueNoise (BlueNoise.h:52-88): point i is the best of 30 hashed candidates, "best" = furthest from its nearest earlier point.  The
// candidates do not depend on the points (a hash of a running seed), so one launch makes them all; the choice is sequential over the points
// and parallel inside a point: a workgroup per sequence, 30 groups of 32 lanes — a group per candidate, a lane per 32nd earlier point —
// min over squared distances (the square root is monotone, it is taken once per candidate), then one lane picks the first candidate whose
// nearest distance exceeds the running maximum, as the serial loop does.  15 n^2 distance tests per sequence of n points.
static constexpr int kBlueCandidates = 30, kBlueLds = 8192;
__global__ __launch_bounds__(256) void k_blue_candidates(int32_t seq0, uint32_t nSeq, uint32_t count, float2 *__restrict__ cand, float2 *__restrict__ outBase, size_t stride)
{
    const uint32_t perSeq = (count - 1u) * (uint32_t)kBlueCandidates + 1u; // (entry 0: the first point itself)
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t s = (uint32_t)(gid / perSeq), k = (uint32_t)(gid % perSeq);
    if (s >= nSeq) return;
    const uint32_t seed = blueSeed(seq0 + (int32_t)s) + 2u * k;
    const float2 p = make_float2(blueRandom(seed), blueRandom(seed + 1u));
    if (k == 0)
        outBase[(size_t)s * stride] = p;
    else
        cand[(size_t)s * (perSeq - 1u) + (k - 1u)] = p;
}
template <bool LDS>
__global__ __launch_bounds__(1024) void k_blue_noise(uint32_t count, const float2 *__restrict__ candBase, float2 *__restrict__ outBase, size_t stride)
{
    __shared__ float2 ptsLds[LDS ? kBlueLds : 1];
    __shared__ float nearest[32];
    float2 *out = outBase + (size_t)blockIdx.x * stride;
    const float2 *cand = candBase + (size_t)blockIdx.x * (count - 1u) * kBlueCandidates;
    float2 *pts = LDS ? ptsLds : out;
    const uint32_t tid = threadIdx.x, g = tid >> 5, l = tid & 31u;
    if (LDS && tid == 0) ptsLds[0] = out[0];
    __syncthreads();
    const float diag = sqrt_(1.0f * 1.0f + 1.0f * 1.0f);
    for (uint32_t i = 1; i < count; ++i) {
        if (g < (uint32_t)kBlueCandidates) {
            const float2 c = cand[(size_t)(i - 1u) * kBlueCandidates + g];
            float m = 3.0e38f;
            for (uint32_t j = l; j < i; j += 32u) {
                const float2 p = pts[j];
                const float dx = p.x - c.x, dy = p.y - c.y;
                m = fmin_(m, dx * dx + dy * dy);
            }
#pragma unroll
            for (int sh = 16; sh > 0; sh >>= 1) m = fmin_(m, __shfl_xor(m, sh));
            if (l == 0) nearest[g] = fmin_(diag, sqrt_(m));
        }
        __syncthreads();
        if (tid == 0) {
            float furthest = 0.0f;
            int best = -1;
            for (int c = 0; c < kBlueCandidates; ++c)
                if (nearest[c] > furthest) furthest = nearest[c], best = c;
            const float2 p = best < 0 ? make_float2(0.0f, 0.0f) : cand[(size_t)(i - 1u) * kBlueCandidates + best];
            pts[i] = p;
            if (LDS) out[i] = p;
        }
        __syncthreads();
    }
}
// cand: scratch of nSeq * (count - 1) * 30 points
void launchBlueNoise(hipStream_t st, int32_t seq0, int nSeq, uint32_t count, float2 *out, size_t stride, float2 *cand)
{
    if (count == 0 || nSeq <= 0) return;
    const uint64_t total = (uint64_t)nSeq * ((uint64_t)(count - 1u) * kBlueCandidates + 1u);
    hipLaunchKernelGGL(k_blue_candidates, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, seq0, (uint32_t)nSeq, count, cand, out, stride);
    int dev = 0, ldsMax = 0; // (the LDS copy of the points is 64 KB + the candidates' row: more than a gfx9 before gfx950 gives one workgroup)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ldsMax, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) ldsMax = 0;
    if (count <= (uint32_t)kBlueLds && (size_t)ldsMax >= sizeof(float2) * kBlueLds + 256)
        hipLaunchKernelGGL(k_blue_noise<true>, dim3((uint32_t)nSeq), dim3(1024), 0, st, count, cand, out, stride);
    else
        hipLaunchKernelGGL(k_blue_noise<false>, dim3((uint32_t)nSeq), dim3(1024), 0, st, count, cand, out, stride);
}

// ------------------------------------------------------------------------------ multiscatter LUT
// MultiScatterUtil.cpp:20-139: one thread per texel integrates 4096 Sobol samples of the GGX lobe.
HRD float lutG1(float NdotI, float alpha)
{
    const float alpha2 = alpha * alpha;
    const float denom = sqrt_(alpha2 + (1.0f - alpha2) * (NdotI * NdotI)) + NdotI;
    return (2.0f * NdotI) / fmax_(denom, 1e-5f);
}
__global__ __launch_bounds__(128) void k_multiscatter_lut(const float2 *__restrict__ seq, int samples, float *__restrict__ out, int dim)
{
    const int col = threadIdx.x, row = blockIdx.x;
    const float roughness = clamp_(((float)row + 0.5f) / (float)dim, 0.0f, 1.0f);
    const float alpha = roughness * roughness;
    const float NdotV = clamp_(((float)col + 0.5f) / (float)dim, 0.0f, 1.0f);
    const v3 V(sqrt_(1.0f - (NdotV * NdotV)), 0.0f, NdotV);
    float result = 0.0f;
    for (int i = 0; i < samples; ++i) {
        const float2 r = seq[i];
        const float a2 = alpha * alpha;
        const float cosTheta = sqrt_(fmax_(0.0f, (1.0f - r.x) / ((a2 - 1.0f) * r.x + 1.0f)));
        const float sinTheta = sqrt_(fmax_(0.0f, 1.0f - cosTheta * cosTheta));
        float sn, cs;
        sincos_(6.28318530717958647692f * r.y, &sn, &cs);
        v3 H(sinTheta * cs, sinTheta * sn, cosTheta);
        H = normalize(H);
        const v3 L = 2.0f * dot(V, H) * H - V;
        const float NdotL = clamp_(L.z, 0.0f, 1.0f);
        if (NdotL > 0.0f) {
            const float VdotH = clamp_(dot(V, H), 0.0f, 1.0f);
            const float NdotH = clamp_(H.z, 0.0f, 1.0f);
            result += (lutG1(NdotL, alpha) * lutG1(NdotV, alpha) * VdotH) / (NdotV * NdotH);
        }
    }
    const float value = result / (float)samples;
    out[row * dim + col] = (1.0f - value) / value;
}

void launchMultiscatterLUT(hipStream_t st, const float2 *sobol4096, float *out128x128)
{
    hipLaunchKernelGGL(k_multiscatter_lut, dim3(128), dim3(128), 0, st, sobol4096, 4096, out128x128, 128);
}

} // namespace hr
