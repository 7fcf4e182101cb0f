// oracle_scene.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// Scene assembly: what OpenRL does with the buffers handed over by
// Mesh::Mesh (/root/reference/Source/HeatrayRenderer/Scene/Mesh.cpp:29-153) and the
// vertex shader (Resources/shaders/vertex.rlsl:25-43), plus texture sampling
// (closed inside OpenRL; bilinear at LOD 0 is this oracle's stated assumption,
// SURVEY §8a row a6).
//
// PARITY STATUS: vertex transform and strip ordering follow vertex.rlsl / GL conventions; texture filtering inside
// OpenRL is "parity unpinned" (no fixture in the reference pins it) — level-0 bilinear is this oracle's definition.
#include "oracle_internal.h"

#include <cmath>

namespace ora {

static inline int wrapIndex(int i, int n, int mode)
{
    if (mode == HR_WRAP_CLAMP_TO_EDGE) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    int m = i % n;
    return m < 0 ? m + n : m;
}

static inline vec4 texel(const Texture &t, int x, int y)
{
    const float *p = &t.px[((size_t)y * t.w + x) * t.c];
    if (t.c == 1) return vec4{p[0], p[0], p[0], 1.0f}; // RL_LUMINANCE
    if (t.c == 3) return vec4{p[0], p[1], p[2], 1.0f};
    return vec4{p[0], p[1], p[2], p[3]};
}

// texture2D at LOD 0.  GL texel addressing: texel centres at (i + 0.5) / size.
vec4 texelOf(const Texture &t, int x, int y) { return texel(t, x, y); }

// HR_ESTIMATOR_ENV_MIS: the importance table of the environment map (EnvTable, oracle_internal.h)
void buildEnvTable(Context &ctx)
{
    EnvTable &E = ctx.env;
    const int id = ctx.lights.env_texture;
    if (!ctx.lights.env_enabled || id < 0 || id >= (int)ctx.textures.size() || !ctx.textures[id].alive) {
        E = EnvTable();
        return;
    }
    if (E.tex == id && E.w == ctx.textures[id].w && E.h == ctx.textures[id].h) return;
    const Texture &T = ctx.textures[id];
    const int w = T.w, h = T.h;
    E.w = w, E.h = h, E.tex = id;
    std::vector<float> lum((size_t)w * h);
    float maxLum = 0.0f;
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            const vec4 c = texel(T, i, j);
            float l = (c.x * 0.33f + c.y * 0.59f) + c.z * 0.11f; // utility.rlsl:163-166 luminosity
            l = l > 0.0f ? l : 0.0f;
            l = l < 1e30f ? l : 1e30f; // (an infinite texel must not turn the normalisation into Inf / Inf)
            lum[(size_t)j * w + i] = l;
            maxLum = l > maxLum ? l : maxLum;
        }
    // the light shader filters the map bilinearly, which smears a bright texel over its neighbours: weight every texel with the
    // brightest of its 3 x 3 neighbourhood (wrapping in azimuth), so that the halo is importance-sampled too
    {
        std::vector<float> dil((size_t)w * h);
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i) {
                float m = 0.0f;
                for (int dj = -1; dj <= 1; ++dj) {
                    const int jj = j + dj < 0 ? 0 : (j + dj >= h ? h - 1 : j + dj);
                    for (int di = -1; di <= 1; ++di) {
                        const int ii = (i + di + w) % w;
                        const float l = lum[(size_t)jj * w + ii];
                        m = l > m ? l : m;
                    }
                }
                dil[(size_t)j * w + i] = m;
            }
        lum.swap(dil);
    }
    const float floorLum = maxLum * (1.0f / 65536.0f); // keeps every texel reachable without taking the samples away from the bright ones
    const float norm = maxLum + floorLum; // >= every weight (cos <= 1)
    std::vector<uint32_t> wq((size_t)w * h);
    std::vector<unsigned long long> rowSum(h);
    unsigned long long total = 0;
    for (int j = 0; j < h; ++j) {
        const float elevation = (((float)j + 0.5f) / (float)h - 0.5f) * kPI; // row 0 = bottom of the map = looking down
        const float c = cos_(elevation);
        unsigned long long r = 0;
        for (int i = 0; i < w; ++i) {
            const float wt = (lum[(size_t)j * w + i] + floorLum) * c;
            const float q = norm > 0.0f ? (wt / norm) * 1048576.0f : 0.0f;
            const uint32_t v = (uint32_t)q + 1u;
            wq[(size_t)j * w + i] = v;
            r += v;
        }
        rowSum[j] = r;
        total += r;
    }
    E.rowCdf.assign((size_t)h + 1, 0.0f), E.colCdf.assign((size_t)h * (w + 1), 0.0f), E.prob.assign((size_t)w * h, 0.0f);
    unsigned long long acc = 0;
    for (int j = 0; j < h; ++j) {
        E.rowCdf[j] = (float)acc / (float)total;
        acc += rowSum[j];
        unsigned long long a = 0;
        for (int i = 0; i < w; ++i) {
            E.colCdf[(size_t)j * (w + 1) + i] = (float)a / (float)rowSum[j];
            a += wq[(size_t)j * w + i];
        }
        E.colCdf[(size_t)j * (w + 1) + w] = 1.0f;
    }
    E.rowCdf[h] = 1.0f;
    // The probability of a texel is what sampleEnv() really draws it with: the float CDF steps the two searches invert — not the
    // exact ratio of the integer weights, which a float CDF cannot resolve for dim texels of a map with a 2^20 : 1 weight range
    // (the balance heuristic needs the density the samples are drawn with).
    for (int j = 0; j < h; ++j) {
        const float pRow = E.rowCdf[j + 1] - E.rowCdf[j];
        for (int i = 0; i < w; ++i) {
            const float cThis = E.colCdf[(size_t)j * (w + 1) + i], cNext = E.colCdf[(size_t)j * (w + 1) + i + 1];
            E.prob[(size_t)j * w + i] = pRow * (cNext - cThis);
        }
    }
    // mean luminosity over the sphere from the same integers: sum(weight) / sum(cos) — rows in order, one float addition per row
    float sumC = 0.0f;
    for (int j = 0; j < h; ++j) sumC = sumC + cos_((((float)j + 0.5f) / (float)h - 0.5f) * kPI);
    E.meanLum = (((float)total / 1048576.0f) * norm) / ((float)w * sumC);
}

vec4 sampleTexture(const Texture &t, float u, float v)
{
    if (t.filter == HR_FILTER_NEAREST) {
        int x = wrapIndex((int)floorf(u * (float)t.w), t.w, t.wrapS);
        int y = wrapIndex((int)floorf(v * (float)t.h), t.h, t.wrapT);
        return texel(t, x, y);
    }
    float x = u * (float)t.w - 0.5f;
    float y = v * (float)t.h - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapIndex((int)x0f, t.w, t.wrapS), x1 = wrapIndex((int)x0f + 1, t.w, t.wrapS);
    int y0 = wrapIndex((int)y0f, t.h, t.wrapT), y1 = wrapIndex((int)y0f + 1, t.h, t.wrapT);
    vec4 c00 = texel(t, x0, y0), c10 = texel(t, x1, y0), c01 = texel(t, x0, y1), c11 = texel(t, x1, y1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    vec4 r;
    r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
    r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
    r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
    r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
    return r;
}

// ---- HR_TEXTURE_LOD_CONE (include/hrcore.h): mip chain + trilinear lookup; the product's heatray_amd/csrc/hr_texture.h and the
// mip / density kernels of hr_build.hip follow THIS arithmetic, operation for operation.
static inline int mipDim(int n, int level)
{
    const int d = n >> level;
    return d < 1 ? 1 : d;
}
static inline size_t mipOffset(const Texture &t, int level)
{
    size_t off = 0;
    for (int l = 1; l < level; ++l) off += (size_t)mipDim(t.w, l) * (size_t)mipDim(t.h, l) * (size_t)t.c;
    return off;
}
static inline vec4 texelAt(const Texture &t, int level, size_t levelOff, int lw, int x, int y)
{
    if (level == 0) return texel(t, x, y);
    const float *f = &t.mips[levelOff + ((size_t)y * lw + x) * t.c];
    if (t.c == 1) return vec4{f[0], f[0], f[0], 1.0f};
    if (t.c == 3) return vec4{f[0], f[1], f[2], 1.0f};
    return vec4{f[0], f[1], f[2], f[3]};
}
static vec4 sampleTextureLevel(const Texture &t, int level, float u, float v)
{
    if (level == 0) return sampleTexture(t, u, v);
    const int lw = mipDim(t.w, level), lh = mipDim(t.h, level);
    const size_t off = mipOffset(t, level);
    float x = u * (float)lw - 0.5f;
    float y = v * (float)lh - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapIndex((int)x0f, lw, t.wrapS), x1 = wrapIndex((int)x0f + 1, lw, t.wrapS);
    int y0 = wrapIndex((int)y0f, lh, t.wrapT), y1 = wrapIndex((int)y0f + 1, lh, t.wrapT);
    vec4 c00 = texelAt(t, level, off, lw, x0, y0), c10 = texelAt(t, level, off, lw, x1, y0), c01 = texelAt(t, level, off, lw, x0, y1),
         c11 = texelAt(t, level, off, lw, x1, y1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    vec4 r;
    r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
    r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
    r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
    r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
    return r;
}
vec4 sampleTextureLod(const Texture &t, float u, float v, float lambda)
{
    if (t.nLevels <= 1 || t.filter == HR_FILTER_NEAREST || !(lambda > 0.0f)) return sampleTexture(t, u, v);
    const float top = (float)(t.nLevels - 1);
    const float l = lambda < top ? lambda : top;
    const float lf = floorf(l);
    const int l0 = (int)lf;
    const float f = l - lf;
    const vec4 a = sampleTextureLevel(t, l0, u, v);
    if (!(f > 0.0f)) return a;
    const vec4 b = sampleTextureLevel(t, l0 + 1, u, v);
    const float g = 1.0f - f;
    return vec4{a.x * g + b.x * f, a.y * g + b.y * f, a.z * g + b.z * f, a.w * g + b.w * f};
}

void buildTextureLod(Context &ctx)
{
    for (Texture &t : ctx.textures) {
        if (!t.alive || t.nLevels != 0) continue;
        int levels = 1;
        t.mips.clear();
        const float *src = t.px.data();
        int sw = t.w, sh = t.h;
        std::vector<float> prev;
        while ((sw > 1 || sh > 1) && t.filter != HR_FILTER_NEAREST) {
            const int dw = sw / 2 < 1 ? 1 : sw / 2, dh = sh / 2 < 1 ? 1 : sh / 2;
            std::vector<float> lvl((size_t)dw * dh * t.c);
            for (int y = 0; y < dh; ++y)
                for (int x = 0; x < dw; ++x) {
                    const int x0 = 2 * x < sw ? 2 * x : sw - 1, x1 = 2 * x + 1 < sw ? 2 * x + 1 : sw - 1;
                    const int y0 = 2 * y < sh ? 2 * y : sh - 1, y1 = 2 * y + 1 < sh ? 2 * y + 1 : sh - 1;
                    for (int k = 0; k < t.c; ++k) {
                        const float a = src[((size_t)y0 * sw + x0) * t.c + k], b = src[((size_t)y0 * sw + x1) * t.c + k];
                        const float c = src[((size_t)y1 * sw + x0) * t.c + k], d = src[((size_t)y1 * sw + x1) * t.c + k];
                        lvl[((size_t)y * dw + x) * t.c + k] = ((a + b) + (c + d)) * 0.25f;
                    }
                }
            t.mips.insert(t.mips.end(), lvl.begin(), lvl.end());
            prev.swap(lvl);
            src = prev.data(), sw = dw, sh = dh;
            ++levels;
        }
        t.nLevels = levels;
        t.lodScale = 0.5f * (log_((float)t.w * (float)t.h) * 1.4426950408889634f);
    }
    const size_t n = ctx.tris.size();
    ctx.texDensity.assign(n, -1e30f);
    for (size_t i = 0; i < n; ++i) {
        const Tri &tr = ctx.tris[i];
        const TriAttr &a = ctx.attrs[i];
        const float world2 = length(cross(tr.e1, tr.e2));
        const float du1 = a.uv[1].x - a.uv[0].x, dv1 = a.uv[1].y - a.uv[0].y, du2 = a.uv[2].x - a.uv[0].x, dv2 = a.uv[2].y - a.uv[0].y;
        const float uv2 = fabsf(du1 * dv2 - dv1 * du2);
        if (world2 > 0.0f && uv2 > 0.0f) ctx.texDensity[i] = 0.5f * (log_(uv2 / world2) * 1.4426950408889634f);
    }
}

static inline vec3 fetch3(const std::vector<float> &a, uint32_t i) { return vec3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

void commitScene(Context &ctx)
{
    ctx.tris.clear();
    ctx.attrs.clear();
    for (const Geom &g : ctx.geoms) {
        if (!g.alive) continue;
        const bool strip = (g.mode == HR_TRIANGLE_STRIP);
        const size_t nTri = strip ? (g.idx.size() >= 3 ? g.idx.size() - 2 : 0) : g.idx.size() / 3;
        for (size_t t = 0; t < nTri; ++t) {
            uint32_t i0, i1, i2;
            if (strip) { // RL_TRIANGLE_STRIP, GL ordering: odd triangles swap the first two vertices
                i0 = g.idx[t], i1 = g.idx[t + 1], i2 = g.idx[t + 2];
                if (t & 1) {
                    uint32_t s = i0;
                    i0 = i1;
                    i1 = s;
                }
            } else {
                i0 = g.idx[3 * t], i1 = g.idx[3 * t + 1], i2 = g.idx[3 * t + 2];
            }
            const uint32_t id[3] = {i0, i1, i2};
            // vertex.rlsl:27 — rl_Position = worldFromEntity * vec4(position, 1)
            vec3 p[3];
            for (int k = 0; k < 3; ++k) p[k] = xformPoint(g.world, fetch3(g.pos, id[k]));
            Tri tri;
            tri.v0 = p[0];
            tri.e1 = p[1] - p[0];
            tri.e2 = p[2] - p[0];
            TriAttr a{};
            a.material = g.material;
            a.flags = (g.frontFaceCW ? TF_FRONT_CW : 0u) | (g.isOccluder ? 0u : TF_NON_OCCLUDER);
            for (int k = 0; k < 3; ++k) {
                // vertex.rlsl:28-29 — normal = mat3(worldFromEntity) * normalAttribute (no inverse-transpose)
                a.n[k] = xformVector(g.world, fetch3(g.nrm, id[k]));
                if (!g.uv.empty()) a.uv[k] = vec2{g.uv[2 * id[k]], g.uv[2 * id[k] + 1]};
                if (!g.tan.empty() && !g.bit.empty()) { // vertex.rlsl:35-38
                    a.tan[k] = xformVector(g.world, fetch3(g.tan, id[k]));
                    a.bit[k] = xformVector(g.world, fetch3(g.bit, id[k]));
                }
                if (!g.col.empty()) a.col[k] = fetch3(g.col, id[k]); // vertex.rlsl:40-42
            }
            if (!g.uv.empty()) a.flags |= TF_HAS_UV;
            if (!g.tan.empty() && !g.bit.empty()) a.flags |= TF_HAS_TANGENTS;
            if (!g.col.empty()) a.flags |= TF_HAS_COLORS;
            ctx.tris.push_back(tri);
            ctx.attrs.push_back(a);
        }
    }
    // Scene bounds over every triangle vertex; self-intersection epsilon.
    vec3 lo(INFINITY), hi(-INFINITY);
    for (const Tri &t : ctx.tris) {
        const vec3 p[3] = {t.v0, t.v0 + t.e1, t.v0 + t.e2};
        for (int k = 0; k < 3; ++k) {
            lo = min3(lo, p[k]);
            hi = max3(hi, p[k]);
        }
    }
    if (ctx.tris.empty()) lo = hi = vec3(0.0f);
    ctx.aabbLo[0] = lo.x, ctx.aabbLo[1] = lo.y, ctx.aabbLo[2] = lo.z;
    ctx.aabbHi[0] = hi.x, ctx.aabbHi[1] = hi.y, ctx.aabbHi[2] = hi.z;
    ctx.rayEps = 1e-4f * length(hi - lo);
    ctx.hitPad = 0.5f * (1e-5f * length(hi - lo)); // half of buildLBVH's leaf padding
    buildLBVH(ctx.tris, ctx.bvh);
    ctx.committed = true;
}

// physicallyBased.rlsl:57-91 as seen by an occlusion ray on a non-occluder (alpha-masked)
// primitive: alpha < 1 lets the ray continue, otherwise the ray is shadowed.
bool alphaPasses(const Context &ctx, int prim, float u, float v)
{
    const TriAttr &a = ctx.attrs[prim];
    if (a.material < 0 || a.material >= (int)ctx.materials.size()) return false;
    const hr_material &m = ctx.materials[a.material];
    if (m.type != HR_MAT_PBR || !(m.flags & HR_MF_ALPHA_MASK)) return false;
    float alpha = 1.0f;
    if ((m.flags & HR_MF_HAS_BASE_COLOR_TEXTURE) && m.base_color_texture >= 0 &&
        m.base_color_texture < (int)ctx.textures.size() && ctx.textures[m.base_color_texture].alive) {
        float w = 1.0f - u - v;
        float tu = a.uv[0].x * w + a.uv[1].x * u + a.uv[2].x * v;
        float tv = a.uv[0].y * w + a.uv[1].y * u + a.uv[2].y * v;
        alpha = sampleTexture(ctx.textures[m.base_color_texture], tu, tv).w;
    }
    return alpha < 1.0f;
}

} // namespace ora
