// hr_texture.h — texture2D at LOD 0 (bilinear / nearest, repeat / clamp-to-edge), and the optional mip chain of HR_TEXTURE_LOD_CONE.
//
// Replaces OpenRL's texture unit for openrl::Texture objects
// (/root/reference/Source/RLWrapper/Texture.h:26-93).  OpenRL's filtering without ray differentials
// is parity-unpinned; this build samples level 0 with GL texel addressing.
#pragma once

#include "hr_types.h"

namespace hr {

HRD int wrapIndex(int i, int n, int mode)
{
    if (mode == HR_WRAP_CLAMP_TO_EDGE) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    int m = i % n;
    return m < 0 ? m + n : m;
}

template <class TD> HRD v4 texel(const TD &t, int x, int y)
{
    const size_t at = ((size_t)y * t.w + x) * t.c;
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (t.dtype == HR_TEX_U8) { // openrl::Texture with RL_UNSIGNED_BYTE data: a quarter of the bytes of a float copy per texel fetched
        const HR_GLOBAL uint8_t *b = (const HR_GLOBAL uint8_t *)t.px + at;
        for (int k = 0; k < t.c; ++k) p[k] = (float)b[k] / 255.0f;
    } else {
        const HR_GLOBAL float *f = (const HR_GLOBAL float *)t.px + at;
        for (int k = 0; k < t.c; ++k) p[k] = f[k];
    }
    v4 r;
    if (t.c == 1) { // RL_LUMINANCE
        r.x = r.y = r.z = p[0];
        r.w = 1.0f;
    } else if (t.c == 3) {
        r.x = p[0], r.y = p[1], r.z = p[2], r.w = 1.0f;
    } else {
        r.x = p[0], r.y = p[1], r.z = p[2], r.w = p[3];
    }
    return r;
}

template <class TD> HRD v4 sampleTexture(const TD &t, float u, float v)
{
    if (t.filter == HR_FILTER_NEAREST) {
        int x = wrapIndex((int)floor_(u * (float)t.w), t.w, t.wrapS);
        int y = wrapIndex((int)floor_(v * (float)t.h), t.h, t.wrapT);
        return texel(t, x, y);
    }
    float x = u * (float)t.w - 0.5f;
    float y = v * (float)t.h - 0.5f;
    float x0f = floor_(x), y0f = floor_(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapIndex((int)x0f, t.w, t.wrapS), x1 = wrapIndex((int)x0f + 1, t.w, t.wrapS);
    int y0 = wrapIndex((int)y0f, t.h, t.wrapT), y1 = wrapIndex((int)y0f + 1, t.h, t.wrapT);
    v4 c00 = texel(t, x0, y0), c10 = texel(t, x1, y0), c01 = texel(t, x0, y1), c11 = texel(t, x1, y1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    v4 r;
    r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
    r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
    r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
    r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
    return r;
}

// ---- HR_TEXTURE_LOD_CONE (include/hrcore.h): mip chain + trilinear lookup.  Not in the reference path (OpenRL's level selection
// in ray shaders is closed); the arithmetic below is the oracle's (oracle/oracle_scene.cpp), operation for operation.
HRD int mipDim(int n, int level)
{
    const int d = n >> level;
    return d < 1 ? 1 : d;
}
// element offset of level `level` (>= 1) inside TexDesc::mips
template <class TD> HRD size_t mipOffset(const TD &t, int level)
{
    size_t off = 0;
    for (int l = 1; l < level; ++l) off += (size_t)mipDim(t.w, l) * (size_t)mipDim(t.h, l) * (size_t)t.c;
    return off;
}
template <class TD> HRD v4 texelAt(const TD &t, int level, size_t levelOff, int lw, int x, int y)
{
    if (level == 0) return texel(t, x, y);
    const HR_GLOBAL float *f = (const HR_GLOBAL float *)t.mips + levelOff + ((size_t)y * lw + x) * t.c;
    v4 r;
    if (t.c == 1) {
        r.x = r.y = r.z = f[0], r.w = 1.0f;
    } else if (t.c == 3) {
        r.x = f[0], r.y = f[1], r.z = f[2], r.w = 1.0f;
    } else {
        r.x = f[0], r.y = f[1], r.z = f[2], r.w = f[3];
    }
    return r;
}
template <class TD> HRD v4 sampleTextureLevel(const TD &t, int level, float u, float v)
{
    if (level == 0) return sampleTexture(t, u, v);
    const int lw = mipDim(t.w, level), lh = mipDim(t.h, level);
    const size_t off = mipOffset(t, level);
    float x = u * (float)lw - 0.5f;
    float y = v * (float)lh - 0.5f;
    float x0f = floor_(x), y0f = floor_(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapIndex((int)x0f, lw, t.wrapS), x1 = wrapIndex((int)x0f + 1, lw, t.wrapS);
    int y0 = wrapIndex((int)y0f, lh, t.wrapT), y1 = wrapIndex((int)y0f + 1, lh, t.wrapT);
    v4 c00 = texelAt(t, level, off, lw, x0, y0), c10 = texelAt(t, level, off, lw, x1, y0), c01 = texelAt(t, level, off, lw, x0, y1),
       c11 = texelAt(t, level, off, lw, x1, y1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    v4 r;
    r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
    r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
    r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
    r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
    return r;
}
// trilinear: `lambda` is the level at which one texel covers the footprint (clamped to the chain)
template <class TD> HRD v4 sampleTextureLod(const TD &t, float u, float v, float lambda)
{
    if (t.nLevels <= 1 || t.filter == HR_FILTER_NEAREST || !(lambda > 0.0f)) return sampleTexture(t, u, v);
    const float top = (float)(t.nLevels - 1);
    const float l = lambda < top ? lambda : top;
    const float lf = floor_(l);
    const int l0 = (int)lf;
    const float f = l - lf;
    const v4 a = sampleTextureLevel(t, l0, u, v);
    if (!(f > 0.0f)) return a;
    const v4 b = sampleTextureLevel(t, l0 + 1, u, v);
    const float g = 1.0f - f;
    v4 r;
    r.x = a.x * g + b.x * f, r.y = a.y * g + b.y * f, r.z = a.z * g + b.z * f, r.w = a.w * g + b.w * f;
    return r;
}

} // namespace hr
