// oracle_display.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// The display resolve: /root/reference/Resources/shaders/displayGL.frag:74-151 evaluated per
// pixel of the accumulation buffer, with the uniforms DisplayProgram::bind uploads
// (/root/reference/Source/HeatrayRenderer/HeatrayRenderer.h:222-248) and the full-screen quad of displayGL.vert
// (texture coordinate of a fragment = its pixel centre).  The HDR format follows saveScreenshot
// (/root/reference/Source/HeatrayRenderer/HeatrayRenderer.cpp:1624-1645).
//
// PARITY STATUS: "parity unpinned" — the reference has no fixture for its display shader and GLSL leaves the
// precision of pow() to the GL implementation; pow is defined here as exp(y * log(x)) (oracle_math.h).  Known-answer
// tests (tests/test_oracle_display.py) pin the sRGB curve, the ACES fit and the identity of the neutral settings.
#include "oracle_internal.h"

#include <cmath>
#include <cstdint>

namespace ora {

// displayGL.frag:47-58
static inline float linearToSRGB1(float c)
{
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * pow_(c, 1.0f / 2.4f) - 0.055f;
}
// displayGL.frag:60-72
static inline float srgbToLinear1(float c)
{
    if (c <= 0.04045f) return c / 12.92f;
    return pow_((c + 0.055f) / (1.0f + 0.055f), 2.4f);
}
// displayGL.frag:40-45
static inline float rrtAndOdtFit1(float v)
{
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
static inline float step_(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

// displayGL.frag:74-151 for one fragment; rgba = accumulated sample sum and sample count, (u, v) = textureCoords
static inline void displayFragment(const float rgba[4], float u, float v, const hr_display_params &P, float out[3])
{
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (rgba[3] != 0.0f) r = rgba[0] / rgba[3], g = rgba[1] / rgba[3], b = rgba[2] / rgba[3]; // :78 (no samples yet: black)
    if (P.tonemapping_enabled == 1) { // :82-90
        r = linearToSRGB1(r), g = linearToSRGB1(g), b = linearToSRGB1(b);
        // ACESInputMat * v, columns (:28-32): c0*v.x + c1*v.y + c2*v.z
        float x = (0.59719f * r + 0.35458f * g) + 0.04823f * b;
        float y = (0.07600f * r + 0.90834f * g) + 0.01566f * b;
        float z = (0.02840f * r + 0.13383f * g) + 0.83777f * b;
        x = rrtAndOdtFit1(x), y = rrtAndOdtFit1(y), z = rrtAndOdtFit1(z);
        // ACESOutputMat (:34-38)
        r = (1.60475f * x + -0.53108f * y) + -0.07367f * z;
        g = (-0.10208f * x + 1.10813f * y) + -0.00605f * z;
        b = (-0.00327f * x + -0.07276f * y) + 1.07602f * z;
        r = clamp_(r, 0.0f, 1.0f), g = clamp_(g, 0.0f, 1.0f), b = clamp_(b, 0.0f, 1.0f);
        r = srgbToLinear1(r), g = srgbToLinear1(g), b = srgbToLinear1(b);
    }
    // :95-97 brightness / contrast
    r = (r - 0.5f) * P.contrast + 0.5f + P.brightness;
    g = (g - 0.5f) * P.contrast + 0.5f + P.brightness;
    b = (b - 0.5f) * P.contrast + 0.5f + P.brightness;
    // :100-129 hue / saturation / vibrance through HSV
    {
        const float kx = 0.0f, ky = -1.0f / 3.0f, kz = 2.0f / 3.0f, kw = -1.0f;
        const float s1 = step_(b, g);
        const float px = mix(b, g, s1), py = mix(g, b, s1), pz = mix(kw, kx, s1), pw = mix(kz, ky, s1);
        const float s2 = step_(px, r);
        const float qx = mix(px, r, s2), qy = mix(py, py, s2), qz = mix(pw, pz, s2), qw = mix(r, px, s2);
        const float d = qx - fmin_(qw, qy);
        const float e = 1.0e-10f;
        float hx = fabsf(qz + (qw - qy) / (6.0f * d + e)), hy = d / (qx + e), hz = qx;
        hx = hx * P.hue;
        hy = hy * P.saturation;
        const float mapped = sqrtf(hy) * P.vibrance;
        hy = hy * (1.0f + mapped);
        const float k1 = 1.0f, k2 = 2.0f / 3.0f, k3 = 1.0f / 3.0f, k4 = 3.0f;
        const float p1 = fabsf(fract(hx + k1) * 6.0f - k4), p2 = fabsf(fract(hx + k2) * 6.0f - k4), p3 = fabsf(fract(hx + k3) * 6.0f - k4);
        r = hz * mix(k1, clamp_(p1 - k1, 0.0f, 1.0f), hy);
        g = hz * mix(k1, clamp_(p2 - k1, 0.0f, 1.0f), hy);
        b = hz * mix(k1, clamp_(p3 - k1, 0.0f, 1.0f), hy);
    }
    // :132-136 RGB levels
    r = r * P.red, g = g * P.green, b = b * P.blue;
    // :139-143 vignette (the "+ blue" is the reference's)
    {
        const float dx = 0.5f - u, dy = 0.5f - v;
        const float dist = sqrtf(dx * dx + dy * dy);
        const float vig = smoothstep(0.8f, P.vignette_falloff * 0.799f, dist * (P.vignette_intensity + P.blue));
        r = r * vig, g = g * vig, b = b * vig;
    }
    // :146 exposure, :149 encoding
    r = r * P.camera_exposure, g = g * P.camera_exposure, b = b * P.camera_exposure;
    out[0] = linearToSRGB1(r), out[1] = linearToSRGB1(g), out[2] = linearToSRGB1(b);
}

static inline uint32_t toByte(float c)
{
    if (!(c == c)) return 0u; // NaN
    return (uint32_t)floorf(clamp_(c, 0.0f, 1.0f) * 255.0f + 0.5f);
}

void displayResolve(const Context &ctx, const hr_display_params &P, int format, void *out)
{
    const int W = ctx.W, H = ctx.H, tile = ctx.tile > 0 ? ctx.tile : 32;
    const int tilesX = (W + tile - 1) / tile;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            const bool owned = (((y / tile) * tilesX + (x / tile)) % ctx.world) == ctx.rank;
            const float *px = &ctx.fb[4 * i];
            if (format == HR_DISPLAY_HDR_RGBA32F) {
                float *o = (float *)out + 4 * i;
                if (!owned || px[3] == 0.0f) {
                    o[0] = o[1] = o[2] = 0.0f, o[3] = owned ? px[3] : 0.0f;
                } else {
                    const float divisor = 1.0f / px[3];
                    o[0] = px[0] * divisor, o[1] = px[1] * divisor, o[2] = px[2] * divisor, o[3] = px[3];
                }
                continue;
            }
            float c[3] = {0.0f, 0.0f, 0.0f};
            if (owned) displayFragment(px, ((float)x + 0.5f) / (float)W, ((float)y + 0.5f) / (float)H, P, c);
            if (format == HR_DISPLAY_RGBA32F) {
                float *o = (float *)out + 4 * i;
                o[0] = c[0], o[1] = c[1], o[2] = c[2], o[3] = owned ? 1.0f : 0.0f;
            } else {
                uint32_t *o = (uint32_t *)out + i;
                *o = owned ? (toByte(c[0]) | (toByte(c[1]) << 8) | (toByte(c[2]) << 16) | 0xFF000000u) : 0u;
            }
        }
    }
}

} // namespace ora
