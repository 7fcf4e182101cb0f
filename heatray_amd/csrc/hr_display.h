// hr_display.h — the display resolve: Resources/shaders/displayGL.frag:74-151 per pixel of the
// accumulation buffer, on the device.  Same operation order as oracle/oracle_display.cpp (arithmetic contract).
#pragma once

#include "hr_math.h"

namespace hr {

HRD float linearToSRGB1(float c) // displayGL.frag:47-58
{
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * pow_(c, 1.0f / 2.4f) - 0.055f;
}
HRD float srgbToLinear1(float c) // displayGL.frag:60-72
{
    if (c <= 0.04045f) return c / 12.92f;
    return pow_((c + 0.055f) / (1.0f + 0.055f), 2.4f);
}
HRD float rrtAndOdtFit1(float v) // displayGL.frag:40-45
{
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}
HRD float step_(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

// one fragment: rgba = accumulated sample sum and sample count, (u, v) = textureCoords (pixel centre)
HRD void displayFragment(const float4 rgba, float u, float v, const hr_display_params &P, float out[3])
{
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (rgba.w != 0.0f) r = rgba.x / rgba.w, g = rgba.y / rgba.w, b = rgba.z / rgba.w; // :78 (no samples yet: black)
    if (P.tonemapping_enabled == 1) { // :82-90 ACES, matrices as column sums
        r = linearToSRGB1(r), g = linearToSRGB1(g), b = linearToSRGB1(b);
        float x = (0.59719f * r + 0.35458f * g) + 0.04823f * b;
        float y = (0.07600f * r + 0.90834f * g) + 0.01566f * b;
        float z = (0.02840f * r + 0.13383f * g) + 0.83777f * b;
        x = rrtAndOdtFit1(x), y = rrtAndOdtFit1(y), z = rrtAndOdtFit1(z);
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
        float hx = abs_(qz + (qw - qy) / (6.0f * d + e)), hy = d / (qx + e), hz = qx;
        hx = hx * P.hue;
        hy = hy * P.saturation;
        const float mapped = sqrt_(hy) * P.vibrance;
        hy = hy * (1.0f + mapped);
        const float k1 = 1.0f, k2 = 2.0f / 3.0f, k3 = 1.0f / 3.0f, k4 = 3.0f;
        const float p1 = abs_(fract(hx + k1) * 6.0f - k4), p2 = abs_(fract(hx + k2) * 6.0f - k4), p3 = abs_(fract(hx + k3) * 6.0f - k4);
        r = hz * mix(k1, clamp_(p1 - k1, 0.0f, 1.0f), hy);
        g = hz * mix(k1, clamp_(p2 - k1, 0.0f, 1.0f), hy);
        b = hz * mix(k1, clamp_(p3 - k1, 0.0f, 1.0f), hy);
    }
    r = r * P.red, g = g * P.green, b = b * P.blue; // :132-136
    { // :139-143 vignette (the "+ blue" is the reference's)
        const float dx = 0.5f - u, dy = 0.5f - v;
        const float dist = sqrt_(dx * dx + dy * dy);
        const float vig = smoothstep(0.8f, P.vignette_falloff * 0.799f, dist * (P.vignette_intensity + P.blue));
        r = r * vig, g = g * vig, b = b * vig;
    }
    r = r * P.camera_exposure, g = g * P.camera_exposure, b = b * P.camera_exposure; // :146
    out[0] = linearToSRGB1(r), out[1] = linearToSRGB1(g), out[2] = linearToSRGB1(b); // :149
}

HRD uint32_t toByte(float c)
{
    if (!(c == c)) return 0u; // NaN
    return (uint32_t)floor_(clamp_(c, 0.0f, 1.0f) * 255.0f + 0.5f);
}

} // namespace hr
