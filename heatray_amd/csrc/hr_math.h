// hr_math.h — device arithmetic of libhrcore (gfx950).
//
// The hot path's arithmetic contract: every operation is one IEEE-754
// binary32 operation in the order written, no FMA contraction (the library is built with
// -ffp-contract=off), correctly rounded sqrt / divide (hipcc default), GLSL min/max semantics,
// and sin / cos / atan / exp as the Cephes single-precision algorithms (S. Moshier) in plain
// float operations instead of the ocml versions, whose last-bit behaviour is not specified.
// This is what makes the HDR buffer reproducible bit for bit by an independent implementation.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define HRD __device__ __forceinline__

// Pointers that the kernels read out of structures in memory (scene block, step table, texture descriptors) are "generic" to
// the compiler: every access becomes a FLAT instruction, whose result can only be waited for with vmcnt(0) lgkmcnt(0) — a full
// drain of everything in flight.  All of these pointers are hipMalloc'ed, so device code reads them through address space 1
// (global_load / global_store, partial waits, SGPR-base addressing).  The host pass sees plain pointers.
#if defined(__HIP_DEVICE_COMPILE__)
#define HR_GLOBAL __attribute__((address_space(1)))
#else
#define HR_GLOBAL
#endif
template <class T> HRD const HR_GLOBAL T *G(const T *p) { return (const HR_GLOBAL T *)p; }
template <class T> HRD HR_GLOBAL T *G(T *p) { return (HR_GLOBAL T *)p; }

namespace hr {

struct v2 {
    float x, y;
};
struct v3 {
    float x, y, z;
    __host__ __device__ __forceinline__ v3() {}
    __host__ __device__ __forceinline__ v3(float a) : x(a), y(a), z(a) {}
    __host__ __device__ __forceinline__ v3(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct v4 {
    float x, y, z, w;
};

HRD v3 operator+(v3 a, v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HRD v3 operator-(v3 a, v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
HRD v3 operator*(v3 a, v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
HRD v3 operator/(v3 a, v3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
HRD v3 operator*(v3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
HRD v3 operator*(float s, v3 a) { return v3(s * a.x, s * a.y, s * a.z); }
HRD v3 operator/(v3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
HRD v3 operator-(v3 a) { return v3(-a.x, -a.y, -a.z); }

// GLSL: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x
HRD float fmin_(float x, float y) { return (y < x) ? y : x; }
HRD float fmax_(float x, float y) { return (x < y) ? y : x; }
HRD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
HRD float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
HRD v3 min3(v3 a, v3 b) { return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
HRD v3 max3(v3 a, v3 b) { return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }

HRD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HRD v3 cross(v3 a, v3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HRD float sqrt_(float x) { return __builtin_sqrtf(x); } // correctly rounded (v_sqrt_f32 + residual fix-up); __fsqrt_rn is NOT
HRD float inversesqrt(float x) { return 1.0f / sqrt_(x); }
HRD float length(v3 v) { return sqrt_(dot(v, v)); }
HRD v3 normalize(v3 v) { return v * inversesqrt(dot(v, v)); }
HRD float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
HRD v3 mix(v3 x, v3 y, v3 a) { return v3(mix(x.x, y.x, a.x), mix(x.y, y.y, a.y), mix(x.z, y.z, a.z)); }
HRD float floor_(float x) { return __builtin_floorf(x); }
HRD float fract(float x) { return x - floor_(x); }
HRD float abs_(float x) { return __builtin_fabsf(x); }
HRD float smoothstep(float e0, float e1, float x)
{
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
HRD v3 refract(v3 I, v3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f);
    return eta * I - (eta * d + sqrt_(k)) * N;
}

struct m3 {
    v3 c0, c1, c2;
};
HRD v3 mul(const m3 &m, v3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }
HRD v3 mulT(const m3 &m, v3 v) { return v3(dot(m.c0, v), dot(m.c1, v), dot(m.c2, v)); }

// column-major 4x4: point = M[0]*x + M[1]*y + M[2]*z + M[3]; vector = mat3(M) * v
HRD v3 xformPoint(const float *m, v3 p)
{
    return v3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
HRD v3 xformVector(const float *m, v3 v)
{
    return v3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
              m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// utility.rlsl:9-13 evaluated in float
#define HR_KPI 3.14159265359f
#define HR_KTWOPI (2.0f * HR_KPI)
#define HR_KONEOVERPI (1.0f / HR_KPI)

// ---- Cephes single-precision transcendental functions ----
HRD void sincos_(float xx, float *s, float *c)
{
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float x = abs_(xx);
    int sin_sign = (xx < 0.0f) ? -1 : 1;
    int cos_sign = 1;
    if (!(x <= 8192.0f)) {
        *s = 0.0f;
        *c = 1.0f;
        return;
    }
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        sin_sign = -sin_sign;
        cos_sign = -cos_sign;
        j -= 4;
    }
    if (j > 1) cos_sign = -cos_sign;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x;
    ps = ps + x;
    float sv, cv;
    if (j == 1 || j == 2) {
        sv = pc;
        cv = ps;
    } else {
        sv = ps;
        cv = pc;
    }
    *s = (sin_sign < 0) ? -sv : sv;
    *c = (cos_sign < 0) ? -cv : cv;
}
HRD float sin_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return s;
}
HRD float cos_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return c;
}

HRD float atan_(float xx)
{
    float x = abs_(xx);
    float y;
    if (x > 2.414213562373095f) {
        y = 1.5707963267948966192f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = 0.7853981633974483096f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return (xx < 0.0f) ? -y : y;
}
HRD float atan2_(float y, float x)
{
    const float PIF = 3.14159265358979323846f;
    const float PIO2F = 1.5707963267948966192f;
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2F;
        if (y < 0.0f) return -PIO2F;
        return 0.0f;
    }
    if (y == 0.0f) return (x < 0.0f) ? PIF : 0.0f;
    float w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -PIF : PIF;
    return w + atan_(y / x);
}
HRD float exp_(float xx)
{
    if (xx > 88.0f) return __builtin_inff();
    if (!(xx >= -87.0f)) return (xx != xx) ? xx : 0.0f;
    float x = xx;
    float z = floor_(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return z * __uint_as_float((uint32_t)(n + 127) << 23);
}

// Cephes logf (single precision), x > 0 and normal (same operation order as oracle/oracle_math.h)
HRD float log_(float xx)
{
    uint32_t bits = __float_as_uint(xx);
    int e = (int)((bits >> 23) & 0xFFu) - 126; // xx = m * 2^e with m in [0.5, 1)
    float x = __uint_as_float((bits & 0x807FFFFFu) | 0x3F000000u);
    if (x < 0.707106781186547524f) {
        e -= 1;
        x = x + x - 1.0f;
    } else {
        x = x - 1.0f;
    }
    float z = x * x;
    float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x + 1.4249322787e-1f) * x -
                   1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f) * x * z;
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = x + y;
    z = z + 0.693359375f * fe;
    return z;
}

// pow of the display pipeline: exp(y * log(x)) for x > 0, 0 otherwise (see oracle/oracle_math.h)
HRD float pow_(float x, float y)
{
    if (!(x > 1.17549435e-38f)) return 0.0f;
    return exp_(y * log_(x));
}

} // namespace hr
