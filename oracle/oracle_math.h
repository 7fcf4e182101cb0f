// oracle_math.h — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// The ARITHMETIC SPEC of the hot path.  The reference's
// arithmetic runs inside the closed OpenRL RLSL compiler, so operation order and
// transcendental precision are parity-unpinned.  This oracle fixes
// them: every operation is one IEEE-754 binary32 operation, evaluated in the
// order written here, with no FMA contraction (-ffp-contract=off); sqrt and
// divide are correctly rounded; sin/cos/atan/exp are the Cephes single-precision
// algorithms (S. Moshier, "Cephes Mathematical Library", sinf.c/atanf.c/expf.c)
// restated below in plain float operations.  The HIP kernels implement the same
// sequence of operations, which is what lets the parity tests demand bit-exact
// HDR buffers rather than a statistical tolerance.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace ora {

struct vec2 {
    float x, y;
};

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a) : x(a), y(a), z(a) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }

// GLSL min/max: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x.
inline float fmin_(float x, float y) { return (y < x) ? y : x; }
inline float fmax_(float x, float y) { return (x < y) ? y : x; }
inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
inline vec3 min3(vec3 a, vec3 b) { return vec3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
inline vec3 max3(vec3 a, vec3 b) { return vec3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }

// dot: ((x*x' + y*y') + z*z')
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b)
{
    return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
// normalize: v * (1/sqrt(dot(v,v)))
inline vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }
// GLSL mix: x*(1-a) + y*a
inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 mix(vec3 x, vec3 y, vec3 a) { return vec3(mix(x.x, y.x, a.x), mix(x.y, y.y, a.y), mix(x.z, y.z, a.z)); }
inline float fract(float x) { return x - floorf(x); }
inline float smoothstep(float e0, float e1, float x)
{
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
// GLSL refract
inline vec3 refract(vec3 I, vec3 N, float eta)
{
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return vec3(0.0f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}

// 3x3 matrix as three columns (GLSL mat3(X, N, Z)).
struct mat3 {
    vec3 c0, c1, c2;
};
// M * v = c0*v.x + c1*v.y + c2*v.z
inline vec3 mul(const mat3 &m, vec3 v) { return m.c0 * v.x + m.c1 * v.y + m.c2 * v.z; }
// transpose(M) * v = (dot(c0,v), dot(c1,v), dot(c2,v))  (utility.rlsl:146-151)
inline vec3 mulT(const mat3 &m, vec3 v) { return vec3(dot(m.c0, v), dot(m.c1, v), dot(m.c2, v)); }

// Column-major 4x4 (glm::mat4): m[c*4 + r].
// point:  M[0]*x + M[1]*y + M[2]*z + M[3]   (vertex.rlsl:27, perspective.rlsl:84)
inline vec3 xformPoint(const float *m, vec3 p)
{
    return vec3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
// vector: mat3(M) * v                         (vertex.rlsl:28-29, perspective.rlsl:85)
inline vec3 xformVector(const float *m, vec3 v)
{
    return vec3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
                m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// ---- constants of utility.rlsl:9-13, evaluated in float like RLSL globals ----
static const float kPI = 3.14159265359f;
static const float kTwoPI = 2.0f * kPI;
static const float kOneOverPI = 1.0f / kPI;
static const float kOneOverTwoPI = 1.0f / kTwoPI;
static const float kPIOverTwo = kPI * 0.5f;

// ---- transcendental functions (Cephes single precision, restated) ----

// sinf/cosf: octant reduction with the 3-part Cody-Waite constant for pi/4.
inline void sincos_(float xx, float *s, float *c)
{
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float x = fabsf(xx);
    int sin_sign = (xx < 0.0f) ? -1 : 1;
    int cos_sign = 1;
    if (!(x <= 8192.0f)) { // out of the supported range (also NaN): defined result
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
inline float sin_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return s;
}
inline float cos_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return c;
}

inline float atan_(float xx)
{
    float x = fabsf(xx);
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
// GLSL atan(y, x)
inline float atan2_(float y, float x)
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

inline float exp_(float xx)
{
    if (xx > 88.0f) return INFINITY;
    if (!(xx >= -87.0f)) return (xx != xx) ? xx : 0.0f;
    float x = xx;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    // ldexp by exponent-field construction; n in [-126, 127] after the range check above.
    uint32_t bits = (uint32_t)(n + 127) << 23;
    float p;
    std::memcpy(&p, &bits, 4);
    return z * p;
}

// Cephes logf (single precision), x > 0 and normal; restated in plain float operations like exp_ above.
inline float log_(float xx)
{
    uint32_t bits;
    std::memcpy(&bits, &xx, 4);
    int e = (int)((bits >> 23) & 0xFFu) - 126; // xx = m * 2^e with m in [0.5, 1)
    bits = (bits & 0x807FFFFFu) | 0x3F000000u;
    float x;
    std::memcpy(&x, &bits, 4);
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

// pow for the display pipeline (displayGL.frag: sRGB curves).  GLSL leaves pow's precision to the implementation; this
// build defines it as exp(y * log(x)) for x > 0 and 0 for x <= 0 (y > 0 in every use).
inline float pow_(float x, float y)
{
    if (!(x > 1.17549435e-38f)) return 0.0f;
    return exp_(y * log_(x));
}

} // namespace ora
