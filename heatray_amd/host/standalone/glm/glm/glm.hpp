// mini glm — STANDALONE BUILDS ONLY.
//
// The drop-in layer in heatray_amd/host/ is written against glm exactly like the classes it replaces; in the
// Heatray tree the real glm (3rdParty/glm) is on the include path and this directory is not.  This header
// provides the handful of glm types and functions the host layer and its tests use, so that the layer can be
// compiled and tested without any third-party checkout (no network in the build image).  Column-major like glm.
#pragma once

#include <cmath>
#include <cstddef>

namespace glm {

template <typename T> struct tvec2 {
    T x{}, y{};
    constexpr tvec2() = default;
    constexpr explicit tvec2(T s) : x(s), y(s) {}
    constexpr tvec2(T a, T b) : x(a), y(b) {}
    T &operator[](int i) { return (&x)[i]; }
    const T &operator[](int i) const { return (&x)[i]; }
};
using vec2 = tvec2<float>;
using ivec2 = tvec2<int>;

struct vec3 {
    float x{}, y{}, z{};
    constexpr vec3() = default;
    constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
    constexpr vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x{}, y{}, z{}, w{};
    constexpr vec4() = default;
    constexpr vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    constexpr vec4(const vec3 &v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
    vec3 xyz() const { return vec3(x, y, z); }
};

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(vec3 a, vec3 b) { return !(a == b); }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 min(vec3 a, vec3 b) { return vec3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z); }
inline vec3 max(vec3 a, vec3 b) { return vec3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z); }
template <typename T> inline T max(T a, T b) { return a > b ? a : b; }
template <typename T> inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> inline T abs(T a) { return a < T(0) ? -a : a; }
enum qualifier { defaultp };
template <typename T, qualifier Q = defaultp> inline T saturate(T x) { return x < T(0) ? T(0) : (x > T(1) ? T(1) : x); }
inline vec3 saturate(vec3 v) { return vec3(saturate<float>(v.x), saturate<float>(v.y), saturate<float>(v.z)); }
template <typename T> constexpr T pi() { return T(3.14159265358979323846264338327950288); }
template <typename T> constexpr T half_pi() { return T(1.57079632679489661923132169163975144); }
template <typename T> constexpr T two_pi() { return T(6.28318530717958647692528676655900576); }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

struct mat4 {
    vec4 c[4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) { c[0] = vec4(d, 0, 0, 0), c[1] = vec4(0, d, 0, 0), c[2] = vec4(0, 0, d, 0), c[3] = vec4(0, 0, 0, d); }
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
};
using mat4x4 = mat4;
inline vec4 operator*(const mat4 &m, const vec4 &v)
{
    vec4 r;
    for (int k = 0; k < 4; ++k) r[k] = m[0][k] * v.x + m[1][k] * v.y + m[2][k] * v.z + m[3][k] * v.w;
    return r;
}
inline mat4 operator*(const mat4 &a, const mat4 &b)
{
    mat4 r(0.0f);
    for (int j = 0; j < 4; ++j) r[j] = a * b[j];
    return r;
}
inline mat4 translate(const mat4 &m, const vec3 &v)
{
    mat4 r = m;
    r[3] = m * vec4(v, 1.0f);
    return r;
}
inline float determinant(const mat4 &m)
{
    const float a0 = m[0][0] * m[1][1] - m[1][0] * m[0][1], a1 = m[0][0] * m[2][1] - m[2][0] * m[0][1], a2 = m[0][0] * m[3][1] - m[3][0] * m[0][1];
    const float a3 = m[1][0] * m[2][1] - m[2][0] * m[1][1], a4 = m[1][0] * m[3][1] - m[3][0] * m[1][1], a5 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    const float b0 = m[0][2] * m[1][3] - m[1][2] * m[0][3], b1 = m[0][2] * m[2][3] - m[2][2] * m[0][3], b2 = m[0][2] * m[3][3] - m[3][2] * m[0][3];
    const float b3 = m[1][2] * m[2][3] - m[2][2] * m[1][3], b4 = m[1][2] * m[3][3] - m[3][2] * m[1][3], b5 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    return a0 * b5 - a1 * b4 + a2 * b3 + a3 * b2 - a4 * b1 + a5 * b0;
}

struct quat {
    float w{1}, x{}, y{}, z{};
};
inline quat angleAxis(float angle, const vec3 &axis)
{
    const float s = std::sin(angle * 0.5f);
    return quat{std::cos(angle * 0.5f), axis.x * s, axis.y * s, axis.z * s};
}
inline quat operator*(const quat &p, const quat &q)
{
    return quat{p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z, p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y,
                p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z, p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x};
}
inline quat inverse(const quat &q)
{
    const float n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return quat{q.w / n, -q.x / n, -q.y / n, -q.z / n};
}
inline mat4 mat4_cast(const quat &q)
{
    mat4 m(1.0f);
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    m[0][0] = 1 - 2 * (yy + zz), m[0][1] = 2 * (xy + wz), m[0][2] = 2 * (xz - wy);
    m[1][0] = 2 * (xy - wz), m[1][1] = 1 - 2 * (xx + zz), m[1][2] = 2 * (yz + wx);
    m[2][0] = 2 * (xz + wy), m[2][1] = 2 * (yz - wx), m[2][2] = 1 - 2 * (xx + yy);
    return m;
}

} // namespace glm
