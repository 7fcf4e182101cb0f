// hr_shade.h — device shading: the RLSL programs OpenRL runs per ray, as inline HIP functions.
//
// Each function cites the shader text it replaces (paths relative to
// /root/reference/Resources/shaders).  Called by the SoA shade kernel (hr_kernels.hip), one lane per
// path; the emitted rays leave through wave-compacted queues instead of OpenRL's emitRay().
#pragma once

#include "hr_texture.h"
#include "hr_types.h"

namespace hr {

enum { LIGHT_TYPE_DIRECTIONAL = 1, LIGHT_TYPE_POINT = 2, LIGHT_TYPE_SPOT = 3, LIGHT_TYPE_ENVIRONMENT = 4 }; // lightDefines.rlsl:13-16

// rayAttributes.rlsl:8-11 + the built-in ray fields the shaders touch
struct Ray {
    v3 o, d;
    float maxT;
    v3 weight;
    int sequenceID, sequenceIndexOffset;
    float extraT;
    int depth;
    bool occlusionTest;
    int missKind, missIdx;
    bool valid;
    float coneW, coneG; // ray cone (HR_TEXTURE_LOD_CONE): width at the origin and spread angle; travels as two bf16-truncated floats
};

// The cone rides in one dword of the ray record: the upper halves of the two floats (truncation is part of the contract)
HRD uint32_t packCone(float w, float g) { return (__float_as_uint(w) & 0xFFFF0000u) | (__float_as_uint(g) >> 16); }
HRD void unpackCone(uint32_t bits, float &w, float &g) { w = __uint_as_float(bits & 0xFFFF0000u), g = __uint_as_float(bits << 16); }
HRD float widenCone(float g, float roughness) { return fmin_(g + 0.25f * roughness, 1.0f); }

// An additional occlusion ray of HR_ESTIMATOR_ALL_LIGHTS as the shading kernel keeps it until the queue append: it starts at the hit
// point like the vertex's first one, so direction, range and the value its light shader adds are all there is to it (a third of
// the registers of a Ray; the value is evaluated where the ray is made)
struct ExtraRay {
    v3 d, value;
    float maxT;
    bool valid;
};

// LOD: compiled with the ray-cone texture lookups of HR_TEXTURE_LOD_CONE.  The shading kernel exists in both variants; the one
// without is what runs until a pass asks for the mode (its code and register allocation are those of the level-0 sampler alone).
template <int MODE> struct ShaderT {
    static constexpr bool LOD = (MODE & 1) != 0; // HR_TEXTURE_LOD_CONE compiled in
    static constexpr bool ALL = (MODE & 2) != 0; // HR_ESTIMATOR_ALL_LIGHTS compiled in (a second occlusion ray per vertex)
    const SceneDev &S;
    const hr_pass_params &pp;
    HR_GLOBAL float *px; // RGBA of the pixel this path belongs to (single owner: plain read-modify-write)
    uint32_t nAccum;
    float lodBase; // HR_TEXTURE_LOD_CONE: log2(texels of a unit-uv-square texture under the cone's footprint); level 0 when <= 0

    HRD ShaderT(const SceneDev &s, const hr_pass_params &p, HR_GLOBAL float *pixel) : S(s), pp(p), px(pixel), nAccum(0), lodBase(-1e30f) {}

    // Advance the ray's cone to the hit and derive the footprint's level offset on this triangle (HR_TEXTURE_LOD_CONE)
    HRD void setFootprint(Ray &in, v3 normal, float t, uint32_t prim)
    {
        lodBase = -1e30f;
        if (!LOD) return;
        const float hitW = in.coneW + in.coneG * t;
        in.coneW = hitW;
        if (pp.texture_lod != HR_TEXTURE_LOD_CONE || !S.texDensity || !(hitW > 0.0f)) return;
        const float cosT = fmax_(abs_(dot(in.d, normal)), 0.1f);
        lodBase = G(S.texDensity)[prim] + log_(hitW / cosT) * 1.4426950408889634f;
    }

    // sequence.rlsl:18-28
    HRD v2 getSequenceValue(int sequenceIndex, int sampleIndex) const
    {
        int ws = sequenceIndex % S.nSeq;
        int wv = sampleIndex % S.seqLen;
        float2 f = G(S.seq)[(size_t)ws * S.seqLen + wv];
        return v2{f.x, f.y};
    }

    // accumulator.rlsl:12-28: the value accumulate() receives
    HRD v3 accumulateValue(v3 color) const
    {
        v3 value(0.0f);
        if (pp.enable_accumulator_visualizer == 1) {
            if (pp.show_nans == 1) {
                bool any = (color.x != color.x) || (color.y != color.y) || (color.z != color.z);
                value = any ? v3(100.0f) : (min3(color, v3(1.0f)) * 0.1f);
            } else if (pp.show_inf == 1) {
                bool any = __builtin_isinf(color.x) || __builtin_isinf(color.y) || __builtin_isinf(color.z);
                value = any ? v3(100.0f) : (min3(color, v3(1.0f)) * 0.1f);
            }
        } else {
            value = min3(v3(pp.max_channel_value), color);
        }
        return value;
    }
    HRD void accumulate3(v3 c)
    {
        px[0] = px[0] + c.x;
        px[1] = px[1] + c.y;
        px[2] = px[2] + c.z;
        ++nAccum;
    }
    HRD void accumulate4(v3 c, float a)
    {
        accumulate3(c);
        px[3] = px[3] + a;
    }
    HRD void performAccumulate(v3 color) { accumulate3(accumulateValue(color)); }

    HRD v4 tex(int id, v2 uv) const
    {
        if (id < 0 || id >= S.nTextures) return v4{1.0f, 1.0f, 1.0f, 1.0f}; // dummy white texel (Texture.h:188-203)
        const auto &t = G(S.textures)[id];
        if (!t.px) return v4{1.0f, 1.0f, 1.0f, 1.0f};
        if (LOD && pp.texture_lod == HR_TEXTURE_LOD_CONE) return sampleTextureLod(t, uv.x, uv.y, lodBase + t.lodScale);
        return sampleTexture(t, uv.x, uv.y);
    }

    // environmentLight.rlsl:19-34 — radiance the environment shader passes to performAccumulate
    HRD v3 environmentRadiance(v3 dir, v3 weight) const
    {
        float theta = atan2_(dir.x, -dir.z) + S.lights.env_theta_rotation;
        if (theta > HR_KTWOPI) theta = theta - HR_KTWOPI;
        float phi = atan2_(dir.y, sqrt_(dir.x * dir.x + dir.z * dir.z));
        float u = (theta / HR_KTWOPI) + 0.5f;
        float v = (-phi * HR_KONEOVERPI) + 0.5f;
        v4 t = v4{0.0f, 0.0f, 0.0f, 0.0f};
        int id = S.lights.env_texture;
        if (id >= 0 && id < S.nTextures && G(S.textures)[id].px) t = sampleTexture(G(S.textures)[id], u, 1.0f - v);
        v3 sample = v3(t.x, t.y, t.z) * S.lights.env_exposure;
        return weight * sample;
    }
    // directionalLight.rlsl:20-26, pointLight.rlsl:20-29, spotLight.rlsl:20-36, environmentLight.rlsl.
    // Evaluated when the occlusion ray is created (every input is known then); the trace kernel adds
    // `value` only if the ray turns out unoccluded.  Returns false when the light shader accumulates nothing.
    HRD bool lightShaderValue(const Ray &r, v3 &value) const
    {
        const hr_lights &L = S.lights;
        switch (r.missKind) {
        case MISS_ENV:
            value = accumulateValue(environmentRadiance(r.d, r.weight));
            return true;
        case MISS_DIR: {
            const float *c = L.directional_colors[r.missIdx];
            value = accumulateValue(r.weight * v3(c[0], c[1], c[2]));
            return true;
        }
        case MISS_POINT: {
            float s = r.maxT + r.extraT;
            float attenuation = 1.0f / (s * s);
            const float *c = L.point_colors[r.missIdx];
            value = accumulateValue(r.weight * v3(c[0], c[1], c[2]) * attenuation);
            return true;
        }
        case MISS_SPOT: {
            const float *sd = L.spot_directions[r.missIdx];
            float rayAngle = dot(-r.d, v3(sd[0], sd[1], sd[2]));
            if (rayAngle >= 0.0f) {
                float s = r.maxT + r.extraT;
                float attenuation = 1.0f / (s * s);
                const float *c = L.spot_colors[r.missIdx];
                v3 result = r.weight * v3(c[0], c[1], c[2]) * attenuation;
                result = result * (1.0f - smoothstep(L.spot_angles[r.missIdx][0], L.spot_angles[r.missIdx][1], rayAngle));
                value = accumulateValue(result);
                return true;
            }
            return false;
        }
        default:
            return false;
        }
    }

    // ---- utility.rlsl ----
    static HRD float square(float x) { return x * x; }
    static HRD float getSign(float x) { return x < 0.0f ? -1.0f : 1.0f; }             // :35-38
    static HRD float pow5(float x) { return x * square(x) * square(x); }              // :141-144
    static HRD float greaterThanZero(float f) { return fmax_(1e-5f, f); }             // :153-156
    static HRD float luminosity(v3 c) { return dot(c, v3(0.33f, 0.59f, 0.11f)); }     // :163-166
    static HRD m3 orthonormalFrame(v3 N)                                              // :43-60
    {
        v3 lh(N.x, N.z, N.y);
        float s = getSign(lh.z);
        float a = -1.0f / (s + lh.z);
        float b = lh.x * lh.y * a;
        v3 X(1.0f + s * lh.x * lh.x * a, s * b, -s * lh.x);
        v3 Z(b, s + lh.y * lh.y * a, -lh.y);
        m3 m;
        m.c0 = v3(X.x, X.z, X.y);
        m.c1 = N;
        m.c2 = v3(Z.x, Z.z, Z.y);
        return m;
    }
    static HRD v3 cosineWeightedSample(float u1, float u2) // :64-75
    {
        float theta = sqrt_(u1);
        float phi = HR_KTWOPI * u2;
        float s, c;
        sincos_(phi, &s, &c);
        float x = theta * c;
        float y = sqrt_(fmax_(0.0f, 1.0f - u1));
        float z = theta * s;
        return normalize(v3(x, y, z));
    }
    static HRD v3 sampleVisibleGGX(v3 localSpaceV, float u1, float u2, float roughnessAlpha) // :109-139
    {
        v3 zUpV(localSpaceV.x, localSpaceV.z, localSpaceV.y);
        v3 Vh = normalize(v3(zUpV.x * roughnessAlpha, zUpV.y * roughnessAlpha, zUpV.z));
        float lengthSquared = (Vh.x * Vh.x) + (Vh.y * Vh.y);
        v3 T1 = (lengthSquared > 0.0f) ? v3(-Vh.y, Vh.x, 0.0f) * inversesqrt(lengthSquared) : v3(1.0f, 0.0f, 0.0f);
        v3 T2 = cross(Vh, T1);
        float r = sqrt_(u1);
        float phi = HR_KTWOPI * u2;
        float sn, cs;
        sincos_(phi, &sn, &cs);
        float t1 = r * cs;
        float t2 = r * sn;
        float s = 0.5f * (1.0f + Vh.z);
        float t1Squared = square(t1);
        t2 = (1.0f - s) * sqrt_(1.0f - t1Squared) + (s * t2);
        v3 Nh = (t1 * T1) + (t2 * T2) + sqrt_(fmax_(0.0f, 1.0f - t1Squared - square(t2))) * Vh;
        v3 zUp = normalize(v3(roughnessAlpha * Nh.x, roughnessAlpha * Nh.y, fmax_(0.0f, Nh.z)));
        return v3(zUp.x, zUp.z, zUp.y);
    }

    // ---- brdfs.rlsl ----
    static HRD v3 F_Schlick(v3 Cspec, float cosTheta) { return Cspec + (v3(1.0f) - Cspec) * pow5(1.0f - cosTheta); } // :46-50
    static HRD float F_Schlick(float f0, float cosTheta) { return f0 + (1.0f - f0) * pow5(1.0f - cosTheta); }         // :53-57
    static HRD float F_Fresnel(float eta, float cosThetaI)                                                           // :59-71
    {
        float sinThetaT2 = square(eta) * (1.0f - square(cosThetaI));
        if (sinThetaT2 < 1.0f) {
            float cosThetaT = sqrt_(1.0f - sinThetaT2);
            float perpendicular = square((eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT));
            float parallel = square((cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT));
            return 0.5f * (perpendicular + parallel);
        }
        return 1.0f;
    }
    static HRD float D_GGX(float NdotH, float roughnessAlpha) // :73-78
    {
        float alpha2 = square(roughnessAlpha);
        float denominator = square(square(NdotH) * (alpha2 - 1.0f) + 1.0f);
        return HR_KONEOVERPI * (alpha2 / greaterThanZero(denominator));
    }
    static HRD float G1_Smith_GGX(float NdotI, float roughnessAlpha) // :88-93
    {
        float alpha2 = square(roughnessAlpha);
        float denom = sqrt_(alpha2 + (1.0f - alpha2) * greaterThanZero(square(NdotI))) + NdotI;
        return (2.0f * NdotI) / greaterThanZero(denom);
    }
    static HRD float G2_Smith_GGX(float NdotO, float NdotI, float roughnessAlpha) // :95-98
    {
        return G1_Smith_GGX(NdotO, roughnessAlpha) * G1_Smith_GGX(NdotI, roughnessAlpha);
    }

    // ---- lightSampling.rlsl:11-161 ----
    struct LightSample {
        v3 dir;
        int missKind, missIdx;
        float probability;
        float maxDistance;
        int type;
    };
    // withoutEnv: the pick among the analytic lights only (HR_ESTIMATOR_ALL_LIGHTS); type ENVIRONMENT with probability 0 if there is none
    HRD LightSample computeLightSample(v3 N, float lightProbability, v3 P, bool withoutEnv = false) const
    {
        const hr_lights &L = S.lights;
        LightSample out;
        out.dir = v3(0.0f);
        out.missKind = MISS_NONE, out.missIdx = 0;
        out.probability = 0.0f;
        out.maxDistance = __builtin_inff();
        out.type = 0;
        float probabilitySum = 0.0f;
        float directional[HR_MAX_DIRECTIONAL_LIGHTS];
        float point[HR_MAX_POINT_LIGHTS];
        float spot[HR_MAX_SPOT_LIGHTS];
#pragma unroll
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) {
            directional[i] = 0.0f;
            if (i < L.n_directional) {
                const float *d = L.directional_directions[i], *c = L.directional_colors[i];
                directional[i] = saturate(dot(N, v3(d[0], d[1], d[2]))) * luminosity(v3(c[0], c[1], c[2]));
                probabilitySum += directional[i];
            }
        }
#pragma unroll
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) {
            point[i] = 0.0f;
            if (i < L.n_point) {
                const float *p = L.point_positions[i], *c = L.point_colors[i];
                v3 dir = normalize(v3(p[0], p[1], p[2]) - P);
                point[i] = saturate(dot(N, dir)) * luminosity(v3(c[0], c[1], c[2]));
                probabilitySum += point[i];
            }
        }
#pragma unroll
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) {
            spot[i] = 0.0f;
            if (i < L.n_spot) {
                const float *p = L.spot_positions[i], *c = L.spot_colors[i], *sd = L.spot_directions[i];
                v3 dir = normalize(v3(p[0], p[1], p[2]) - P);
                float rayAngle = dot(v3(sd[0], sd[1], sd[2]), -dir);
                spot[i] = saturate(dot(N, dir)) * luminosity(v3(c[0], c[1], c[2])) * ((rayAngle > 0.0f) ? 1.0f : 0.0f) *
                          ((rayAngle < L.spot_angles[i][1]) ? 0.0f : 1.0f) *
                          (1.0f - smoothstep(L.spot_angles[i][0], L.spot_angles[i][1], rayAngle));
                probabilitySum += spot[i];
            }
        }
        float environment = 0.0f;
        if (L.env_enabled && !withoutEnv) {
            // lightSampling.rlsl:74-79's constant 50, or for HR_ESTIMATOR_ENV_MIS the irradiance the map can deliver (pi x mean luminosity)
            environment = (pp.estimator != HR_ESTIMATOR_REFERENCE && S.envW > 0) ? (S.envMeanLum * HR_KPI) * L.env_exposure : 50.0f * L.env_exposure;
            probabilitySum += environment;
        }
        float norm = 1.0f / greaterThanZero(probabilitySum);
        environment *= norm;
#pragma unroll
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) directional[i] *= norm;
#pragma unroll
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) point[i] *= norm;
#pragma unroll
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) spot[i] *= norm;

        float currentProbability = 0.0f;
#pragma unroll
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) {
            if (i < L.n_directional) {
                currentProbability += directional[i];
                if (directional[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *d = L.directional_directions[i];
                    out.dir = v3(d[0], d[1], d[2]);
                    out.missKind = MISS_DIR, out.missIdx = i;
                    out.probability = directional[i];
                    out.type = LIGHT_TYPE_DIRECTIONAL;
                    return out;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) {
            if (i < L.n_point) {
                currentProbability += point[i];
                if (point[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *p = L.point_positions[i];
                    out.dir = normalize(v3(p[0], p[1], p[2]) - P);
                    out.missKind = MISS_POINT, out.missIdx = i;
                    out.probability = point[i];
                    out.maxDistance = length(v3(p[0], p[1], p[2]) - P);
                    out.type = LIGHT_TYPE_POINT;
                    return out;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) {
            if (i < L.n_spot) {
                currentProbability += spot[i];
                if (spot[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *p = L.spot_positions[i];
                    out.dir = normalize(v3(p[0], p[1], p[2]) - P);
                    out.missKind = MISS_SPOT, out.missIdx = i;
                    out.probability = spot[i];
                    out.maxDistance = length(v3(p[0], p[1], p[2]) - P);
                    out.type = LIGHT_TYPE_SPOT;
                    return out;
                }
            }
        }
        out.type = LIGHT_TYPE_ENVIRONMENT;
        out.probability = environment;
        return out;
    }

    // createRay(): the child inherits every attribute of rl_InRay, starts at the hit point, depth + 1
    static HRD Ray createRay(const Ray &in, v3 P)
    {
        Ray r = in;
        r.o = P;
        r.depth = in.depth + 1;
        r.valid = true;
        return r;
    }
    // emitRay(): a continuation ray travels whole; of an occlusion ray only what its queue entry needs is kept — direction, range and the
    // value its light's shader adds when nothing is in the way, evaluated right here (an occlusion ray starts at the hit point like every ray
    // a vertex emits): a third of the registers of a Ray across the rest of the shader
    HRD void emit(const Ray &r, ExtraRay &nee, Ray &next) const
    {
        if (r.occlusionTest)
            keep(r, nee);
        else
            next = r;
    }

    // ---- microfacet.rlsl ----
    HRD v3 computeMultiscattering(int lut, v3 Cspec, float NdotI, float roughness) const // :17-23
    {
        float ms = 0.0f;
        if (lut >= 0 && lut < S.nTextures && G(S.textures)[lut].px) ms = sampleTexture(G(S.textures)[lut], NdotI, roughness).x;
        return v3(1.0f) + Cspec * ms;
    }
    HRD void indirectDiffuseSample(const Ray &in, v3 P, v3 N, v3 Cdiff, float sampleProbability, float optionalLightSampleProbability, v2 rand,
                                   const m3 &frame, int missKind, ExtraRay &nee, Ray &next) const // :25-50
    {
        v3 dir = cosineWeightedSample(rand.x, rand.y);
        v3 O = mul(frame, dir);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            v3 reflectance = Cdiff;
            reflectance = reflectance * in.weight;
            reflectance = reflectance / sampleProbability;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                if (LOD) r.coneG = widenCone(in.coneG, 1.0f);
                if (missKind == MISS_ENV && !S.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    // ---- HR_ESTIMATOR_ENV_MIS (include/hrcore.h): importance sampling of the environment map + one-sample MIS.  Not in the
    // reference; the arithmetic is the oracle's (oracle/oracle_shade.cpp, same operations in the same order).
    HRD bool envMis() const { return pp.estimator != HR_ESTIMATOR_REFERENCE && S.envW > 0; }
    HRD bool allLights() const { return ALL && pp.estimator == HR_ESTIMATOR_ALL_LIGHTS; }
    // evaluate an additional occlusion ray's light shader now and keep what the queue append needs
    HRD void keep(const Ray &r, ExtraRay &x) const
    {
        v3 value(0.0f);
        x.valid = r.valid && lightShaderValue(r, value);
        x.d = r.d, x.maxT = r.maxT, x.value = value;
    }
    static constexpr int kPrimaryEnvSamples = 3; // environment samples HR_ESTIMATOR_ALL_LIGHTS takes at a camera ray's hit
    HRD void envTexelOf(v3 dir, int &i, int &j) const
    {
        float theta = atan2_(dir.x, -dir.z) + S.lights.env_theta_rotation;
        if (theta > HR_KTWOPI) theta = theta - HR_KTWOPI;
        float phi = atan2_(dir.y, sqrt_(dir.x * dir.x + dir.z * dir.z));
        float u = (theta / HR_KTWOPI) + 0.5f;
        float t = 1.0f - ((-phi * HR_KONEOVERPI) + 0.5f);
        const int w = S.envW, h = S.envH;
        int ii = (int)floor_(u * (float)w) % w;
        i = ii < 0 ? ii + w : ii;
        int jj = (int)floor_(t * (float)h);
        j = jj < 0 ? 0 : (jj >= h ? h - 1 : jj);
    }
    HRD float envPdf(v3 dir) const
    {
        int i, j;
        envTexelOf(dir, i, j);
        const float cosEl = sqrt_(dir.x * dir.x + dir.z * dir.z);
        const float K = ((float)S.envW * (float)S.envH) / (HR_KTWOPI * HR_KPI);
        return (G(S.envProb)[(size_t)j * S.envW + i] * K) / fmax_(cosEl, 1e-6f);
    }
    // largest index in [lo, hi] whose CDF value is <= x (cdf[lo] <= x is given)
    static HRD int cdfFind(const HR_GLOBAL float *cdf, int lo, int hi, float x)
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
    HRD v3 sampleEnv(float u1, float u2) const
    {
        const int w = S.envW, h = S.envH;
        const HR_GLOBAL float *rc = G(S.envRowCdf);
        // both searches start from a guide table (same result as a search over the whole table, a fifth of the dependent loads)
        int kr = (int)(u1 * (float)kEnvRowGuide);
        kr = kr < 0 ? 0 : (kr > kEnvRowGuide - 1 ? kEnvRowGuide - 1 : kr);
        const int j = cdfFind(rc, (int)G(S.envRowGuide)[kr], (int)G(S.envRowGuide)[kr + 1], u1);
        const float fy = (u1 - rc[j]) / fmax_(rc[j + 1] - rc[j], 1e-20f);
        const HR_GLOBAL float *cc = G(S.envColCdf) + (size_t)j * (w + 1);
        int kc = (int)(u2 * (float)kEnvColGuide);
        kc = kc < 0 ? 0 : (kc > kEnvColGuide - 1 ? kEnvColGuide - 1 : kc);
        const HR_GLOBAL uint16_t *cg = G(S.envColGuide) + (size_t)j * (kEnvColGuide + 1);
        const int i = cdfFind(cc, (int)cg[kc], (int)cg[kc + 1], u2);
        const float fx = (u2 - cc[i]) / fmax_(cc[i + 1] - cc[i], 1e-20f);
        const float t = ((float)j + saturate(fy)) / (float)h, u = ((float)i + saturate(fx)) / (float)w;
        const float elevation = (t - 0.5f) * HR_KPI;
        const float azimuth = (u - 0.5f) * HR_KTWOPI - S.lights.env_theta_rotation;
        float se, ce, sa, ca;
        sincos_(elevation, &se, &ce);
        sincos_(azimuth, &sa, &ca);
        return v3(ce * sa, se, -(ce * ca));
    }
    // which: 0 = the vertex's environment sample; 1, 2 = the extra ones HR_ESTIMATOR_ALL_LIGHTS takes at a camera ray's hit (own sequence values)
    HRD void envMisDiffuse(const Ray &in, v3 P, v3 N, v3 Cdiff, float sampleProbability, float envProbability, v2 rand, const m3 &frame, ExtraRay &nee,
                           Ray &next, int which = 0) const
    {
        const v2 sel = getSequenceValue(in.sequenceID + in.depth + 5 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // the map's row variable is stratified over the vertex's samples (sample j of n draws it from the j-th n-th of [0, 1): the
        // map's bright regions each get their sample), everything else comes from the sample's own sequence values
        const float envU1 = (rand.x + (float)which) / envProbability;
        if (which > 0) rand = getSequenceValue(in.sequenceID + in.depth + 4 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // how often the lobe's own sampler is used instead of the map's: half the time (ENV_MIS), an eighth with ALL_LIGHTS — a
        // cosine lobe rarely finds a small bright source, and every such sample is one the source does not get
        const float cLobe = (pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) ? 0.125f : 0.5f, cMap = 1.0f - cLobe;
        v3 O = (sel.x < cLobe) ? mul(frame, cosineWeightedSample(rand.x, rand.y)) : sampleEnv((pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) ? envU1 : rand.x, rand.y);
        float NdotO = dot(N, O);
        if (!(NdotO > 0.0f)) return;
        NdotO = saturate(NdotO);
        const float pLobe = NdotO / HR_KPI, pMap = envPdf(O);
        v3 reflectance = (Cdiff / HR_KPI) * NdotO;
        reflectance = reflectance * in.weight;
        reflectance = reflectance / (cLobe * pLobe + cMap * pMap);
        reflectance = reflectance / sampleProbability;
        reflectance = reflectance / envProbability;
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }
    HRD void envMisSpecular(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 Cspec, float roughnessAlpha, int lut, float roughness,
                            float sampleProbability, float envProbability, v2 rand, const m3 &frame, ExtraRay &nee, Ray &next, int which = 0) const
    {
        const v2 sel = getSequenceValue(in.sequenceID + in.depth + 5 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // the map's row variable is stratified over the vertex's samples (sample j of n draws it from the j-th n-th of [0, 1): the
        // map's bright regions each get their sample), everything else comes from the sample's own sequence values
        const float envU1 = (rand.x + (float)which) / envProbability;
        if (which > 0) rand = getSequenceValue(in.sequenceID + in.depth + 4 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        v3 O, H;
        if (sel.x < 0.5f) {
            H = mul(frame, sampleVisibleGGX(mulT(frame, I), rand.x, rand.y, roughnessAlpha));
            O = normalize(2.0f * saturate(dot(I, H)) * H - I);
        } else {
            O = sampleEnv((pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) ? envU1 : rand.x, rand.y);
            H = normalize(I + O);
        }
        float NdotO = dot(N, O);
        if (!(NdotO > 0.0f)) return;
        NdotO = saturate(NdotO);
        const float NdotH = saturate(dot(N, H)), IdotH = saturate(dot(I, H));
        const float D = D_GGX(NdotH, roughnessAlpha);
        const v3 F = F_Schlick(Cspec, IdotH);
        const float G2 = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha), G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
        v3 specular = (D * F * G2) / greaterThanZero(4.0f * NdotI);
        specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
        const float pLobe = (D * G1) / greaterThanZero(4.0f * NdotI), pMap = envPdf(O);
        v3 reflectance = specular * in.weight;
        reflectance = reflectance / greaterThanZero(0.5f * pLobe + 0.5f * pMap);
        reflectance = reflectance / sampleProbability;
        reflectance = reflectance / envProbability;
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }

    // (`ls` = computeLightSample(N, lightProbability, P), lightSampling.rlsl:11-161: evaluated by the caller, once for whichever lobe the
    // vertex samples, so that the light pick — a loop over up to fifteen lights — exists and runs once per wave, not once per lobe)
    HRD void directDiffuseSample(const Ray &in, v3 P, v3 N, v3 Cdiff, float sampleProbability, float lightProbability, v2 rand, const m3 &frame,
                                 const LightSample &ls, ExtraRay &nee, Ray &next) const // :52-98
    {
        if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (lightProbability > 0.0f)) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                v3 diffuse = (Cdiff / HR_KPI) * NdotO;
                v3 reflectance = diffuse;
                reflectance = reflectance * in.weight;
                reflectance = reflectance / sampleProbability;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P);
                    r.d = ls.dir;
                    r.weight = reflectance;
                    r.occlusionTest = true;
                    r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                    r.extraT = 0.0f;
                    if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                    emit(r, nee, next);
                }
            }
        } else if (ls.probability > 0.0f) {
            if (envMis())
                envMisDiffuse(in, P, N, Cdiff, sampleProbability, ls.probability, rand, frame, nee, next);
            else
                indirectDiffuseSample(in, P, N, Cdiff, sampleProbability, ls.probability, rand, frame, MISS_ENV, nee, next);
        }
    }
    HRD void indirectSpecularSample(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 Cspec, float roughnessAlpha, int lut, float roughness,
                                    float sampleProbability, float optionalLightSampleProbability, v2 rand, const m3 &frame, int missKind,
                                    ExtraRay &nee, Ray &next) const // :100-151
    {
        v3 localSpaceI = mulT(frame, I);
        v3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float IdotH = saturate(dot(I, H));
        v3 O = normalize(2.0f * IdotH * H - I);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            NdotO = saturate(NdotO);
            v3 F = F_Schlick(Cspec, IdotH);
            float G2 = G2_Smith_GGX(NdotI, NdotO, roughnessAlpha);
            float G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
            v3 specular = (F * G2) / greaterThanZero(G1);
            specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
            v3 reflectance = specular;
            reflectance = reflectance * in.weight;
            reflectance = reflectance / sampleProbability;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                if (LOD) r.coneG = widenCone(in.coneG, roughness);
                if (missKind == MISS_ENV && !S.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    // the single-scatter GGX lobe x cos towards O, with the multiscatter factor: (D F G2) / (4 N.I) x ms (microfacet.rlsl:153-220)
    HRD v3 specularTowards(v3 N, v3 I, float NdotI, v3 Cspec, float roughnessAlpha, int lut, float roughness, v3 O, float NdotO) const
    {
        v3 H = normalize(I + O);
        float NdotH = saturate(dot(N, H));
        float IdotH = saturate(dot(I, H));
        float D = D_GGX(NdotH, roughnessAlpha);
        v3 F = F_Schlick(Cspec, IdotH);
        float G = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha);
        v3 specular = (D * F * G) / greaterThanZero(4.0f * NdotI);
        specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
        return specular;
    }
    HRD void directSpecularSample(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 Cspec, float roughnessAlpha, int lut, float roughness,
                                  float sampleProbability, float lightProbability, v2 rand, const m3 &frame, const LightSample &ls, ExtraRay &nee,
                                  Ray &next) const // :153-220
    {
        if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (lightProbability > 0.0f)) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                v3 specular = specularTowards(N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, ls.dir, NdotO);
                v3 reflectance = specular;
                reflectance = reflectance * in.weight;
                reflectance = reflectance / sampleProbability;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P);
                    r.d = ls.dir;
                    r.weight = reflectance;
                    r.occlusionTest = true;
                    r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                    r.extraT = 0.0f;
                    if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                    emit(r, nee, next);
                }
            }
        } else if (ls.probability > 0.0f) {
            if (envMis())
                envMisSpecular(in, P, N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, sampleProbability, ls.probability, rand, frame, nee, next);
            else
                indirectSpecularSample(in, P, N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, sampleProbability, ls.probability, rand, frame,
                                       MISS_ENV, nee, next);
        }
    }

    struct Surface {
        v3 P, normal, tangent, bitangent, color;
        v2 uv;
        bool frontFacing;
        uint32_t triFlags;
    };
    static HRD v3 lerp3(const HR_GLOBAL float *a, float w, float u, float v)
    {
        return v3(a[0], a[1], a[2]) * w + v3(a[3], a[4], a[5]) * u + v3(a[6], a[7], a[8]) * v;
    }
    HRD Surface surface(const Ray &in, uint32_t prim, bool ccwFront, float t, float u, float v, uint32_t &material) const
    {
        const auto &a = G(S.attrs)[prim];
        Surface s;
        float w = 1.0f - u - v;
        s.P = in.o + in.d * t; // rl_IntersectionPoint
        s.normal = lerp3(a.n, w, u, v);
        s.uv = v2{a.uv[0] * w + a.uv[2] * u + a.uv[4] * v, a.uv[1] * w + a.uv[3] * u + a.uv[5] * v};
        const uint32_t mf = a.matflags;
        material = mf & kMatMask;
        s.triFlags = mf >> 24;
        s.tangent = s.bitangent = s.color = v3(0.0f);
        if (S.attrsExt && (s.triFlags & (TF_HAS_TANGENTS | TF_HAS_COLORS))) {
            const auto &e = G(S.attrsExt)[prim];
            s.tangent = lerp3(e.tan, w, u, v);
            s.bitangent = lerp3(e.bit, w, u, v);
            s.color = lerp3(e.col, w, u, v);
        }
        // rl_FrontFacing: CCW winding seen from the ray origin, flipped by rlFrontFace(RL_CW) (Mesh.cpp:86-91)
        s.frontFacing = (s.triFlags & TF_FRONT_CW) ? !ccwFront : ccwFront;
        return s;
    }

    // ---- physicallyBased.rlsl:55-331 ----
    HRD void physicallyBased(const Ray &inRay, const Surface &sf, const HR_GLOBAL hr_material &M, ExtraRay &nee, Ray &next, ExtraRay &nee2, ExtraRay &nee3, ExtraRay &nee4)
    {
        Ray in = inRay;
        const uint32_t F = M.flags;
        const bool hasTextures = (F & (HR_MF_HAS_BASE_COLOR_TEXTURE | HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | HR_MF_HAS_EMISSIVE_TEXTURE |
                                       HR_MF_HAS_NORMALMAP | HR_MF_HAS_CLEARCOAT_TEXTURE | HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE |
                                       HR_MF_HAS_CLEARCOAT_NORMALMAP)) != 0;
        const bool useTangentSpace = (F & (HR_MF_HAS_NORMALMAP | HR_MF_HAS_CLEARCOAT_NORMALMAP)) != 0;
        v3 baseColor(M.base_color[0], M.base_color[1], M.base_color[2]);
        float alpha = 1.0f;
        if (F & HR_MF_HAS_BASE_COLOR_TEXTURE) { // :59-65
            v4 s = tex(M.base_color_texture, sf.uv);
            baseColor = baseColor * v3(s.x, s.y, s.z);
            alpha = s.w;
        }
        if ((F & HR_MF_VERTEX_COLORS) && (sf.triFlags & TF_HAS_COLORS)) baseColor = baseColor * sf.color; // :66-68
        if (F & HR_MF_ALPHA_MASK) { // :70-91 (occlusion rays are resolved inside the any-hit traversal)
            if (alpha < 1.0f) {
                next = createRay(in, sf.P);
                return;
            }
        }
        v3 N = normalize(sf.normal); // :93
        if (F & HR_MF_DOUBLE_SIDED) { // :95-108
            if (!sf.frontFacing) N = -N;
        } else if (!sf.frontFacing) {
            next = createRay(in, sf.P);
            return;
        }
        v3 clearCoatN = N;
        if (F & HR_MF_HAS_NORMALMAP) { // :112-118
            m3 nt{normalize(sf.tangent), normalize(sf.bitangent), N};
            v4 s = tex(M.normalmap, sf.uv);
            v3 normalTS = v3(s.x, s.y, s.z) * 2.0f - v3(1.0f);
            N = normalize(mul(nt, normalTS));
        }
        if (F & HR_MF_HAS_CLEARCOAT_NORMALMAP) { // :120-126
            m3 nt{normalize(sf.tangent), normalize(sf.bitangent), clearCoatN};
            v4 s = tex(M.clear_coat_normalmap, sf.uv);
            v3 normalTS = v3(s.x, s.y, s.z) * 2.0f - v3(1.0f);
            clearCoatN = normalize(mul(nt, normalTS));
        }
        m3 frame = orthonormalFrame(N); // :128
        v3 V = -in.d;
        float NdotV = saturate(dot(N, V));
        float metallic = M.metallic, roughness = M.roughness, roughnessAlpha = M.roughness_alpha;
        if (F & HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE) { // :135-140 (.bg swizzle)
            v4 s = tex(M.metallic_roughness_texture, sf.uv);
            metallic = metallic * s.z;
            roughness = roughness * s.y;
            roughnessAlpha = roughness * roughness;
        }
        float clearCoat = M.clear_coat, clearCoatRoughness = M.clear_coat_roughness, clearCoatRoughnessAlpha = M.clear_coat_roughness_alpha;
        if (F & HR_MF_HAS_CLEARCOAT_TEXTURE) clearCoat = clearCoat * tex(M.clear_coat_texture, sf.uv).x; // :145-147
        if (F & HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE) { // :148-151
            clearCoatRoughness = clearCoatRoughness * tex(M.clear_coat_roughness_texture, sf.uv).x;
            clearCoatRoughnessAlpha = clearCoatRoughness * clearCoatRoughness;
        }
        v3 emissive(M.emissive_color[0], M.emissive_color[1], M.emissive_color[2]);
        if (F & HR_MF_HAS_EMISSIVE_TEXTURE) { // :154-156
            v4 s = tex(M.emissive_texture, sf.uv);
            emissive = v3(s.x, s.y, s.z);
        }
        if (pp.enable_visualizer == 1) { // :158-203
            switch (pp.visualizer_mode) {
            case HR_VIS_GEOMETRIC_NORMALS: accumulate4((sf.normal + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_UVS: if (hasTextures) accumulate4(v3(sf.uv.x, sf.uv.y, 0.0f), 1.0f); break;
            case HR_VIS_TANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.tangent + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BITANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.bitangent + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_NORMALMAP:
                if (F & HR_MF_HAS_NORMALMAP) {
                    v4 s = tex(M.normalmap, sf.uv);
                    accumulate4(v3(s.x, s.y, s.z), 1.0f);
                }
                break;
            case HR_VIS_FINAL_NORMALS: accumulate4((N + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BASE_COLOR: accumulate4(baseColor, 1.0f); break;
            case HR_VIS_EMISSIVE: accumulate4(emissive, 1.0f); break;
            case HR_VIS_ROUGHNESS: accumulate4(v3(roughness), 1.0f); break;
            case HR_VIS_METALLIC: accumulate4(v3(metallic), 1.0f); break;
            case HR_VIS_CLEARCOAT: accumulate4(v3(clearCoat), 1.0f); break;
            case HR_VIS_CLEARCOAT_ROUGHNESS: accumulate4(v3(clearCoatRoughness), 1.0f); break;
            case HR_VIS_SHADER: accumulate4(v3(1.0f, 0.0f, 0.0f), 1.0f); break;
            case HR_VIS_CLEARCOAT_NORMALMAP:
                if (F & HR_MF_HAS_CLEARCOAT_NORMALMAP) {
                    v4 s = tex(M.clear_coat_normalmap, sf.uv);
                    accumulate4(v3(s.x, s.y, s.z), 1.0f);
                }
                break;
            default: break;
            }
            return;
        }
        performAccumulate(in.weight * emissive); // :205
        float clearCoatNdotV = saturate(dot(clearCoatN, V));
        float clearCoatF = F_Schlick(0.04f, clearCoatNdotV); // :210
        float clearCoatScale = clearCoatF * clearCoat;
        float clearCoatBottomLayerScale = 1.0f - clearCoatScale;
        v3 Cdiff = (baseColor * (1.0f - metallic)) * clearCoatBottomLayerScale;                      // :214
        v3 Cspec = mix(v3(M.specular_f0), baseColor, v3(metallic)) * clearCoatBottomLayerScale;      // :220
        float diffuseLuminance = luminosity(Cdiff);
        float specularLuminance = luminosity(Cspec);
        float probabilityNormalization = 1.0f / greaterThanZero(diffuseLuminance + specularLuminance + clearCoatScale);
        float diffuseProbability = diffuseLuminance * probabilityNormalization;
        float specularProbability = specularLuminance * probabilityNormalization;
        float clearCoatProbability = clearCoatScale * probabilityNormalization;

        const int si = pp.sample_index + in.sequenceIndexOffset;
        { // direct lighting :236-273
            v2 rand = getSequenceValue(in.sequenceID + in.depth, si);
            v2 probability = getSequenceValue(in.sequenceID + in.depth + 1, si);
            if (allLights()) {
                // HR_ESTIMATOR_ALL_LIGHTS.  (1) One analytic light, picked among the analytic lights only, lights the WHOLE BSDF
                // (diffuse + specular + clearcoat): a light in a single direction needs no choice of lobe, and the choice is noise.
                LightSample ls = computeLightSample(N, probability.y, sf.P, true);
                if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (probability.y > 0.0f)) {
                    v3 f(0.0f);
                    float NdotO = dot(N, ls.dir);
                    if (NdotO > 0.0f) {
                        NdotO = saturate(NdotO);
                        f = (Cdiff / HR_KPI) * NdotO;
                        if (specularProbability > 0.0f)
                            f = f + specularTowards(N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness, ls.dir, NdotO);
                    }
                    float coatNdotO = dot(clearCoatN, ls.dir);
                    if (clearCoatProbability > 0.0f && coatNdotO > 0.0f)
                        f = f + specularTowards(clearCoatN, V, clearCoatNdotV, v3(clearCoatScale), clearCoatRoughnessAlpha, M.multiscatter_lut,
                                                clearCoatRoughness, ls.dir, saturate(coatNdotO));
                    v3 reflectance = f * in.weight;
                    reflectance = reflectance / ls.probability;
                    if (dot(reflectance, reflectance) > 0.0f) {
                        Ray r = createRay(in, sf.P);
                        r.d = ls.dir;
                        r.weight = reflectance;
                        r.occlusionTest = true;
                        r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                        r.extraT = 0.0f;
                        if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                        keep(r, nee2);
                    }
                }
                // (2) The environment, always: one MIS-weighted sample per vertex, three at a camera ray's hit (that is where the
                // image's noise comes from: c3 needs 1341 passes with direct lighting alone, 1381 with eight bounces), each with its
                // own choice of lobe (the lobe variable shifted by thirds) and its own sequence values
                if (S.lights.env_enabled) {
                    const bool mis = envMis();
                    const int nSamples = (mis && in.depth == 0) ? kPrimaryEnvSamples : 1;
                    const float nEnv = (float)nSamples;
                    for (int j = 0; j < nSamples; ++j) {
                        float u = probability.x + (float)j * 0.333333343f;
                        if (u > 1.0f) u = u - 1.0f;
                        // the lobe of this sample: 0 diffuse, 1 clearcoat, 2 specular, 3 none
                        const int lobe = (u <= diffuseProbability) ? 0
                                         : ((u <= (diffuseProbability + clearCoatProbability))
                                                ? 1
                                                : ((u <= (diffuseProbability + clearCoatProbability + specularProbability)) ? 2 : 3));
                        if (lobe == 3) continue;
                        ExtraRay out;
                        Ray unusedNext;
                        out.valid = false;
                        if (lobe == 0) {
                            if (mis)
                                envMisDiffuse(in, sf.P, N, Cdiff, diffuseProbability, nEnv, rand, frame, out, unusedNext, j);
                            else
                                indirectDiffuseSample(in, sf.P, N, Cdiff, diffuseProbability, 1.0f, rand, frame, MISS_ENV, out, unusedNext);
                        } else {
                            const bool coat = lobe == 1;
                            const v3 lN = coat ? clearCoatN : N, lC = coat ? v3(clearCoatScale) : Cspec;
                            const float lNdotV = coat ? clearCoatNdotV : NdotV, lAlpha = coat ? clearCoatRoughnessAlpha : roughnessAlpha,
                                        lRough = coat ? clearCoatRoughness : roughness, lProb = coat ? clearCoatProbability : specularProbability;
                            if (mis)
                                envMisSpecular(in, sf.P, lN, V, lNdotV, lC, lAlpha, M.multiscatter_lut, lRough, lProb, nEnv, rand, frame, out, unusedNext, j);
                            else
                                indirectSpecularSample(in, sf.P, lN, V, lNdotV, lC, lAlpha, M.multiscatter_lut, lRough, lProb, 1.0f, rand, frame, MISS_ENV, out,
                                                       unusedNext);
                        }
                        if (!out.valid) continue;
                        if (j == 0)
                            nee = out;
                        else if (j == 1)
                            nee3 = out;
                        else
                            nee4 = out;
                    }
                }
            } else if (probability.x <= (diffuseProbability + clearCoatProbability + specularProbability)) {
                // diffuse, clearcoat or specular lobe (physicallyBased.rlsl:246-272).  The light is picked about the lobe's normal, once;
                // clearcoat and specular are the same function on the lobe's own parameters — selected first and called once, so that
                // lanes of both lobes run together and the code exists once
                const bool diffuse = probability.x <= diffuseProbability;
                const bool coat = !diffuse && probability.x <= (diffuseProbability + clearCoatProbability);
                const v3 lN = coat ? clearCoatN : N;
                const LightSample ls = computeLightSample(lN, probability.y, sf.P);
                if (diffuse) {
                    directDiffuseSample(in, sf.P, N, Cdiff, diffuseProbability, probability.y, rand, frame, ls, nee, next);
                } else {
                    const v3 lC = coat ? v3(clearCoatScale) : Cspec;
                    const float lNdotV = coat ? clearCoatNdotV : NdotV, lAlpha = coat ? clearCoatRoughnessAlpha : roughnessAlpha,
                                lRough = coat ? clearCoatRoughness : roughness, lProb = coat ? clearCoatProbability : specularProbability;
                    directSpecularSample(in, sf.P, lN, V, lNdotV, lC, lAlpha, M.multiscatter_lut, lRough, lProb, probability.y, rand, frame, ls, nee, next);
                }
            }
        }
        if (in.depth < pp.max_ray_depth) { // :277-330
            if (in.depth > 3) {
                v2 rand = getSequenceValue(in.sequenceID + in.depth + 2, si);
                float probability = fmax_(in.weight.x, fmax_(in.weight.y, in.weight.z));
                if (rand.x >= probability) return;
                in.weight = in.weight / probability;
            }
            v2 rand = getSequenceValue(in.sequenceID + in.depth + 3, si);
            v2 probability = getSequenceValue(in.sequenceID + in.depth + 4, si);
            ExtraRay dummyNee;
            dummyNee.valid = false;
            if (probability.x <= diffuseProbability) {
                indirectDiffuseSample(in, sf.P, N, Cdiff, diffuseProbability, 1.0f, rand, frame, MISS_NONE, dummyNee, next);
            } else if (probability.x <= (diffuseProbability + clearCoatProbability + specularProbability)) {
                const bool coat = probability.x <= (diffuseProbability + clearCoatProbability); // (:305-327, as above)
                const v3 lN = coat ? clearCoatN : N, lC = coat ? v3(clearCoatScale) : Cspec;
                const float lNdotV = coat ? clearCoatNdotV : NdotV, lAlpha = coat ? clearCoatRoughnessAlpha : roughnessAlpha,
                            lRough = coat ? clearCoatRoughness : roughness, lProb = coat ? clearCoatProbability : specularProbability;
                indirectSpecularSample(in, sf.P, lN, V, lNdotV, lC, lAlpha, M.multiscatter_lut, lRough, lProb, 1.0f, rand, frame, MISS_NONE, dummyNee, next);
            }
        }
    }

    // ---- glass.rlsl ----
    HRD void indirectSpecularGlassSample(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 weight, v3 baseColor, float roughnessAlpha,
                                         float materialRoughnessAlpha, float optionalLightSampleProbability, v2 rand, const m3 &frame, int missKind,
                                         ExtraRay &nee, Ray &next) const // :47-81
    {
        v3 localSpaceI = mulT(frame, I);
        v3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float IdotH = saturate(dot(I, H));
        v3 O = normalize(2.0f * IdotH * H - I);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            NdotO = saturate(NdotO);
            float NdotH = saturate(dot(N, H));
            float G = G2_Smith_GGX(NdotO, NdotI, materialRoughnessAlpha); // :63 uses Material.roughnessAlpha
            v3 reflectance = baseColor * ((G * IdotH) / (NdotH * NdotI));
            reflectance = reflectance * weight;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                if (LOD) r.coneG = widenCone(in.coneG, roughnessAlpha);
                if (missKind == MISS_ENV && !S.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    // HR_ESTIMATOR_ENV_MIS / HR_ESTIMATOR_ALL_LIGHTS on glass (include/hrcore.h; the reference lobe-samples the map here: glass.rlsl:83-129
    // -> :47-81): the reflection's next-event ray towards the environment comes from the visible-normal lobe or from the map's importance
    // table, half the time each, balance heuristic; BRDF x cos as glass.rlsl:104-109 has it for an analytic light (D G2 / (4 N.I) x
    // baseColor), lobe density D G1 / (4 N.I); selection variable: sequence ID + depth + 5.  Same operations as the oracle's.
    HRD void envMisGlass(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 weight, v3 baseColor, float roughnessAlpha, float envProbability, v2 rand,
                         const m3 &frame, ExtraRay &nee, Ray &next) const
    {
        const v2 sel = getSequenceValue(in.sequenceID + in.depth + 5, pp.sample_index + in.sequenceIndexOffset);
        v3 O, H;
        if (sel.x < 0.5f) {
            H = mul(frame, sampleVisibleGGX(mulT(frame, I), rand.x, rand.y, roughnessAlpha));
            O = normalize(2.0f * saturate(dot(I, H)) * H - I);
        } else {
            O = sampleEnv(rand.x, rand.y);
            H = normalize(I + O);
        }
        float NdotO = dot(N, O);
        if (!(NdotO > 0.0f)) return;
        NdotO = saturate(NdotO);
        const float NdotH = saturate(dot(N, H));
        const float D = D_GGX(NdotH, roughnessAlpha);
        const float G2 = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha), G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
        const float specular = (D * G2) / greaterThanZero(4.0f * NdotI);
        const float pLobe = (D * G1) / greaterThanZero(4.0f * NdotI), pMap = envPdf(O);
        v3 reflectance = specular * baseColor;
        reflectance = reflectance * weight;
        reflectance = reflectance / greaterThanZero(0.5f * pLobe + 0.5f * pMap);
        reflectance = reflectance / envProbability;
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }
    HRD void directSpecularGlassSample(const Ray &in, v3 P, v3 N, v3 I, float NdotI, v3 weight, v3 baseColor, float roughnessAlpha,
                                       float materialRoughnessAlpha, float lightProbability, v2 rand, const m3 &frame, ExtraRay &nee, Ray &next,
                                       ExtraRay &nee2) const // :83-129
    {
        const bool both = allLights(); // HR_ESTIMATOR_ALL_LIGHTS: an analytic light (-> nee2) AND the environment (-> nee), not one of them
        LightSample ls = computeLightSample(N, lightProbability, P, both);
        if (ls.type != LIGHT_TYPE_ENVIRONMENT) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                v3 H = normalize(I + ls.dir);
                float NdotH = saturate(dot(N, H));
                float D = D_GGX(NdotH, roughnessAlpha);
                float G = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha);
                float specular = (D * G) / greaterThanZero(4.0f * NdotI);
                v3 reflectance = specular * baseColor;
                reflectance = reflectance * weight;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P);
                    r.d = ls.dir;
                    r.weight = reflectance;
                    r.occlusionTest = true;
                    r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                    r.extraT = 0.0f;
                    if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                    if (both)
                        keep(r, nee2);
                    else
                        emit(r, nee, next);
                }
            }
        }
        if (both) {
            if (S.lights.env_enabled) {
                if (envMis())
                    envMisGlass(in, P, N, I, NdotI, weight, baseColor, roughnessAlpha, 1.0f, rand, frame, nee, next);
                else
                    indirectSpecularGlassSample(in, P, N, I, NdotI, weight, baseColor, roughnessAlpha, materialRoughnessAlpha, 1.0f, rand, frame, MISS_ENV,
                                                nee, next);
            }
        } else if (ls.type != LIGHT_TYPE_ENVIRONMENT) {
        } else if (ls.probability > 0.0f) {
            if (envMis())
                envMisGlass(in, P, N, I, NdotI, weight, baseColor, roughnessAlpha, ls.probability, rand, frame, nee, next);
            else
                indirectSpecularGlassSample(in, P, N, I, NdotI, weight, baseColor, roughnessAlpha, materialRoughnessAlpha, ls.probability, rand, frame,
                                            MISS_ENV, nee, next);
        }
    }
    HRD void glass(const Ray &in, const Surface &sf, float hitT, const HR_GLOBAL hr_material &M, ExtraRay &nee, Ray &next, ExtraRay &nee2) // :138-280
    {
        const uint32_t F = M.flags;
        const bool hasTextures = (F & (HR_MF_HAS_BASE_COLOR_TEXTURE | HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | HR_MF_HAS_NORMALMAP)) != 0;
        const bool useTangentSpace = (F & HR_MF_HAS_NORMALMAP) != 0;
        v3 N = normalize(sf.normal);
        float nIn = 1.0f;
        float nOut = M.ior;
        v3 weight = in.weight;
        if (F & HR_MF_HAS_NORMALMAP) { // :145-151
            m3 nt{normalize(sf.tangent), normalize(sf.bitangent), N};
            v4 s = tex(M.normalmap, sf.uv);
            v3 normalTS = v3(s.x, s.y, s.z) * 2.0f - v3(1.0f);
            N = normalize(mul(nt, normalTS));
        }
        v3 baseColor(M.base_color[0], M.base_color[1], M.base_color[2]);
        if (F & HR_MF_HAS_BASE_COLOR_TEXTURE) { // :154-156
            v4 s = tex(M.base_color_texture, sf.uv);
            baseColor = baseColor * v3(s.x, s.y, s.z);
        }
        if ((F & HR_MF_VERTEX_COLORS) && (sf.triFlags & TF_HAS_COLORS)) baseColor = baseColor * sf.color;
        if (!sf.frontFacing) { // :161-167, beersLaw :131-136
            N = -N;
            nIn = M.ior;
            nOut = 1.0f;
            v3 absorption = v3(1.0f) - baseColor;
            float rayLength = hitT;
            v3 e = absorption * M.density * -rayLength;
            weight = in.weight * v3(exp_(e.x), exp_(e.y), exp_(e.z));
        }
        float roughness = M.roughness, roughnessAlpha = M.roughness_alpha;
        if (F & HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE) { // :171-176
            v4 s = tex(M.metallic_roughness_texture, sf.uv);
            roughness = roughness * s.y;
            roughnessAlpha = roughness * roughness;
        }
        if (pp.enable_visualizer == 1) { // :179-210
            switch (pp.visualizer_mode) {
            case HR_VIS_GEOMETRIC_NORMALS: accumulate4((sf.normal + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_FINAL_NORMALS: accumulate4((N + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BASE_COLOR: accumulate4(baseColor, 1.0f); break;
            case HR_VIS_ROUGHNESS: accumulate4(v3(roughness), 1.0f); break;
            case HR_VIS_SHADER: accumulate4(v3(0.0f, 1.0f, 0.0f), 1.0f); break;
            case HR_VIS_UVS: if (hasTextures) accumulate4(v3(sf.uv.x, sf.uv.y, 0.0f), 1.0f); break;
            case HR_VIS_TANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.tangent + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BITANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.bitangent + v3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_NORMALMAP:
                if (F & HR_MF_HAS_NORMALMAP) {
                    v4 s = tex(M.normalmap, sf.uv);
                    accumulate4(v3(s.x, s.y, s.z), 1.0f);
                }
                break;
            default: break;
            }
            return;
        }
        m3 frame = orthonormalFrame(N); // :212
        v3 I = -in.d;
        float eta = nIn / nOut;
        v3 localSpaceI = mulT(frame, I);
        const int si = pp.sample_index + in.sequenceIndexOffset;
        v2 rand = getSequenceValue(in.sequenceID + in.depth, si);
        v3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float HdotI = saturate(dot(H, I));
        v2 refractProbability = getSequenceValue(in.sequenceID + in.depth + 1, si);
        float Fr = F_Fresnel(eta, HdotI);
        float NdotI = saturate(dot(N, I));
        if (!sf.frontFacing) refractProbability = v2{refractProbability.x, 0.0f}; // :227-231
        if (refractProbability.y < (1.0f - Fr)) { // :234-256
            v3 O = normalize(refract(-I, H, eta));
            float NdotO = abs_(dot(N, O));
            float G2 = G2_Smith_GGX(NdotI, NdotO, roughnessAlpha);
            float G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
            v3 transmission = baseColor * G2 / greaterThanZero(G1);
            transmission = transmission * weight;
            if (dot(transmission, transmission) > 1e-5f && in.depth < pp.max_ray_depth) {
                Ray r = createRay(in, sf.P);
                r.d = O;
                r.weight = transmission;
                r.occlusionTest = false;
                r.extraT = 0.0f;
                r.missKind = S.lights.env_enabled ? MISS_ENV : MISS_NONE;
                r.missIdx = 0;
                if (LOD) r.coneG = widenCone(in.coneG, roughnessAlpha);
                next = r;
            }
        } else { // :257-279
            {
                rand = getSequenceValue(in.sequenceID + in.depth + 2, si);
                directSpecularGlassSample(in, sf.P, N, I, NdotI, weight, baseColor, roughnessAlpha, M.roughness_alpha, refractProbability.x, rand,
                                          frame, nee, next, nee2);
            }
            if (in.depth < pp.max_ray_depth) {
                if (in.depth > 3) {
                    v2 rr = getSequenceValue(in.sequenceID + in.depth + 3, si);
                    float probability = fmax_(weight.x, fmax_(weight.y, weight.z));
                    if (rr.x >= probability) return;
                    weight = weight / probability;
                }
                rand = getSequenceValue(in.sequenceID + in.depth + 4, si);
                ExtraRay dummyNee;
                dummyNee.valid = false;
                indirectSpecularGlassSample(in, sf.P, N, I, NdotI, weight, baseColor, roughnessAlpha, M.roughness_alpha, 1.0f, rand, frame, MISS_NONE,
                                            dummyNee, next);
            }
        }
    }
};
using Shader = ShaderT<0>;

// ---- perspective.rlsl:39-93 (frame shader) ----
HRD float randomRL(float sx, float sy) { return fract(sin_(sx * 12.9898f + sy * 78.233f) * 43758.5453123f); } // utility.rlsl:15-18

// Returns false when the pixel is skipped this pass (interactive block mode).
HRD bool generatePrimary(const SceneDev &S, const hr_pass_params &pp, int W, int H, int x, int y, Ray &out)
{
    const float Wf = (float)W, Hf = (float)H;
    const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;
    if (pp.interactive_mode != 0) { // :42-57; block table: hr_interactive_blocks_set (the unshuffled list by default)
        const int bsx = pp.block_size[0], bsy = pp.block_size[1];
        const int bix = (int)(fcx - 0.5f) / bsx, biy = (int)(fcy - 0.5f) / bsy;
        float randX = randomRL((float)bix, (float)biy);
        float randY = randomRL((float)biy, (float)bix);
        float su = (1.0f / (float)bsx) * (float)pp.current_block_pixel[0] + randX;
        float sv = (1.0f / (float)bsy) * (float)pp.current_block_pixel[1] + randY;
        int tx = (int)floor_(su * (float)bsx) % bsx, ty = (int)floor_(sv * (float)bsy) % bsy;
        int sampleX = ty, sampleY = tx;
        if (S.blockNx == bsx && S.blockNy == bsy) sampleX = S.blockCoords[2 * (ty * bsx + tx)], sampleY = S.blockCoords[2 * (ty * bsx + tx) + 1];
        int thisX = (int)(fcx - 0.5f) % bsx, thisY = (int)(fcy - 0.5f) % bsy;
        if (thisX != sampleX || thisY != sampleY) return false;
    }
    int sequenceID = (int)floor_(randomRL(fcx / Wf, fcy / Hf) * (float)S.nSeq); // :62
    int offIdx = (int)(fcy * Hf + fcx);                                         // :64 (height stride, pixel centres)
    offIdx = offIdx % S.nSeqOffsets;                                            // defined wrap instead of the reference's OOB read
    float rnd = G(S.seqOffsets)[offIdx].x;
    int sequenceIndex = (int)floor_(rnd * pp.max_sample_index); // :65
    int ws = sequenceID % S.nSeq, wv = (pp.sample_index + sequenceIndex) % S.seqLen;
    float2 sampleOffset = G(S.seq)[(size_t)ws * S.seqLen + wv];
    float spx = (fcx - 0.5f) + sampleOffset.x, spy = (fcy - 0.5f) + sampleOffset.y;
    float u = spx / Wf, v = spy / Hf;
    float cx = (2.0f * u - 1.0f) * pp.aspect_ratio * pp.fov_tan; // :72
    float cy = (1.0f - 2.0f * v) * pp.fov_tan * -1.0f;           // :73
    v3 dirCameraSpace = normalize(v3(cx, cy, -1.0f));
    v3 focalPoint = pp.focus_distance * dirCameraSpace; // :77
    int apIdx = (sequenceID * S.seqLen + pp.sample_index) % (S.nSeq * S.seqLen); // :78, wrapped
    float2 ap = G(S.aperture)[apIdx];
    float ax = ((ap.x * 2.0f) - 1.0f) * pp.aperture_radius, ay = ((ap.y * 2.0f) - 1.0f) * pp.aperture_radius;
    v3 origin(ax, ay, 0.0f);
    v3 dir = focalPoint - origin;
    out.o = xformPoint(pp.view_matrix, origin);          // :84
    out.d = normalize(xformVector(pp.view_matrix, dir)); // :85
    out.maxT = __builtin_inff();
    out.missKind = S.lights.env_enabled ? MISS_ENV : MISS_NONE;
    out.missIdx = 0;
    out.weight = v3(1.0f);
    out.sequenceID = sequenceID;
    out.sequenceIndexOffset = sequenceIndex;
    out.extraT = 0.0f;
    out.depth = 0;
    out.occlusionTest = false;
    out.valid = true;
    out.coneW = 0.0f;                     // ray cone of HR_TEXTURE_LOD_CONE: a pinhole pixel (the aperture is ignored)
    out.coneG = 2.0f * pp.fov_tan / Hf;   // one pixel's angle at the image centre
    return true;
}

} // namespace hr
