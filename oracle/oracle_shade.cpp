// oracle_shade.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// Scalar restatement of the RLSL shaders OpenRL runs inside rlRenderFrame()
// (/root/reference/Source/HeatrayRenderer/PassGenerator.cpp:386).  Every function cites
// the shader text it follows (paths relative to /root/reference/Resources/shaders).
// One path per owned pixel per pass.  A pass's sample is the sum, in path order
// (A+=1, then for each bounce: emissive, NEE light, (miss) environment), of the path's
// accumulate() calls, starting from zero; the sample is added to the accumulation
// buffer once when the path ends — the order the HIP wavefront pipeline reproduces.
//
// PARITY STATUS: the shading arithmetic runs inside the closed OpenRL RLSL compiler in the reference, which is
// absent from the tree and has no tests: this restatement is "parity unpinned" against OpenRL.
// It is anchored by line-by-line citations of the RLSL text and by the analytic known-answer tests of
// tests/test_oracle_render.py; the sample tables it consumes ARE pinned (oracle_qmc.cpp).
#include "oracle_internal.h"

#include <cmath>
#include <cstring>
#include <omp.h>

namespace ora {

enum MissKind { MISS_NONE = 0, MISS_ENV = 1, MISS_DIR = 2, MISS_POINT = 3, MISS_SPOT = 4 };
// lightDefines.rlsl:13-16
enum { LIGHT_TYPE_DIRECTIONAL = 1, LIGHT_TYPE_POINT = 2, LIGHT_TYPE_SPOT = 3, LIGHT_TYPE_ENVIRONMENT = 4 };

// rayAttributes.rlsl:8-11 + the built-in ray fields the shaders touch
struct Ray {
    vec3 o, d;
    float maxT = INFINITY;
    vec3 weight;
    int sequenceID = 0, sequenceIndexOffset = 0;
    float extraT = 0.0f;
    int depth = 0;
    bool occlusionTest = false;
    int missKind = MISS_NONE, missIdx = 0; // rl_OutRay.defaultPrimitive
    int srcPrim = -1;
    bool valid = false;
    float coneW = 0.0f, coneG = 0.0f; // ray cone of HR_TEXTURE_LOD_CONE: width at the origin, spread angle
};

// The product keeps the cone in one dword of its ray record (upper halves of the two floats): the truncation is part of the contract
static inline float coneQuant(float x)
{
    uint32_t b;
    std::memcpy(&b, &x, 4);
    b &= 0xFFFF0000u;
    std::memcpy(&x, &b, 4);
    return x;
}
static inline float widenCone(float g, float roughness) { return fmin_(g + 0.25f * roughness, 1.0f); }

struct Shader {
    Context &ctx;
    const hr_pass_params &pp;
    float *fbPixel;  // RGBA of the pixel being shaded in the accumulation buffer
    float px[4];     // this pass's sample: the path's accumulate() calls are summed here in path order
                     // and added to the accumulation buffer ONCE when the path ends
    hr_pass_stats &st;
    TraceCounters tc, tcAny;
    float lodBase = -1e30f; // HR_TEXTURE_LOD_CONE: level offset of the current hit's footprint
    // HR_ESTIMATOR_ALL_LIGHTS: partial sums 1..3 of the pass's sample (analytic-light ray; 2nd, 3rd environment sample of the primary hit)
    float pxX[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    int part = 0;

    Shader(Context &c, const hr_pass_params &p, float *pixel, hr_pass_stats &s) : ctx(c), pp(p), fbPixel(pixel), st(s)
    {
        px[0] = px[1] = px[2] = px[3] = 0.0f;
    }

    // ---- sequence.rlsl:18-28 ----
    vec2 getSequenceValue(int sequenceIndex, int sampleIndex) const
    {
        int ws = sequenceIndex % ctx.nSeq;
        int wv = sampleIndex % ctx.seqLen;
        return ctx.seq[(size_t)ws * ctx.seqLen + wv];
    }

    // ---- accumulator.rlsl:12-28 ----
    void accumulate3(vec3 c)
    {
        float *t = part ? pxX[part - 1] : px; // HR_ESTIMATOR_ALL_LIGHTS: every extra occlusion ray adds to a partial sum of its own
        t[0] = t[0] + c.x;
        t[1] = t[1] + c.y;
        t[2] = t[2] + c.z;
        st.accumulates++;
    }
    void accumulate4(vec3 c, float a) // accumulate(vec4) used by the debug visualisers
    {
        accumulate3(c);
        px[3] = px[3] + a;
    }
    void performAccumulate(vec3 color)
    {
        vec3 value(0.0f);
        if (pp.enable_accumulator_visualizer == 1) {
            if (pp.show_nans == 1) {
                bool any = (color.x != color.x) || (color.y != color.y) || (color.z != color.z);
                value = any ? vec3(100.0f) : (min3(color, vec3(1.0f)) * 0.1f);
            } else if (pp.show_inf == 1) {
                bool any = std::isinf(color.x) || std::isinf(color.y) || std::isinf(color.z);
                value = any ? vec3(100.0f) : (min3(color, vec3(1.0f)) * 0.1f);
            }
        } else {
            value = min3(vec3(pp.max_channel_value), color);
        }
        accumulate3(value);
    }

    vec4 tex(int id, vec2 uv) const
    {
        if (id < 0 || id >= (int)ctx.textures.size() || !ctx.textures[id].alive) return vec4{1.0f, 1.0f, 1.0f, 1.0f}; // dummy white texel (Texture.h:188-203)
        if (pp.texture_lod == HR_TEXTURE_LOD_CONE) return sampleTextureLod(ctx.textures[id], uv.x, uv.y, lodBase + ctx.textures[id].lodScale);
        return sampleTexture(ctx.textures[id], uv.x, uv.y);
    }

    // ---- light shaders run when an occlusion ray reaches its light / a ray misses ----
    // environmentLight.rlsl:19-34
    void environmentLight(vec3 dir, vec3 weight)
    {
        float theta = atan2_(dir.x, -dir.z) + ctx.lights.env_theta_rotation;
        if (theta > kTwoPI) theta = theta - kTwoPI;
        float phi = atan2_(dir.y, sqrtf(dir.x * dir.x + dir.z * dir.z));
        float u = (theta / kTwoPI) + 0.5f;
        float v = (-phi * kOneOverPI) + 0.5f;
        vec4 t = vec4{0.0f, 0.0f, 0.0f, 0.0f};
        int id = ctx.lights.env_texture;
        if (id >= 0 && id < (int)ctx.textures.size() && ctx.textures[id].alive) t = sampleTexture(ctx.textures[id], u, 1.0f - v);
        vec3 sample = vec3(t.x, t.y, t.z) * ctx.lights.env_exposure;
        performAccumulate(weight * sample);
    }
    // directionalLight.rlsl:20-26, pointLight.rlsl:20-29, spotLight.rlsl:20-36
    void lightShader(const Ray &r, float intersectionT)
    {
        const hr_lights &L = ctx.lights;
        switch (r.missKind) {
        case MISS_ENV:
            environmentLight(r.d, r.weight);
            break;
        case MISS_DIR: {
            const float *c = L.directional_colors[r.missIdx];
            performAccumulate(r.weight * vec3(c[0], c[1], c[2]));
            break;
        }
        case MISS_POINT: {
            float s = intersectionT + r.extraT;
            float attenuation = 1.0f / (s * s);
            const float *c = L.point_colors[r.missIdx];
            performAccumulate(r.weight * vec3(c[0], c[1], c[2]) * attenuation);
            break;
        }
        case MISS_SPOT: {
            const float *sd = L.spot_directions[r.missIdx];
            float rayAngle = dot(-r.d, vec3(sd[0], sd[1], sd[2]));
            if (rayAngle >= 0.0f) {
                float s = intersectionT + r.extraT;
                float attenuation = 1.0f / (s * s);
                const float *c = L.spot_colors[r.missIdx];
                vec3 result = r.weight * vec3(c[0], c[1], c[2]) * attenuation;
                result = result * (1.0f - smoothstep(L.spot_angles[r.missIdx][0], L.spot_angles[r.missIdx][1], rayAngle));
                performAccumulate(result);
            }
            break;
        }
        default:
            break;
        }
    }

    // ---- utility.rlsl ----
    static float square(float x) { return x * x; }
    static float getSign(float x) { return x < 0.0f ? -1.0f : 1.0f; }               // :35-38
    static float pow5(float x) { return x * square(x) * square(x); }                // :141-144
    static float greaterThanZero(float f) { return fmax_(1e-5f, f); }               // :153-156
    static float luminosity(vec3 c) { return dot(c, vec3(0.33f, 0.59f, 0.11f)); }   // :163-166
    static mat3 orthonormalFrame(vec3 N)                                            // :43-60
    {
        vec3 lh(N.x, N.z, N.y);
        float s = getSign(lh.z);
        float a = -1.0f / (s + lh.z);
        float b = lh.x * lh.y * a;
        vec3 X(1.0f + s * lh.x * lh.x * a, s * b, -s * lh.x);
        vec3 Z(b, s + lh.y * lh.y * a, -lh.y);
        mat3 m;
        m.c0 = vec3(X.x, X.z, X.y);
        m.c1 = N;
        m.c2 = vec3(Z.x, Z.z, Z.y);
        return m;
    }
    static vec3 cosineWeightedSample(float u1, float u2) // :64-75
    {
        float theta = sqrtf(u1);
        float phi = kTwoPI * u2;
        float s, c;
        sincos_(phi, &s, &c);
        float x = theta * c;
        float y = sqrtf(fmax_(0.0f, 1.0f - u1));
        float z = theta * s;
        return normalize(vec3(x, y, z));
    }
    static vec3 sampleVisibleGGX(vec3 localSpaceV, float u1, float u2, float roughnessAlpha) // :109-139
    {
        vec3 zUpV(localSpaceV.x, localSpaceV.z, localSpaceV.y);
        vec3 Vh = normalize(vec3(zUpV.x * roughnessAlpha, zUpV.y * roughnessAlpha, zUpV.z));
        float lengthSquared = (Vh.x * Vh.x) + (Vh.y * Vh.y);
        vec3 T1 = (lengthSquared > 0.0f) ? vec3(-Vh.y, Vh.x, 0.0f) * inversesqrt(lengthSquared) : vec3(1.0f, 0.0f, 0.0f);
        vec3 T2 = cross(Vh, T1);
        float r = sqrtf(u1);
        float phi = kTwoPI * u2;
        float sn, cs;
        sincos_(phi, &sn, &cs);
        float t1 = r * cs;
        float t2 = r * sn;
        float s = 0.5f * (1.0f + Vh.z);
        float t1Squared = square(t1);
        t2 = (1.0f - s) * sqrtf(1.0f - t1Squared) + (s * t2);
        vec3 Nh = (t1 * T1) + (t2 * T2) + sqrtf(fmax_(0.0f, 1.0f - t1Squared - square(t2))) * Vh;
        vec3 zUp = normalize(vec3(roughnessAlpha * Nh.x, roughnessAlpha * Nh.y, fmax_(0.0f, Nh.z)));
        return vec3(zUp.x, zUp.z, zUp.y);
    }

    // ---- brdfs.rlsl ----
    static vec3 F_Schlick(vec3 Cspec, float cosTheta) { return Cspec + (vec3(1.0f) - Cspec) * pow5(1.0f - cosTheta); } // :46-50
    static float F_Schlick(float f0, float cosTheta) { return f0 + (1.0f - f0) * pow5(1.0f - cosTheta); }             // :53-57
    static float F_Fresnel(float eta, float cosThetaI)                                                                // :59-71
    {
        float sinThetaT2 = square(eta) * (1.0f - square(cosThetaI));
        if (sinThetaT2 < 1.0f) {
            float cosThetaT = sqrtf(1.0f - sinThetaT2);
            float perpendicular = square((eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT));
            float parallel = square((cosThetaI - eta * cosThetaT) / (cosThetaI + eta * cosThetaT));
            return 0.5f * (perpendicular + parallel);
        }
        return 1.0f;
    }
    static float D_GGX(float NdotH, float roughnessAlpha) // :73-78
    {
        float alpha2 = square(roughnessAlpha);
        float denominator = square(square(NdotH) * (alpha2 - 1.0f) + 1.0f);
        return kOneOverPI * (alpha2 / greaterThanZero(denominator));
    }
    static float G1_Smith_GGX(float NdotI, float roughnessAlpha) // :88-93
    {
        float alpha2 = square(roughnessAlpha);
        float denom = sqrtf(alpha2 + (1.0f - alpha2) * greaterThanZero(square(NdotI))) + NdotI;
        return (2.0f * NdotI) / greaterThanZero(denom);
    }
    static float G2_Smith_GGX(float NdotO, float NdotI, float roughnessAlpha) // :95-98
    {
        return G1_Smith_GGX(NdotO, roughnessAlpha) * G1_Smith_GGX(NdotI, roughnessAlpha);
    }

    // ---- lightSampling.rlsl:11-161 ----
    struct LightSample {
        vec3 dir;
        int missKind = MISS_NONE, missIdx = 0;
        float probability = 0.0f;
        float maxDistance = INFINITY;
        int type = 0;
    };
    LightSample computeLightSample(vec3 N, float lightProbability, vec3 P, bool withoutEnv = false) const
    {
        const hr_lights &L = ctx.lights;
        LightSample out;
        float probabilitySum = 0.0f;
        float directional[HR_MAX_DIRECTIONAL_LIGHTS] = {0, 0, 0, 0, 0};
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) {
            if (i < L.n_directional) {
                const float *d = L.directional_directions[i], *c = L.directional_colors[i];
                directional[i] = saturate(dot(N, vec3(d[0], d[1], d[2]))) * luminosity(vec3(c[0], c[1], c[2]));
                probabilitySum += directional[i];
            }
        }
        float point[HR_MAX_POINT_LIGHTS] = {0, 0, 0, 0, 0};
        vec3 pointDirs[HR_MAX_POINT_LIGHTS];
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) {
            if (i < L.n_point) {
                const float *p = L.point_positions[i], *c = L.point_colors[i];
                pointDirs[i] = normalize(vec3(p[0], p[1], p[2]) - P);
                point[i] = saturate(dot(N, pointDirs[i])) * luminosity(vec3(c[0], c[1], c[2]));
                probabilitySum += point[i];
            }
        }
        float spot[HR_MAX_SPOT_LIGHTS] = {0, 0, 0, 0, 0};
        vec3 spotDirs[HR_MAX_SPOT_LIGHTS];
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) {
            if (i < L.n_spot) {
                const float *p = L.spot_positions[i], *c = L.spot_colors[i], *sd = L.spot_directions[i];
                spotDirs[i] = normalize(vec3(p[0], p[1], p[2]) - P);
                float rayAngle = dot(vec3(sd[0], sd[1], sd[2]), -spotDirs[i]);
                spot[i] = saturate(dot(N, spotDirs[i])) * luminosity(vec3(c[0], c[1], c[2])) * ((rayAngle > 0.0f) ? 1.0f : 0.0f) *
                          ((rayAngle < L.spot_angles[i][1]) ? 0.0f : 1.0f) *
                          (1.0f - smoothstep(L.spot_angles[i][0], L.spot_angles[i][1], rayAngle));
                probabilitySum += spot[i];
            }
        }
        float environment = 0.0f;
        if (L.env_enabled && !withoutEnv) {
            // :74-79 "TODO: Add IBL importance sampling. Right now this is just a hack" — the constant 50 starves analytic lights
            // (a sun-like directional light is picked a few per cent of the time).  HR_ESTIMATOR_ENV_MIS weighs the map like the
            // other lights, by the irradiance it can deliver: pi x its mean luminosity.
            environment = envMis() ? (ctx.env.meanLum * kPI) * L.env_exposure : 50.0f * L.env_exposure;
            probabilitySum += environment;
        }
        float norm = 1.0f / greaterThanZero(probabilitySum);
        environment *= norm;
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) directional[i] *= norm;
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) point[i] *= norm;
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) spot[i] *= norm;

        float currentProbability = 0.0f;
        for (int i = 0; i < HR_MAX_DIRECTIONAL_LIGHTS; ++i) {
            if (i < L.n_directional) {
                currentProbability += directional[i];
                if (directional[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *d = L.directional_directions[i];
                    out.dir = vec3(d[0], d[1], d[2]);
                    out.missKind = MISS_DIR, out.missIdx = i;
                    out.probability = directional[i];
                    out.type = LIGHT_TYPE_DIRECTIONAL;
                    return out;
                }
            }
        }
        for (int i = 0; i < HR_MAX_POINT_LIGHTS; ++i) {
            if (i < L.n_point) {
                currentProbability += point[i];
                if (point[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *p = L.point_positions[i];
                    out.dir = pointDirs[i];
                    out.missKind = MISS_POINT, out.missIdx = i;
                    out.probability = point[i];
                    out.maxDistance = length(vec3(p[0], p[1], p[2]) - P);
                    out.type = LIGHT_TYPE_POINT;
                    return out;
                }
            }
        }
        for (int i = 0; i < HR_MAX_SPOT_LIGHTS; ++i) {
            if (i < L.n_spot) {
                currentProbability += spot[i];
                if (spot[i] > 0.0f && lightProbability <= currentProbability) {
                    const float *p = L.spot_positions[i];
                    out.dir = spotDirs[i];
                    out.missKind = MISS_SPOT, out.missIdx = i;
                    out.probability = spot[i];
                    out.maxDistance = length(vec3(p[0], p[1], p[2]) - P);
                    out.type = LIGHT_TYPE_SPOT;
                    return out;
                }
            }
        }
        out.type = LIGHT_TYPE_ENVIRONMENT;
        out.probability = environment;
        return out;
    }

    // createRay(): the child inherits every attribute of rl_InRay, starts at the hit point,
    // depth + 1 (SURVEY §8a a6 [assumed]).
    static Ray createRay(const Ray &in, vec3 P, int prim)
    {
        Ray r = in;
        r.o = P;
        r.depth = in.depth + 1;
        r.srcPrim = prim;
        r.valid = true;
        return r;
    }
    // An "environment" branch with the environment light removed would emit a second
    // non-occlusion ray (lightSampling.rlsl:157-160 + microfacet.rlsl:47); it can only
    // happen for a sample value of exactly 0 and is dropped.
    void emit(Ray r, Ray &nee, Ray &next)
    {
        if (r.occlusionTest)
            nee = r;
        else
            next = r;
    }

    // ---- HR_ESTIMATOR_ENV_MIS (include/hrcore.h): importance sampling of the environment map + one-sample MIS ----
    // Not in the reference (its shaders mark it TODO: lightSampling.rlsl:75-77, microfacet.rlsl:94-96); this is the contract the
    // HIP kernels reproduce bit for bit, with its own known-answer tests (tests/test_oracle_kat.py).
    bool envMis() const { return pp.estimator != HR_ESTIMATOR_REFERENCE && ctx.env.w > 0; }
    bool allLights() const { return pp.estimator == HR_ESTIMATOR_ALL_LIGHTS; }
    static constexpr int kPrimaryEnvSamples = 3; // environment samples HR_ESTIMATOR_ALL_LIGHTS takes at a camera ray's hit
    // the texel a direction falls into, with environmentLight.rlsl:19-34's mapping (u to the right, t upwards)
    void envTexelOf(vec3 dir, int &i, int &j) const
    {
        float theta = atan2_(dir.x, -dir.z) + ctx.lights.env_theta_rotation;
        if (theta > kTwoPI) theta = theta - kTwoPI;
        float phi = atan2_(dir.y, sqrtf(dir.x * dir.x + dir.z * dir.z));
        float u = (theta / kTwoPI) + 0.5f;
        float t = 1.0f - ((-phi * kOneOverPI) + 0.5f);
        const int w = ctx.env.w, h = ctx.env.h;
        int ii = (int)floorf(u * (float)w) % w;
        i = ii < 0 ? ii + w : ii;
        int jj = (int)floorf(t * (float)h);
        j = jj < 0 ? 0 : (jj >= h ? h - 1 : jj);
    }
    // density per solid angle of the table's distribution at `dir`: P(texel) / (texel area in (azimuth, elevation)) / cos(elevation)
    float envPdf(vec3 dir) const
    {
        int i, j;
        envTexelOf(dir, i, j);
        const float cosEl = sqrtf(dir.x * dir.x + dir.z * dir.z);
        const float K = ((float)ctx.env.w * (float)ctx.env.h) / (kTwoPI * kPI);
        return (ctx.env.prob[(size_t)j * ctx.env.w + i] * K) / fmax_(cosEl, 1e-6f);
    }
    static int cdfFind(const float *cdf, int n, float x) // largest k in [0, n) with cdf[k] <= x (cdf[0] = 0)
    {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cdf[mid] <= x)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lo;
    }
    vec3 sampleEnv(float u1, float u2) const
    {
        const int w = ctx.env.w, h = ctx.env.h;
        const float *rc = ctx.env.rowCdf.data();
        const int j = cdfFind(rc, h, u1);
        const float fy = (u1 - rc[j]) / fmax_(rc[j + 1] - rc[j], 1e-20f);
        const float *cc = ctx.env.colCdf.data() + (size_t)j * (w + 1);
        const int i = cdfFind(cc, w, u2);
        const float fx = (u2 - cc[i]) / fmax_(cc[i + 1] - cc[i], 1e-20f);
        const float t = ((float)j + saturate(fy)) / (float)h, u = ((float)i + saturate(fx)) / (float)w;
        const float elevation = (t - 0.5f) * kPI;
        const float azimuth = (u - 0.5f) * kTwoPI - ctx.lights.env_theta_rotation;
        float se, ce, sa, ca;
        sincos_(elevation, &se, &ce);
        sincos_(azimuth, &sa, &ca);
        return vec3(ce * sa, se, -(ce * ca));
    }
    // The next-event ray towards the environment of a diffuse lobe: drawn from the lobe (cosine) or from the map, half the time
    // each; weight f cos / (p_lobe / 2 + p_map / 2) — the balance heuristic of the one-sample model.
    void envMisDiffuse(const Ray &in, vec3 P, int prim, vec3 N, vec3 Cdiff, float sampleProbability, float envProbability, vec2 rand,
                       const mat3 &frame, Ray &nee, Ray &next, int which = 0)
    {
        // which: 0 = the vertex's environment sample; 1, 2 = the extra ones HR_ESTIMATOR_ALL_LIGHTS takes at a camera ray's hit
        const vec2 sel = getSequenceValue(in.sequenceID + in.depth + 5 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // the map's row variable is stratified over the vertex's samples (sample j of n draws it from the j-th n-th of [0, 1): the
        // map's bright regions each get their sample), everything else comes from the sample's own sequence values
        const float envU1 = (rand.x + (float)which) / envProbability;
        if (which > 0) rand = getSequenceValue(in.sequenceID + in.depth + 4 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // how often the lobe's own sampler is used instead of the map's: half the time (ENV_MIS), an eighth with ALL_LIGHTS — a
        // cosine lobe rarely finds a small bright source, and every such sample is one the source does not get
        const float cLobe = (pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) ? 0.125f : 0.5f, cMap = 1.0f - cLobe;
        vec3 O = (sel.x < cLobe) ? mul(frame, cosineWeightedSample(rand.x, rand.y)) : sampleEnv((pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) ? envU1 : rand.x, rand.y);
        float NdotO = dot(N, O);
        if (!(NdotO > 0.0f)) return;
        NdotO = saturate(NdotO);
        const float pLobe = NdotO / kPI, pMap = envPdf(O);
        vec3 reflectance = (Cdiff / kPI) * NdotO;
        reflectance = reflectance * in.weight;
        reflectance = reflectance / (cLobe * pLobe + cMap * pMap);
        reflectance = reflectance / sampleProbability;
        reflectance = reflectance / envProbability;
        // (no 1e-5 cut-off on the weight here: an importance-sampled ray towards a bright texel carries a small weight and a large radiance)
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P, prim);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }
    void envMisSpecular(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 Cspec, float roughnessAlpha, int lut, float roughness,
                        float sampleProbability, float envProbability, vec2 rand, const mat3 &frame, Ray &nee, Ray &next, int which = 0)
    {
        const vec2 sel = getSequenceValue(in.sequenceID + in.depth + 5 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        // the map's row variable is stratified over the vertex's samples (sample j of n draws it from the j-th n-th of [0, 1): the
        // map's bright regions each get their sample), everything else comes from the sample's own sequence values
        const float envU1 = (rand.x + (float)which) / envProbability;
        if (which > 0) rand = getSequenceValue(in.sequenceID + in.depth + 4 + 2 * which, pp.sample_index + in.sequenceIndexOffset);
        vec3 O, H;
        if (sel.x < 0.5f) { // the lobe's own sampler (microfacet.rlsl:100-151)
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
        const vec3 F = F_Schlick(Cspec, IdotH);
        const float G2 = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha), G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
        // BRDF x cos as in directSpecularSample (microfacet.rlsl:153-220); density of the visible-normal sampler: D G1 / (4 N.I)
        vec3 specular = (D * F * G2) / greaterThanZero(4.0f * NdotI);
        specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
        const float pLobe = (D * G1) / greaterThanZero(4.0f * NdotI), pMap = envPdf(O);
        vec3 reflectance = specular * in.weight;
        reflectance = reflectance / greaterThanZero(0.5f * pLobe + 0.5f * pMap);
        reflectance = reflectance / sampleProbability;
        reflectance = reflectance / envProbability;
        // (no 1e-5 cut-off on the weight here: an importance-sampled ray towards a bright texel carries a small weight and a large radiance)
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P, prim);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }

    // ---- microfacet.rlsl ----
    vec3 computeMultiscattering(int lut, vec3 Cspec, float NdotI, float roughness) const // :17-23
    {
        float ms = 0.0f;
        if (lut >= 0 && lut < (int)ctx.textures.size() && ctx.textures[lut].alive) ms = sampleTexture(ctx.textures[lut], NdotI, roughness).x;
        return vec3(1.0f) + Cspec * ms;
    }
    void indirectDiffuseSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 Cdiff, float sampleProbability, float optionalLightSampleProbability,
                               vec2 rand, const mat3 &frame, int missKind, Ray &nee, Ray &next) // :25-50
    {
        vec3 dir = cosineWeightedSample(rand.x, rand.y);
        vec3 O = mul(frame, dir);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            vec3 reflectance = Cdiff;
            reflectance = reflectance * in.weight;
            reflectance = reflectance / sampleProbability;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P, prim);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                r.coneG = widenCone(in.coneG, 1.0f);
                if (missKind == MISS_ENV && !ctx.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    void directDiffuseSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 Cdiff, float sampleProbability, float lightProbability, vec2 rand,
                             const mat3 &frame, Ray &nee, Ray &next) // :52-98
    {
        LightSample ls = computeLightSample(N, lightProbability, P);
        if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (lightProbability > 0.0f)) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                vec3 diffuse = (Cdiff / kPI) * NdotO;
                vec3 reflectance = diffuse;
                reflectance = reflectance * in.weight;
                reflectance = reflectance / sampleProbability;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P, prim);
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
                envMisDiffuse(in, P, prim, N, Cdiff, sampleProbability, ls.probability, rand, frame, nee, next);
            else
                indirectDiffuseSample(in, P, prim, N, Cdiff, sampleProbability, ls.probability, rand, frame, MISS_ENV, nee, next);
        }
    }
    void indirectSpecularSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 Cspec, float roughnessAlpha, int lut,
                                float roughness, float sampleProbability, float optionalLightSampleProbability, vec2 rand, const mat3 &frame,
                                int missKind, Ray &nee, Ray &next) // :100-151
    {
        vec3 localSpaceI = mulT(frame, I);
        vec3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float IdotH = saturate(dot(I, H));
        vec3 O = normalize(2.0f * IdotH * H - I);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            NdotO = saturate(NdotO);
            vec3 F = F_Schlick(Cspec, IdotH);
            float G2 = G2_Smith_GGX(NdotI, NdotO, roughnessAlpha);
            float G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
            vec3 specular = (F * G2) / greaterThanZero(G1);
            specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
            vec3 reflectance = specular;
            reflectance = reflectance * in.weight;
            reflectance = reflectance / sampleProbability;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P, prim);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                r.coneG = widenCone(in.coneG, roughness);
                if (missKind == MISS_ENV && !ctx.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    // the single-scatter GGX lobe x cos towards O, with the multiscatter factor: (D F G2) / (4 N.I) x ms (microfacet.rlsl:153-220)
    vec3 specularTowards(vec3 N, vec3 I, float NdotI, vec3 Cspec, float roughnessAlpha, int lut, float roughness, vec3 O, float NdotO) const
    {
        vec3 H = normalize(I + O);
        float NdotH = saturate(dot(N, H));
        float IdotH = saturate(dot(I, H));
        float D = D_GGX(NdotH, roughnessAlpha);
        vec3 F = F_Schlick(Cspec, IdotH);
        float G = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha);
        vec3 specular = (D * F * G) / greaterThanZero(4.0f * NdotI);
        specular = specular * computeMultiscattering(lut, Cspec, NdotI, roughness);
        return specular;
    }
    void directSpecularSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 Cspec, float roughnessAlpha, int lut,
                              float roughness, float sampleProbability, float lightProbability, vec2 rand, const mat3 &frame, Ray &nee,
                              Ray &next) // :153-220
    {
        LightSample ls = computeLightSample(N, lightProbability, P);
        if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (lightProbability > 0.0f)) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                vec3 specular = specularTowards(N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, ls.dir, NdotO);
                vec3 reflectance = specular;
                reflectance = reflectance * in.weight;
                reflectance = reflectance / sampleProbability;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P, prim);
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
                envMisSpecular(in, P, prim, N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, sampleProbability, ls.probability, rand, frame, nee, next);
            else
                indirectSpecularSample(in, P, prim, N, I, NdotI, Cspec, roughnessAlpha, lut, roughness, sampleProbability, ls.probability, rand,
                                       frame, MISS_ENV, nee, next);
        }
    }

    // Barycentric interpolation of a varying: a0*(1-u-v) + a1*u + a2*v
    static vec3 lerp3(const vec3 *a, float w, float u, float v) { return a[0] * w + a[1] * u + a[2] * v; }

    struct Surface {
        vec3 P, normal, tangent, bitangent, color;
        vec2 uv;
        bool frontFacing;
    };
    Surface surface(const Ray &in, const Hit &h) const
    {
        const TriAttr &a = ctx.attrs[h.prim];
        const Tri &tr = ctx.tris[h.prim];
        Surface s;
        float w = 1.0f - h.u - h.v;
        s.P = in.o + in.d * h.t; // rl_IntersectionPoint
        s.normal = lerp3(a.n, w, h.u, h.v);
        s.uv = vec2{a.uv[0].x * w + a.uv[1].x * h.u + a.uv[2].x * h.v, a.uv[0].y * w + a.uv[1].y * h.u + a.uv[2].y * h.v};
        s.tangent = lerp3(a.tan, w, h.u, h.v);
        s.bitangent = lerp3(a.bit, w, h.u, h.v);
        s.color = lerp3(a.col, w, h.u, h.v);
        // rl_FrontFacing: counter-clockwise winding seen from the ray origin, flipped for
        // primitives submitted with rlFrontFace(RL_CW) (Mesh.cpp:86-91).  det of
        // Möller–Trumbore = -dot(d, cross(e1, e2)).
        float det = dot(tr.e1, cross(in.d, tr.e2));
        bool ccwFront = det > 0.0f;
        s.frontFacing = (a.flags & TF_FRONT_CW) ? !ccwFront : ccwFront;
        return s;
    }

    // ---- physicallyBased.rlsl:206-228: layer colours and the lobe-selection probabilities ----
    struct Lobes {
        vec3 Cdiff, Cspec;
        float clearCoatScale, diffuseProbability, specularProbability, clearCoatProbability;
    };
    static Lobes computeLobes(vec3 baseColor, float metallic, float specularF0, float clearCoat, float clearCoatNdotV)
    {
        Lobes lb;
        float clearCoatF = F_Schlick(0.04f, clearCoatNdotV); // :210
        lb.clearCoatScale = clearCoatF * clearCoat;
        float clearCoatBottomLayerScale = 1.0f - lb.clearCoatScale;
        lb.Cdiff = (baseColor * (1.0f - metallic)) * clearCoatBottomLayerScale;                         // :214
        lb.Cspec = mix(vec3(specularF0), baseColor, vec3(metallic)) * clearCoatBottomLayerScale;        // :220
        float diffuseLuminance = luminosity(lb.Cdiff);
        float specularLuminance = luminosity(lb.Cspec);
        float probabilityNormalization = 1.0f / greaterThanZero(diffuseLuminance + specularLuminance + lb.clearCoatScale);
        lb.diffuseProbability = diffuseLuminance * probabilityNormalization;
        lb.specularProbability = specularLuminance * probabilityNormalization;
        lb.clearCoatProbability = lb.clearCoatScale * probabilityNormalization;
        return lb;
    }

    // ---- physicallyBased.rlsl:55-331 ----
    void physicallyBased(const Ray &inRay, const Hit &h, const hr_material &M, Ray &nee, Ray &next, Ray &nee2, Ray &nee3, Ray &nee4)
    {
        Ray in = inRay;
        const TriAttr &attr = ctx.attrs[h.prim];
        Surface sf = surface(in, h);
        const uint32_t F = M.flags;
        const bool hasTextures = (F & (HR_MF_HAS_BASE_COLOR_TEXTURE | HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | HR_MF_HAS_EMISSIVE_TEXTURE |
                                       HR_MF_HAS_NORMALMAP | HR_MF_HAS_CLEARCOAT_TEXTURE | HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE |
                                       HR_MF_HAS_CLEARCOAT_NORMALMAP)) != 0;
        const bool useTangentSpace = (F & (HR_MF_HAS_NORMALMAP | HR_MF_HAS_CLEARCOAT_NORMALMAP)) != 0;
        vec3 baseColor(M.base_color[0], M.base_color[1], M.base_color[2]);
        float alpha = 1.0f;
        if (F & HR_MF_HAS_BASE_COLOR_TEXTURE) { // :59-65
            vec4 s = tex(M.base_color_texture, sf.uv);
            baseColor = baseColor * vec3(s.x, s.y, s.z);
            alpha = s.w;
        }
        if ((F & HR_MF_VERTEX_COLORS) && (attr.flags & TF_HAS_COLORS)) baseColor = baseColor * sf.color; // :66-68
        if (F & HR_MF_ALPHA_MASK) { // :70-91 (occlusion rays are resolved inside traceOccluded)
            if (alpha < 1.0f) {
                next = createRay(in, sf.P, h.prim);
                return;
            }
        }
        vec3 N = normalize(sf.normal); // :93
        if (F & HR_MF_DOUBLE_SIDED) { // :95-108
            if (!sf.frontFacing) N = -N;
        } else if (!sf.frontFacing) {
            next = createRay(in, sf.P, h.prim);
            return;
        }
        vec3 clearCoatN = N;
        if (F & HR_MF_HAS_NORMALMAP) { // :112-118
            mat3 nt{normalize(sf.tangent), normalize(sf.bitangent), N};
            vec4 s = tex(M.normalmap, sf.uv);
            vec3 normalTS = vec3(s.x, s.y, s.z) * 2.0f - vec3(1.0f);
            N = normalize(mul(nt, normalTS));
        }
        if (F & HR_MF_HAS_CLEARCOAT_NORMALMAP) { // :120-126
            mat3 nt{normalize(sf.tangent), normalize(sf.bitangent), clearCoatN};
            vec4 s = tex(M.clear_coat_normalmap, sf.uv);
            vec3 normalTS = vec3(s.x, s.y, s.z) * 2.0f - vec3(1.0f);
            clearCoatN = normalize(mul(nt, normalTS));
        }
        mat3 frame = orthonormalFrame(N); // :128
        vec3 V = -in.d;
        float NdotV = saturate(dot(N, V));
        float metallic = M.metallic, roughness = M.roughness, roughnessAlpha = M.roughness_alpha;
        if (F & HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE) { // :135-140  (.bg swizzle: r <- blue, g <- green)
            vec4 s = tex(M.metallic_roughness_texture, sf.uv);
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
        vec3 emissive(M.emissive_color[0], M.emissive_color[1], M.emissive_color[2]);
        if (F & HR_MF_HAS_EMISSIVE_TEXTURE) { // :154-156
            vec4 s = tex(M.emissive_texture, sf.uv);
            emissive = vec3(s.x, s.y, s.z);
        }
        if (pp.enable_visualizer == 1) { // :158-203
            switch (pp.visualizer_mode) {
            case HR_VIS_GEOMETRIC_NORMALS: accumulate4((sf.normal + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_UVS: if (hasTextures) accumulate4(vec3(sf.uv.x, sf.uv.y, 0.0f), 1.0f); break;
            case HR_VIS_TANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.tangent + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BITANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.bitangent + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_NORMALMAP:
                if (F & HR_MF_HAS_NORMALMAP) {
                    vec4 s = tex(M.normalmap, sf.uv);
                    accumulate4(vec3(s.x, s.y, s.z), 1.0f);
                }
                break;
            case HR_VIS_FINAL_NORMALS: accumulate4((N + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BASE_COLOR: accumulate4(baseColor, 1.0f); break;
            case HR_VIS_EMISSIVE: accumulate4(emissive, 1.0f); break;
            case HR_VIS_ROUGHNESS: accumulate4(vec3(roughness), 1.0f); break;
            case HR_VIS_METALLIC: accumulate4(vec3(metallic), 1.0f); break;
            case HR_VIS_CLEARCOAT: accumulate4(vec3(clearCoat), 1.0f); break;
            case HR_VIS_CLEARCOAT_ROUGHNESS: accumulate4(vec3(clearCoatRoughness), 1.0f); break;
            case HR_VIS_SHADER: accumulate4(vec3(1.0f, 0.0f, 0.0f), 1.0f); break;
            case HR_VIS_CLEARCOAT_NORMALMAP:
                if (F & HR_MF_HAS_CLEARCOAT_NORMALMAP) {
                    vec4 s = tex(M.clear_coat_normalmap, sf.uv);
                    accumulate4(vec3(s.x, s.y, s.z), 1.0f);
                }
                break;
            default: break;
            }
            return;
        }
        performAccumulate(in.weight * emissive); // :205
        float clearCoatNdotV = saturate(dot(clearCoatN, V));
        const Lobes lb = computeLobes(baseColor, metallic, M.specular_f0, clearCoat, clearCoatNdotV);
        const vec3 Cdiff = lb.Cdiff, Cspec = lb.Cspec;
        const float clearCoatScale = lb.clearCoatScale;
        const float diffuseProbability = lb.diffuseProbability, specularProbability = lb.specularProbability,
                    clearCoatProbability = lb.clearCoatProbability;
        // clearCoatFrame (:230-233) is computed but never used by the shader.

        const int si = pp.sample_index + in.sequenceIndexOffset;
        { // direct lighting :236-273
            vec2 rand = getSequenceValue(in.sequenceID + in.depth, si);
            vec2 probability = getSequenceValue(in.sequenceID + in.depth + 1, si);
            if (allLights()) {
                // HR_ESTIMATOR_ALL_LIGHTS.  (1) One analytic light, picked among the analytic lights only, lights the WHOLE BSDF
                // (diffuse + specular + clearcoat): a light in a single direction needs no choice of lobe, and the choice is noise.
                LightSample ls = computeLightSample(N, probability.y, sf.P, true);
                if ((ls.type != LIGHT_TYPE_ENVIRONMENT) && (probability.y > 0.0f)) {
                    vec3 f(0.0f);
                    float NdotO = dot(N, ls.dir);
                    if (NdotO > 0.0f) {
                        NdotO = saturate(NdotO);
                        f = (Cdiff / kPI) * NdotO;
                        if (specularProbability > 0.0f)
                            f = f + specularTowards(N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness, ls.dir, NdotO);
                    }
                    float coatNdotO = dot(clearCoatN, ls.dir);
                    if (clearCoatProbability > 0.0f && coatNdotO > 0.0f)
                        f = f + specularTowards(clearCoatN, V, clearCoatNdotV, vec3(clearCoatScale), clearCoatRoughnessAlpha, M.multiscatter_lut,
                                                clearCoatRoughness, ls.dir, saturate(coatNdotO));
                    vec3 reflectance = f * in.weight;
                    reflectance = reflectance / ls.probability;
                    if (dot(reflectance, reflectance) > 0.0f) {
                        Ray r = createRay(in, sf.P, h.prim);
                        r.d = ls.dir;
                        r.weight = reflectance;
                        r.occlusionTest = true;
                        r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                        r.extraT = 0.0f;
                        if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                        nee2 = r;
                    }
                }
                // (2) The environment, always: one MIS-weighted sample per vertex, three at a camera ray's hit, each with its own
                // choice of lobe (the lobe variable shifted by thirds) and its own sequence values
                if (ctx.lights.env_enabled) {
                    const bool mis = envMis();
                    const int nSamples = (mis && in.depth == 0) ? kPrimaryEnvSamples : 1;
                    const float nEnv = (float)nSamples;
                    for (int j = 0; j < nSamples; ++j) {
                        Ray &out = (j == 0) ? nee : ((j == 1) ? nee3 : nee4);
                        float u = probability.x + (float)j * 0.333333343f;
                        if (u > 1.0f) u = u - 1.0f;
                        if (u <= diffuseProbability) {
                            if (mis)
                                envMisDiffuse(in, sf.P, h.prim, N, Cdiff, diffuseProbability, nEnv, rand, frame, out, next, j);
                            else
                                indirectDiffuseSample(in, sf.P, h.prim, N, Cdiff, diffuseProbability, 1.0f, rand, frame, MISS_ENV, out, next);
                        } else if (u <= (diffuseProbability + clearCoatProbability)) {
                            if (mis)
                                envMisSpecular(in, sf.P, h.prim, clearCoatN, V, clearCoatNdotV, vec3(clearCoatScale), clearCoatRoughnessAlpha,
                                               M.multiscatter_lut, clearCoatRoughness, clearCoatProbability, nEnv, rand, frame, out, next, j);
                            else
                                indirectSpecularSample(in, sf.P, h.prim, clearCoatN, V, clearCoatNdotV, vec3(clearCoatScale), clearCoatRoughnessAlpha,
                                                       M.multiscatter_lut, clearCoatRoughness, clearCoatProbability, 1.0f, rand, frame, MISS_ENV, out, next);
                        } else if (u <= (diffuseProbability + clearCoatProbability + specularProbability)) {
                            if (mis)
                                envMisSpecular(in, sf.P, h.prim, N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness, specularProbability,
                                               nEnv, rand, frame, out, next, j);
                            else
                                indirectSpecularSample(in, sf.P, h.prim, N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness,
                                                       specularProbability, 1.0f, rand, frame, MISS_ENV, out, next);
                        }
                    }
                }
            } else if (probability.x <= diffuseProbability) {
                directDiffuseSample(in, sf.P, h.prim, N, Cdiff, diffuseProbability, probability.y, rand, frame, nee, next);
            } else if (probability.x <= (diffuseProbability + clearCoatProbability)) {
                directSpecularSample(in, sf.P, h.prim, clearCoatN, V, clearCoatNdotV, vec3(clearCoatScale), clearCoatRoughnessAlpha, M.multiscatter_lut,
                                     clearCoatRoughness, clearCoatProbability, probability.y, rand, frame, nee, next);
            } else if (probability.x <= (diffuseProbability + clearCoatProbability + specularProbability)) {
                directSpecularSample(in, sf.P, h.prim, N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness, specularProbability,
                                     probability.y, rand, frame, nee, next);
            }
        }
        if (in.depth < pp.max_ray_depth) { // :277-330
            if (in.depth > 3) {
                vec2 rand = getSequenceValue(in.sequenceID + in.depth + 2, si);
                float probability = fmax_(in.weight.x, fmax_(in.weight.y, in.weight.z));
                if (rand.x >= probability) return;
                in.weight = in.weight / probability;
            }
            vec2 rand = getSequenceValue(in.sequenceID + in.depth + 3, si);
            vec2 probability = getSequenceValue(in.sequenceID + in.depth + 4, si);
            Ray dummyNee; // indirect samples use rl_NullPrimitive: never an occlusion ray
            if (probability.x <= diffuseProbability) {
                indirectDiffuseSample(in, sf.P, h.prim, N, Cdiff, diffuseProbability, 1.0f, rand, frame, MISS_NONE, dummyNee, next);
            } else if (probability.x <= (diffuseProbability + clearCoatProbability)) {
                indirectSpecularSample(in, sf.P, h.prim, clearCoatN, V, clearCoatNdotV, vec3(clearCoatScale), clearCoatRoughnessAlpha,
                                       M.multiscatter_lut, clearCoatRoughness, clearCoatProbability, 1.0f, rand, frame, MISS_NONE, dummyNee, next);
            } else if (probability.x <= (diffuseProbability + clearCoatProbability + specularProbability)) {
                indirectSpecularSample(in, sf.P, h.prim, N, V, NdotV, Cspec, roughnessAlpha, M.multiscatter_lut, roughness, specularProbability,
                                       1.0f, rand, frame, MISS_NONE, dummyNee, next);
            }
        }
    }

    // ---- glass.rlsl ----
    void indirectSpecularGlassSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 weight, vec3 baseColor,
                                     float roughnessAlpha, float materialRoughnessAlpha, float optionalLightSampleProbability, vec2 rand,
                                     const mat3 &frame, int missKind, Ray &nee, Ray &next) // :47-81
    {
        vec3 localSpaceI = mulT(frame, I);
        vec3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float IdotH = saturate(dot(I, H));
        vec3 O = normalize(2.0f * IdotH * H - I);
        float NdotO = dot(N, O);
        if (NdotO > 0.0f) {
            NdotO = saturate(NdotO);
            float NdotH = saturate(dot(N, H));
            float G = G2_Smith_GGX(NdotO, NdotI, materialRoughnessAlpha); // :63 uses Material.roughnessAlpha
            vec3 reflectance = baseColor * ((G * IdotH) / (NdotH * NdotI));
            reflectance = reflectance * weight;
            reflectance = reflectance / optionalLightSampleProbability;
            if (dot(reflectance, reflectance) > 1e-5f) {
                Ray r = createRay(in, P, prim);
                r.d = O;
                r.weight = reflectance;
                r.occlusionTest = (missKind != MISS_NONE);
                r.missKind = missKind, r.missIdx = 0;
                r.extraT = 0.0f;
                r.coneG = widenCone(in.coneG, roughnessAlpha);
                if (missKind == MISS_ENV && !ctx.lights.env_enabled) return;
                emit(r, nee, next);
            }
        }
    }
    // HR_ESTIMATOR_ENV_MIS / HR_ESTIMATOR_ALL_LIGHTS on glass (this repo's contract, include/hrcore.h; the reference lobe-samples the
    // map here: glass.rlsl:83-129 -> :47-81): the reflection's next-event ray towards the environment is drawn from the visible-normal
    // lobe or from the map's importance table, half the time each, and weighted with the balance heuristic.  BRDF x cos is the one
    // glass.rlsl itself uses for an analytic light (:104-109: D G2 / (4 N.I) x baseColor, no Fresnel factor: the reflection branch was
    // chosen with the Fresnel probability), the lobe's density D G1 / (4 N.I).  Own selection variable: sequence ID + depth + 5.
    void envMisGlass(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 weight, vec3 baseColor, float roughnessAlpha, float envProbability,
                     vec2 rand, const mat3 &frame, Ray &nee, Ray &next)
    {
        const vec2 sel = getSequenceValue(in.sequenceID + in.depth + 5, pp.sample_index + in.sequenceIndexOffset);
        vec3 O, H;
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
        vec3 reflectance = specular * baseColor;
        reflectance = reflectance * weight;
        reflectance = reflectance / greaterThanZero(0.5f * pLobe + 0.5f * pMap);
        reflectance = reflectance / envProbability;
        if (dot(reflectance, reflectance) > 0.0f) {
            Ray r = createRay(in, P, prim);
            r.d = O;
            r.weight = reflectance;
            r.occlusionTest = true;
            r.missKind = MISS_ENV, r.missIdx = 0;
            r.extraT = 0.0f;
            emit(r, nee, next);
        }
    }
    void directSpecularGlassSample(const Ray &in, vec3 P, int prim, vec3 N, vec3 I, float NdotI, vec3 weight, vec3 baseColor, float roughnessAlpha,
                                   float materialRoughnessAlpha, float lightProbability, vec2 rand, const mat3 &frame, Ray &nee,
                                   Ray &next, Ray &nee2) // :83-129
    {
        const bool both = allLights(); // HR_ESTIMATOR_ALL_LIGHTS: an analytic light (-> nee2) AND the environment (-> nee), not one of them
        LightSample ls = computeLightSample(N, lightProbability, P, both);
        if (ls.type != LIGHT_TYPE_ENVIRONMENT) {
            float NdotO = dot(N, ls.dir);
            if (NdotO > 0.0f) {
                NdotO = saturate(NdotO);
                vec3 H = normalize(I + ls.dir);
                float NdotH = saturate(dot(N, H));
                float D = D_GGX(NdotH, roughnessAlpha);
                float G = G2_Smith_GGX(NdotO, NdotI, roughnessAlpha);
                float specular = (D * G) / greaterThanZero(4.0f * NdotI);
                vec3 reflectance = specular * baseColor;
                reflectance = reflectance * weight;
                reflectance = reflectance / ls.probability;
                if (dot(reflectance, reflectance) > 1e-5f) {
                    Ray r = createRay(in, P, prim);
                    r.d = ls.dir;
                    r.weight = reflectance;
                    r.occlusionTest = true;
                    r.missKind = ls.missKind, r.missIdx = ls.missIdx;
                    r.extraT = 0.0f;
                    if (ls.type == LIGHT_TYPE_POINT || ls.type == LIGHT_TYPE_SPOT) r.maxT = ls.maxDistance;
                    if (both)
                        nee2 = r;
                    else
                        emit(r, nee, next);
                }
            }
        }
        if (both) {
            if (ctx.lights.env_enabled) {
                if (envMis())
                    envMisGlass(in, P, prim, N, I, NdotI, weight, baseColor, roughnessAlpha, 1.0f, rand, frame, nee, next);
                else
                    indirectSpecularGlassSample(in, P, prim, N, I, NdotI, weight, baseColor, roughnessAlpha, materialRoughnessAlpha, 1.0f, rand, frame,
                                                MISS_ENV, nee, next);
            }
        } else if (ls.type != LIGHT_TYPE_ENVIRONMENT) {
        } else if (ls.probability > 0.0f) {
            if (envMis())
                envMisGlass(in, P, prim, N, I, NdotI, weight, baseColor, roughnessAlpha, ls.probability, rand, frame, nee, next);
            else
                indirectSpecularGlassSample(in, P, prim, N, I, NdotI, weight, baseColor, roughnessAlpha, materialRoughnessAlpha, ls.probability, rand,
                                            frame, MISS_ENV, nee, next);
        }
    }
    void glass(const Ray &in, const Hit &h, const hr_material &M, Ray &nee, Ray &next, Ray &nee2) // :138-280
    {
        const TriAttr &attr = ctx.attrs[h.prim];
        Surface sf = surface(in, h);
        const uint32_t F = M.flags;
        const bool hasTextures = (F & (HR_MF_HAS_BASE_COLOR_TEXTURE | HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | HR_MF_HAS_NORMALMAP)) != 0;
        const bool useTangentSpace = (F & HR_MF_HAS_NORMALMAP) != 0;
        vec3 N = normalize(sf.normal);
        float nIn = 1.0f;
        float nOut = M.ior;
        vec3 weight = in.weight;
        if (F & HR_MF_HAS_NORMALMAP) { // :145-151
            mat3 nt{normalize(sf.tangent), normalize(sf.bitangent), N};
            vec4 s = tex(M.normalmap, sf.uv);
            vec3 normalTS = vec3(s.x, s.y, s.z) * 2.0f - vec3(1.0f);
            N = normalize(mul(nt, normalTS));
        }
        vec3 baseColor(M.base_color[0], M.base_color[1], M.base_color[2]);
        if (F & HR_MF_HAS_BASE_COLOR_TEXTURE) { // :154-156
            vec4 s = tex(M.base_color_texture, sf.uv);
            baseColor = baseColor * vec3(s.x, s.y, s.z);
        }
        if ((F & HR_MF_VERTEX_COLORS) && (attr.flags & TF_HAS_COLORS)) baseColor = baseColor * sf.color;
        if (!sf.frontFacing) { // :161-167, beersLaw :131-136
            N = -N;
            nIn = M.ior;
            nOut = 1.0f;
            vec3 absorption = vec3(1.0f) - baseColor;
            float rayLength = h.t;
            vec3 e = absorption * M.density * -rayLength;
            weight = in.weight * vec3(exp_(e.x), exp_(e.y), exp_(e.z));
        }
        float roughness = M.roughness, roughnessAlpha = M.roughness_alpha;
        if (F & HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE) { // :171-176
            vec4 s = tex(M.metallic_roughness_texture, sf.uv);
            roughness = roughness * s.y;
            roughnessAlpha = roughness * roughness;
        }
        if (pp.enable_visualizer == 1) { // :179-210
            switch (pp.visualizer_mode) {
            case HR_VIS_GEOMETRIC_NORMALS: accumulate4((sf.normal + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_FINAL_NORMALS: accumulate4((N + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BASE_COLOR: accumulate4(baseColor, 1.0f); break;
            case HR_VIS_ROUGHNESS: accumulate4(vec3(roughness), 1.0f); break;
            case HR_VIS_SHADER: accumulate4(vec3(0.0f, 1.0f, 0.0f), 1.0f); break;
            case HR_VIS_UVS: if (hasTextures) accumulate4(vec3(sf.uv.x, sf.uv.y, 0.0f), 1.0f); break;
            case HR_VIS_TANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.tangent + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_BITANGENTS: if (hasTextures && useTangentSpace) accumulate4((sf.bitangent + vec3(1.0f)) * 0.5f, 1.0f); break;
            case HR_VIS_NORMALMAP:
                if (F & HR_MF_HAS_NORMALMAP) {
                    vec4 s = tex(M.normalmap, sf.uv);
                    accumulate4(vec3(s.x, s.y, s.z), 1.0f);
                }
                break;
            default: break;
            }
            return;
        }
        mat3 frame = orthonormalFrame(N); // :212
        vec3 I = -in.d;
        float eta = nIn / nOut;
        vec3 localSpaceI = mulT(frame, I);
        const int si = pp.sample_index + in.sequenceIndexOffset;
        vec2 rand = getSequenceValue(in.sequenceID + in.depth, si);
        vec3 H = mul(frame, sampleVisibleGGX(localSpaceI, rand.x, rand.y, roughnessAlpha));
        float HdotI = saturate(dot(H, I));
        vec2 refractProbability = getSequenceValue(in.sequenceID + in.depth + 1, si);
        float Fr = F_Fresnel(eta, HdotI);
        float NdotI = saturate(dot(N, I));
        if (!sf.frontFacing) refractProbability = vec2{refractProbability.x, 0.0f}; // :227-231
        if (refractProbability.y < (1.0f - Fr)) { // :234-256
            vec3 O = normalize(refract(-I, H, eta));
            float NdotO = fabsf(dot(N, O));
            float G2 = G2_Smith_GGX(NdotI, NdotO, roughnessAlpha);
            float G1 = G1_Smith_GGX(NdotI, roughnessAlpha);
            vec3 transmission = baseColor * G2 / greaterThanZero(G1);
            transmission = transmission * weight;
            if (dot(transmission, transmission) > 1e-5f && in.depth < pp.max_ray_depth) {
                Ray r = createRay(in, sf.P, h.prim);
                r.d = O;
                r.weight = transmission;
                r.occlusionTest = false;
                r.extraT = 0.0f;
                r.missKind = ctx.lights.env_enabled ? MISS_ENV : MISS_NONE;
                r.missIdx = 0;
                r.coneG = widenCone(in.coneG, roughnessAlpha);
                next = r;
            }
        } else { // :257-279
            {
                rand = getSequenceValue(in.sequenceID + in.depth + 2, si);
                directSpecularGlassSample(in, sf.P, h.prim, N, I, NdotI, weight, baseColor, roughnessAlpha, M.roughness_alpha, refractProbability.x,
                                          rand, frame, nee, next, nee2);
            }
            if (in.depth < pp.max_ray_depth) {
                if (in.depth > 3) {
                    vec2 rr = getSequenceValue(in.sequenceID + in.depth + 3, si);
                    float probability = fmax_(weight.x, fmax_(weight.y, weight.z));
                    if (rr.x >= probability) return;
                    weight = weight / probability;
                }
                rand = getSequenceValue(in.sequenceID + in.depth + 4, si);
                Ray dummyNee;
                indirectSpecularGlassSample(in, sf.P, h.prim, N, I, NdotI, weight, baseColor, roughnessAlpha, M.roughness_alpha, 1.0f, rand, frame,
                                            MISS_NONE, dummyNee, next);
            }
        }
    }

    // ---- perspective.rlsl:39-93 (frame shader) ----
    static float random(float sx, float sy) { return fract(sin_(sx * 12.9898f + sy * 78.233f) * 43758.5453123f); } // utility.rlsl:15-18
    bool generatePrimary(int x, int y, Ray &out)
    {
        const float W = (float)ctx.W, H = (float)ctx.H;
        const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;
        if (pp.interactive_mode != 0) { // :42-57 — the 3x3 block texture (RL_NEAREST / RL_REPEAT); the reference shuffles its
            // texels with std::random_device (PassGenerator.cpp:276-278): the table is an input here, unshuffled by default.
            const int bsx = pp.block_size[0], bsy = pp.block_size[1];
            const int bix = (int)(fcx - 0.5f) / bsx, biy = (int)(fcy - 0.5f) / bsy;
            float randX = random((float)bix, (float)biy);
            float randY = random((float)biy, (float)bix);
            float su = (1.0f / (float)bsx) * (float)pp.current_block_pixel[0] + randX;
            float sv = (1.0f / (float)bsy) * (float)pp.current_block_pixel[1] + randY;
            int tx = (int)floorf(su * (float)bsx) % bsx, ty = (int)floorf(sv * (float)bsy) % bsy;
            // texel (tx,ty) of the row-major coords list holds vec3(row=ty... ) : identity layout -> (ty, tx)
            int sampleX = ty, sampleY = tx;
            // PassGenerator.cpp:267-294: the list may be shuffled; the host supplies it (ora_interactive_blocks_set)
            if (ctx.blockNx == bsx && ctx.blockNy == bsy) sampleX = ctx.blockCoords[2 * (ty * bsx + tx)], sampleY = ctx.blockCoords[2 * (ty * bsx + tx) + 1];
            int thisX = (int)(fcx - 0.5f) % bsx, thisY = (int)(fcy - 0.5f) % bsy;
            if (thisX != sampleX || thisY != sampleY) return false;
        }
        px[3] = px[3] + 1.0f; // :60 accumulate(vec4(0,0,0,1))
        int sequenceID = (int)floorf(random(fcx / W, fcy / H) * (float)ctx.nSeq); // :62
        // :64 — row stride is the frame HEIGHT and the coords are pixel centres;
        // the reference reads out of bounds on square/tall frames, defined here as wrap modulo W*H.
        int offIdx = (int)(fcy * H + fcx);
        offIdx = offIdx % (int)ctx.seqOffsets.size();
        float rnd = ctx.seqOffsets[offIdx].x;
        int sequenceIndex = (int)floorf(rnd * pp.max_sample_index); // :65
        vec2 sampleOffset = getSequenceValue(sequenceID, pp.sample_index + sequenceIndex);
        float spx = (fcx - 0.5f) + sampleOffset.x, spy = (fcy - 0.5f) + sampleOffset.y;
        float u = spx / W, v = spy / H;
        float cx = (2.0f * u - 1.0f) * pp.aspect_ratio * pp.fov_tan;  // :72
        float cy = (1.0f - 2.0f * v) * pp.fov_tan * -1.0f;            // :73
        vec3 dirCameraSpace = normalize(vec3(cx, cy, -1.0f));
        vec3 focalPoint = pp.focus_distance * dirCameraSpace; // :77
        // :78 — the aperture index ignores sequenceIndex and is not wrapped in the reference; wrapped here.
        int apIdx = (sequenceID * ctx.seqLen + pp.sample_index) % (ctx.nSeq * ctx.seqLen);
        vec2 ap = ctx.aperture[apIdx];
        float ax = ((ap.x * 2.0f) - 1.0f) * pp.aperture_radius, ay = ((ap.y * 2.0f) - 1.0f) * pp.aperture_radius;
        vec3 origin(ax, ay, 0.0f);
        vec3 dir = focalPoint - origin;
        out = Ray();
        out.o = xformPoint(pp.view_matrix, origin);               // :84
        out.d = normalize(xformVector(pp.view_matrix, dir));      // :85 (OpenRL normalises emitted directions)
        out.missKind = ctx.lights.env_enabled ? MISS_ENV : MISS_NONE;
        out.weight = vec3(1.0f);
        out.sequenceID = sequenceID;
        out.sequenceIndexOffset = sequenceIndex;
        out.extraT = 0.0f;
        out.depth = 0;
        out.valid = true;
        out.coneW = 0.0f;
        out.coneG = 2.0f * pp.fov_tan / H;
        return true;
    }

    // Advance the ray's cone to the hit and derive the footprint's level offset on this triangle (HR_TEXTURE_LOD_CONE)
    void setFootprint(Ray &in, const Hit &h)
    {
        const float hitW = in.coneW + in.coneG * h.t;
        in.coneW = hitW;
        lodBase = -1e30f;
        if (pp.texture_lod != HR_TEXTURE_LOD_CONE || ctx.texDensity.size() != ctx.tris.size() || !(hitW > 0.0f)) return;
        const TriAttr &a = ctx.attrs[h.prim];
        const vec3 normal = lerp3(a.n, 1.0f - h.u - h.v, h.u, h.v);
        const float cosT = fmax_(fabsf(dot(in.d, normal)), 0.1f);
        lodBase = ctx.texDensity[h.prim] + log_(hitW / cosT) * 1.4426950408889634f;
    }

    void tracePath(int x, int y)
    {
        Ray ray;
        if (!generatePrimary(x, y, ray)) return;
        st.paths++;
        while (ray.valid) {
            ray.coneW = coneQuant(ray.coneW), ray.coneG = coneQuant(ray.coneG); // (the product's ray record)
            st.rays_closest++;
            Hit h = traceClosest(ctx, ray.o, ray.d, ctx.rayEps, ray.maxT, ray.srcPrim, &tc, ctx.brute);
            if (h.prim < 0) {
                // a ray that hits nothing runs its defaultPrimitive's shader (none for rl_NullPrimitive)
                if (ray.missKind == MISS_ENV) environmentLight(ray.d, ray.weight);
                break;
            }
            Ray nee, next, extra[3];
            setFootprint(ray, h);
            const int mid = ctx.attrs[h.prim].material;
            if (mid >= 0 && mid < (int)ctx.materials.size()) {
                const hr_material &M = ctx.materials[mid];
                if (M.type == HR_MAT_GLASS) {
                    st.shaded_hits++;
                    glass(ray, h, M, nee, next, extra[0]);
                } else if (M.type == HR_MAT_PBR) {
                    st.shaded_hits++;
                    physicallyBased(ray, h, M, nee, next, extra[0], extra[1], extra[2]);
                }
            }
            if (nee.valid) {
                st.rays_any++;
                if (!traceOccluded(ctx, nee.o, nee.d, ctx.rayEps, nee.maxT, nee.srcPrim, &tcAny, ctx.brute)) lightShader(nee, nee.maxT);
            }
            for (int j = 0; j < 3; ++j) { // HR_ESTIMATOR_ALL_LIGHTS: the analytic-light ray and the extra environment samples
                const Ray &x = extra[j];
                if (!x.valid) continue;
                st.rays_any++;
                part = j + 1;
                if (!traceOccluded(ctx, x.o, x.d, ctx.rayEps, x.maxT, x.srcPrim, &tcAny, ctx.brute)) lightShader(x, x.maxT);
                part = 0;
            }
            ray = next;
        }
        if (pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) // the partial sums meet, in order, then the sample joins the frame
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) px[k] = px[k] + pxX[j][k];
        for (int k = 0; k < 4; ++k) fbPixel[k] = fbPixel[k] + px[k];
    }
};

void renderPass(Context &ctx, const hr_pass_params &pp, int nThreads)
{
    if (pp.estimator == HR_ESTIMATOR_ENV_MIS || pp.estimator == HR_ESTIMATOR_ALL_LIGHTS) buildEnvTable(ctx);
    if (pp.texture_lod == HR_TEXTURE_LOD_CONE) buildTextureLod(ctx);
    const int W = ctx.W, H = ctx.H, tile = ctx.tile > 0 ? ctx.tile : 32;
    const int tilesX = (W + tile - 1) / tile;
    if (nThreads <= 0) nThreads = omp_get_max_threads();
    std::vector<hr_pass_stats> stats(nThreads);
    std::vector<TraceCounters> tcs(nThreads), tcsAny(nThreads);
    const int tilesY = (H + tile - 1) / tile;
    std::vector<int> owned;
    for (int t = ctx.rank; t < tilesX * tilesY; t += ctx.world) owned.push_back(t);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nThreads)
    for (int k = 0; k < (int)owned.size(); ++k) {
        const int tid = omp_get_thread_num();
        const int tx = owned[k] % tilesX, ty = owned[k] / tilesX;
        hr_pass_stats local{};
        TraceCounters tcl, tcla;
        for (int y = ty * tile; y < (ty + 1) * tile && y < H; ++y) {
            for (int x = tx * tile; x < (tx + 1) * tile && x < W; ++x) {
                Shader sh(ctx, pp, &ctx.fb[((size_t)y * W + x) * 4], local);
                sh.tracePath(x, y);
                tcl.nodeVisits += sh.tc.nodeVisits, tcl.triTests += sh.tc.triTests;
                tcla.nodeVisits += sh.tcAny.nodeVisits, tcla.triTests += sh.tcAny.triTests;
            }
        }
        stats[tid].paths += local.paths, stats[tid].rays_closest += local.rays_closest, stats[tid].rays_any += local.rays_any;
        stats[tid].shaded_hits += local.shaded_hits, stats[tid].accumulates += local.accumulates;
        tcs[tid].nodeVisits += tcl.nodeVisits, tcs[tid].triTests += tcl.triTests;
        tcsAny[tid].nodeVisits += tcla.nodeVisits, tcsAny[tid].triTests += tcla.triTests;
    }
    for (int i = 0; i < nThreads; ++i) {
        ctx.stats.paths += stats[i].paths;
        ctx.stats.rays_closest += stats[i].rays_closest;
        ctx.stats.rays_any += stats[i].rays_any;
        ctx.stats.shaded_hits += stats[i].shaded_hits;
        ctx.stats.accumulates += stats[i].accumulates;
        ctx.stats.node_visits += tcs[i].nodeVisits + tcsAny[i].nodeVisits;
        ctx.stats.tri_tests += tcs[i].triTests + tcsAny[i].triTests;
        ctx.stats.node_visits_any += tcsAny[i].nodeVisits;
        ctx.stats.tri_tests_any += tcsAny[i].triTests;
    }
}

} // namespace ora

// ---- function-level probes for the known-answer tests (tests/test_oracle_kat.py): each evaluates ONE restated RLSL
// function on caller-supplied inputs, so that a transcription error shows against an independently written formula.
struct hr_ctx;
namespace ora {
Context &contextOf(hr_ctx *ctx); // oracle_api.cpp
}
extern "C" {
using namespace ora;

// which: 0 F_Fresnel(eta, cosI)  1 F_Schlick(f0, cos)  2 D_GGX(NdotH, alpha)  3 G1_Smith_GGX(NdotI, alpha)
//        4 G2_Smith_GGX(NdotO, NdotI, alpha)  5 luminosity(rgb)  6 smoothstep(e0, e1, x)
float ora_kat_scalar(int which, float a, float b, float c)
{
    switch (which) {
    case 0: return Shader::F_Fresnel(a, b);
    case 1: return Shader::F_Schlick(a, b);
    case 2: return Shader::D_GGX(a, b);
    case 3: return Shader::G1_Smith_GGX(a, b);
    case 4: return Shader::G2_Smith_GGX(a, b, c);
    case 5: return Shader::luminosity(vec3(a, b, c));
    case 6: return smoothstep(a, b, c);
    default: return NAN;
    }
}
// utility.rlsl:109-139 (y-up local space in, y-up microfacet normal out)
void ora_kat_sample_visible_ggx(const float v[3], float u1, float u2, float alpha, float out[3])
{
    vec3 h = Shader::sampleVisibleGGX(vec3(v[0], v[1], v[2]), u1, u2, alpha);
    out[0] = h.x, out[1] = h.y, out[2] = h.z;
}
void ora_kat_cosine_sample(float u1, float u2, float out[3]) // utility.rlsl:64-75
{
    vec3 d = Shader::cosineWeightedSample(u1, u2);
    out[0] = d.x, out[1] = d.y, out[2] = d.z;
}
void ora_kat_frame(const float n[3], float out[9]) // utility.rlsl:43-60, columns c0 c1 c2
{
    mat3 m = Shader::orthonormalFrame(vec3(n[0], n[1], n[2]));
    const vec3 c[3] = {m.c0, m.c1, m.c2};
    for (int k = 0; k < 3; ++k) out[3 * k] = c[k].x, out[3 * k + 1] = c[k].y, out[3 * k + 2] = c[k].z;
}
void ora_kat_refract(const float i[3], const float n[3], float eta, float out[3]) // GLSL refract as glass.rlsl:235 uses it
{
    vec3 r = refract(vec3(i[0], i[1], i[2]), vec3(n[0], n[1], n[2]), eta);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
// physicallyBased.rlsl:206-228 — out: Cdiff[3] Cspec[3] clearCoatScale pDiffuse pSpecular pClearCoat
void ora_kat_pbr_lobes(const float baseColor[3], float metallic, float specularF0, float clearCoat, float clearCoatNdotV, float out[10])
{
    Shader::Lobes lb = Shader::computeLobes(vec3(baseColor[0], baseColor[1], baseColor[2]), metallic, specularF0, clearCoat, clearCoatNdotV);
    out[0] = lb.Cdiff.x, out[1] = lb.Cdiff.y, out[2] = lb.Cdiff.z, out[3] = lb.Cspec.x, out[4] = lb.Cspec.y, out[5] = lb.Cspec.z;
    out[6] = lb.clearCoatScale, out[7] = lb.diffuseProbability, out[8] = lb.specularProbability, out[9] = lb.clearCoatProbability;
}
// lightSampling.rlsl:11-161 with the ctx's light block — out: type, missKind, index | probability, maxDistance, dir[3]
void ora_kat_light_sample(hr_ctx *ctx, const float N[3], const float P[3], float xi, int outI[3], float outF[5])
{
    Context &c = contextOf(ctx);
    hr_pass_params pp{};
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    Shader::LightSample ls = sh.computeLightSample(vec3(N[0], N[1], N[2]), xi, vec3(P[0], P[1], P[2]));
    outI[0] = ls.type, outI[1] = ls.missKind, outI[2] = ls.missIdx;
    outF[0] = ls.probability, outF[1] = ls.maxDistance, outF[2] = ls.dir.x, outF[3] = ls.dir.y, outF[4] = ls.dir.z;
}
// HR_ESTIMATOR_ENV_MIS: light pick with the map weighed by its mean luminosity; mean luminosity of the table; density and a sample
void ora_kat_light_sample_mis(hr_ctx *ctx, const float N[3], const float P[3], float xi, int outI[3], float outF[5])
{
    Context &c = contextOf(ctx);
    buildEnvTable(c);
    hr_pass_params pp{};
    pp.estimator = HR_ESTIMATOR_ENV_MIS;
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    Shader::LightSample ls = sh.computeLightSample(vec3(N[0], N[1], N[2]), xi, vec3(P[0], P[1], P[2]));
    outI[0] = ls.type, outI[1] = ls.missKind, outI[2] = ls.missIdx;
    outF[0] = ls.probability, outF[1] = ls.maxDistance, outF[2] = ls.dir.x, outF[3] = ls.dir.y, outF[4] = ls.dir.z;
}
float ora_kat_env_mean_luminosity(hr_ctx *ctx)
{
    Context &c = contextOf(ctx);
    buildEnvTable(c);
    return c.env.meanLum;
}
float ora_kat_env_pdf(hr_ctx *ctx, const float dir[3])
{
    Context &c = contextOf(ctx);
    buildEnvTable(c);
    hr_pass_params pp{};
    pp.estimator = HR_ESTIMATOR_ENV_MIS;
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    return sh.envPdf(vec3(dir[0], dir[1], dir[2]));
}
// out: direction[3], density per solid angle at that direction
void ora_kat_env_sample(hr_ctx *ctx, float u1, float u2, float out[4])
{
    Context &c = contextOf(ctx);
    buildEnvTable(c);
    hr_pass_params pp{};
    pp.estimator = HR_ESTIMATOR_ENV_MIS;
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    vec3 d = sh.sampleEnv(u1, u2);
    out[0] = d.x, out[1] = d.y, out[2] = d.z, out[3] = sh.envPdf(d);
}
// HR_TEXTURE_LOD_CONE: trilinear lookup of texture `tex` at level `lambda` (builds the mip chain); out = rgba
void ora_kat_texture_lod(hr_ctx *ctx, int tex, float u, float v, float lambda, float out[4])
{
    Context &c = contextOf(ctx);
    buildTextureLod(c);
    const vec4 r = sampleTextureLod(c.textures[tex], u, v, lambda);
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
// number of levels of texture `tex` and a copy of level `level` (>= 1) into out (may be null); returns the level count
int ora_kat_texture_level(hr_ctx *ctx, int tex, int level, float *out)
{
    Context &c = contextOf(ctx);
    buildTextureLod(c);
    const Texture &t = c.textures[tex];
    if (out && level >= 1 && level < t.nLevels) {
        size_t off = 0;
        int w = t.w, h = t.h;
        for (int l = 1; l <= level; ++l) {
            if (l > 1) off += (size_t)w * h * t.c;
            w = w / 2 < 1 ? 1 : w / 2, h = h / 2 < 1 ? 1 : h / 2;
        }
        std::memcpy(out, &t.mips[off], sizeof(float) * (size_t)w * h * t.c);
    }
    return t.nLevels;
}
// the footprint's level offset (Shader::setFootprint) for a ray with cone (coneW, coneG) hitting triangle `prim` at distance t with
// barycentrics (u, v); out[0] = lodBase, out[1] = the cone width at the hit, out[2] = texDensity[prim]
void ora_kat_footprint(hr_ctx *ctx, int prim, const float dir[3], float coneW, float coneG, float t, float u, float v, float out[3])
{
    Context &c = contextOf(ctx);
    buildTextureLod(c);
    hr_pass_params pp{};
    pp.texture_lod = HR_TEXTURE_LOD_CONE;
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    Ray r;
    r.d = vec3(dir[0], dir[1], dir[2]), r.coneW = coneW, r.coneG = coneG;
    Hit h;
    h.prim = prim, h.t = t, h.u = u, h.v = v;
    sh.setFootprint(r, h);
    out[0] = sh.lodBase, out[1] = r.coneW, out[2] = c.texDensity[prim];
}
// the light / miss shaders (environmentLight / directionalLight / pointLight / spotLight.rlsl) for a ray that reached its light:
// missKind 1 env 2 directional 3 point 4 spot; out = the value performAccumulate added (clamped by maxChannelValue)
void ora_kat_light_shader(hr_ctx *ctx, int missKind, int index, const float dir[3], const float weight[3], float t, float extraT,
                          float maxChannelValue, float out[3])
{
    Context &c = contextOf(ctx);
    hr_pass_params pp{};
    pp.max_channel_value = maxChannelValue;
    hr_pass_stats st{};
    float px[4] = {0, 0, 0, 0};
    Shader sh(c, pp, px, st);
    Ray r;
    r.d = vec3(dir[0], dir[1], dir[2]), r.weight = vec3(weight[0], weight[1], weight[2]);
    r.missKind = missKind, r.missIdx = index, r.extraT = extraT;
    sh.lightShader(r, t);
    out[0] = sh.px[0], out[1] = sh.px[1], out[2] = sh.px[2];
}
} // extern "C"
