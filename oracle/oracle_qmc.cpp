// oracle_qmc.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// Restatement of the reference's sample-table generators
// (/root/reference/Source/Utility/Random.h, BlueNoise.h, Hash.h) and of the
// multiscatter LUT generator (Source/HeatrayRenderer/Materials/MultiScatterUtil.cpp:20-139).
// Pinned by tests/golden/ref_vectors.npz, which holds the outputs of the
// reference's own headers compiled in the build container
// (oracle/ref/gen_golden.cpp) and the reference's shipped multiscatter_lut.tiff.
// PARITY STATUS: PINNED — bit-exact against those vectors (tests/test_oracle_qmc.py); LUT within 2e-6 of the TIFF.
#include "oracle_internal.h"

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

namespace ora {

// Random.h:26-29 — uint32_t(f * float(UINT32_MAX)).  float(UINT32_MAX) rounds to
// 2^32; for f == 1.0 the product is 2^32, which x86-64 converts through a 64-bit
// integer and truncates to 0.  Stated explicitly so every platform agrees.
static inline uint32_t toUint32(float f) { return (uint32_t)(uint64_t)(int64_t)(f * 4294967296.0f); }
// Random.h:31-34 — float(u) * (1.0f / float(UINT32_MAX)) == float(u) * 2^-32.
static inline float toNormalizedFloat(uint32_t u) { return (float)u * (1.0f / 4294967296.0f); }

// Random.h:36-45
uint32_t burleyHash(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
// Random.h:47-50
static inline uint32_t burleyHashCombine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
// Random.h:52-60
uint32_t laineKarrasPermutation(uint32_t x, uint32_t seed)
{
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return x;
}
// Random.h:62-70
uint32_t reverseBits(uint32_t v)
{
    uint32_t b = (v << 16) | (v >> 16);
    b = ((b & 0x55555555u) << 1) | ((b & 0xAAAAAAAAu) >> 1);
    b = ((b & 0x33333333u) << 2) | ((b & 0xCCCCCCCCu) >> 2);
    b = ((b & 0x0F0F0F0Fu) << 4) | ((b & 0xF0F0F0F0u) >> 4);
    b = ((b & 0x00FF00FFu) << 8) | ((b & 0xFF00FF00u) >> 8);
    return b;
}
// Random.h:72-78
uint32_t nestedUniformScramble(uint32_t x, uint32_t seed)
{
    x = reverseBits(x);
    x = laineKarrasPermutation(x, seed);
    x = reverseBits(x);
    return x;
}

// Random.h:225-250 — Sobol dimension 0 is the bit reversal (direction numbers
// 2^(31-bit)); dimension 1 has the direction numbers listed there, which are the
// Pascal-triangle-mod-2 matrix: v[b] = v[b-1] ^ (v[b-1] >> 1).
static uint32_t sobolDim(uint32_t index, int dim)
{
    uint32_t result = 0;
    uint32_t v = 0x80000000u;
    for (uint32_t bit = 0; bit < 32; ++bit) {
        if ((index >> bit) & 1u) result ^= v;
        v = (dim == 0) ? (v >> 1) : (v ^ (v >> 1));
    }
    return result;
}

// Random.h:192-204
static float haltonValue(uint32_t index, int base)
{
    float result = 0.0f;
    float f = 1.0f;
    float denom = (float)base;
    uint32_t n = index;
    while (n > 0) {
        f = f / denom;
        result += f * (float)(n % (uint32_t)base);
        n = n / (uint32_t)base;
    }
    return result;
}

// Random.h:172-189 — reproduced verbatim including the non-coprime bases
static const int kHaltonBases[16][2] = {{2, 3},  {2, 5},  {2, 7},  {3, 7}, {4, 5},   {5, 7},  {5, 9},  {5, 11},
                                        {6, 11}, {5, 11}, {8, 11}, {3, 5}, {11, 15}, {2, 15}, {3, 19}, {7, 10}};

// Random.h:85-108 owenScrambleSequence with the three generators folded in
// (sobol :221-264, halton :169-217, hammersley :134-154).
void qmcSequence(int mode, vec2 *results, uint32_t count, uint32_t sequenceIndex)
{
    const uint32_t seed = burleyHash(sequenceIndex + 1);
    const uint32_t seed0 = burleyHashCombine(seed, 0);
    const uint32_t seed1 = burleyHashCombine(seed, 1);
    const float divisor = 1.0f / (float)count; // hammersley :145
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t index = nestedUniformScramble(i, seed);
        float sx, sy;
        switch (mode) {
        case HR_SAMPLE_SOBOL:
            sx = toNormalizedFloat(sobolDim(index, 0));
            sy = toNormalizedFloat(sobolDim(index, 1));
            break;
        case HR_SAMPLE_HALTON:
            sx = haltonValue(index, kHaltonBases[sequenceIndex & 15][0]);
            sy = haltonValue(index, kHaltonBases[sequenceIndex & 15][1]);
            break;
        default: // HR_SAMPLE_HAMMERSLEY
            sx = (float)i * divisor;
            sy = (float)reverseBits(index) * 2.3283064365386963e-10f;
            break;
        }
        results[i].x = toNormalizedFloat(nestedUniformScramble(toUint32(sx), seed0));
        results[i].y = toNormalizedFloat(nestedUniformScramble(toUint32(sy), seed1));
    }
}

// Random.h:268-289.  Uses the C library's sqrtf/cosf/sinf like the reference does, so
// the result is libm-specific in the last bit (SURVEY §4 caveat).
void radialSobol(vec2 *results, uint32_t count, uint32_t sequenceIndex)
{
    qmcSequence(HR_SAMPLE_SOBOL, results, count, sequenceIndex);
    const float two_pi = 6.28318530717958647692f;
    for (uint32_t i = 0; i < count; ++i) {
        float s = results[i].x;
        float t = results[i].y;
        float sqrt_t = sqrtf(t);
        float two_pi_s = two_pi * s;
        float x = sqrt_t * cosf(two_pi_s);
        float y = sqrt_t * sinf(two_pi_s);
        results[i].x = (x + 1.0f) * 0.5f;
        results[i].y = (y + 1.0f) * 0.5f;
    }
}

// Hash.h:17-30 — FNV-1a over the object bytes; `char` is signed on the reference's
// targets and here, so bytes >= 0x80 sign-extend into the 64-bit xor.
static uint64_t fnv1a(const void *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    const signed char *b = (const signed char *)p;
    for (size_t i = 0; i < n; ++i) {
        h ^= (uint64_t)(int64_t)b[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
// BlueNoise.h:97-100
static float blueRandom(uint32_t seed)
{
    uint64_t a = fnv1a(&seed, sizeof(seed));
    uint64_t b = fnv1a(&a, sizeof(a));
    return (float)b / (float)UINT64_MAX;
}
// BlueNoise.h:52-88 + Random.h:158-165 — best-candidate (30 candidates) blue noise.
void blueNoise(vec2 *results, uint32_t count, int sequenceIndex)
{
    int seed = (int)fnv1a(&sequenceIndex, sizeof(int)); // BlueNoise.h:57
    std::vector<vec2> pts;
    auto dist = [](vec2 a, vec2 b) {
        float dx = b.x - a.x, dy = b.y - a.y;
        return sqrtf(dx * dx + dy * dy);
    };
    const float diag = dist(vec2{0, 0}, vec2{1, 1});
    {
        float x = blueRandom((uint32_t)seed++);
        float y = blueRandom((uint32_t)seed++);
        pts.push_back(vec2{x, y});
    }
    for (int i = 0; i < (int)count - 1; ++i) {
        float furthest = 0.0f;
        vec2 best{0, 0};
        for (int c = 0; c < 30; ++c) {
            float x = blueRandom((uint32_t)seed++);
            float y = blueRandom((uint32_t)seed++);
            vec2 cand{x, y};
            float nearest = diag; // NearestPointFinder::FindNearestPoint, BlueNoise.h:32-46
            for (const vec2 &p : pts) {
                float d = dist(cand, p);
                if (d < nearest) nearest = d;
            }
            if (nearest > furthest) {
                best = cand;
                furthest = nearest;
            }
        }
        pts.push_back(best);
    }
    for (uint32_t i = 0; i < count; ++i) results[i] = pts[i];
}

// Random.h:113-130 — std::mt19937 + std::uniform_real_distribution: standard-library
// specific by construction (SURVEY §4 caveat); same calls as the reference.
void uniformRandomFloats(vec2 *results, uint32_t count, uint32_t seed)
{
    std::mt19937 generator(seed);
    std::uniform_real_distribution<float> distribution(0.0f, 1.0f);
    for (uint32_t i = 0; i < count; ++i) {
        results[i].x = distribution(generator);
        results[i].y = distribution(generator);
    }
}

// Random.h:293-355 — std-library specific like the above.
void randomPolygonal(vec2 *results, uint32_t numEdges, uint32_t count, uint32_t seed)
{
    std::vector<vec2> vertices(numEdges + 1);
    const float two_pi = 6.28318530717958647692f;
    float stepSize = two_pi / (float)numEdges;
    for (uint32_t i = 0; i < numEdges; ++i) {
        float theta = stepSize * (float)i;
        vertices[i] = vec2{cosf(theta), sinf(theta)};
    }
    vertices[numEdges] = vec2{0.0f, 0.0f};
    std::mt19937 generator(seed);
    std::uniform_real_distribution<float> floatDistribution(0.0f, 1.0f);
    std::uniform_int_distribution<int> intDistribution(0, (int)numEdges - 1);
    for (uint32_t i = 0; i < count; ++i) {
        int tri = intDistribution(generator);
        float alpha, beta;
        do {
            alpha = floatDistribution(generator);
            beta = floatDistribution(generator);
        } while (alpha + beta > 1.0f);
        float gamma = 1.0f - (alpha + beta);
        vec2 v0 = vertices[numEdges], v1 = vertices[tri], v2 = vertices[(tri + 1) % numEdges];
        float vx = v0.x * alpha + v1.x * beta + v2.x * gamma;
        float vy = v0.y * alpha + v1.y * beta + v2.y * gamma;
        results[i].x = (vx + 1.0f) * 0.5f;
        results[i].y = (vy + 1.0f) * 0.5f;
    }
}

// ---- multiscatter LUT (MultiScatterUtil.cpp:20-139) --------------------------------
// Host-side generator in the reference: libm's cosf / sinf and glm::normalize (v * inversesqrt(dot)).  The last bit of libm's
// cosf / sinf is library-specific, so — like every other transcendental of the arithmetic contract (oracle_math.h) — they are
// the Cephes single-precision algorithms here, on this side and in the product's k_multiscatter_lut alike: the two tables are
// then the same bits (tests/test_gpu_parity.py), and both stay within 2e-6 of the TIFF the reference ships (the reference's
// own libm differs from any other by as much).
static inline float sq(float f) { return f * f; }
static float lutG1(float NdotI, float alpha) // :22-27
{
    const float alpha2 = sq(alpha);
    const float denom = sqrtf(alpha2 + (1.0f - alpha2) * sq(NdotI)) + NdotI;
    return (2.0f * NdotI) / std::max(denom, 1e-5f);
}
static float lutValue(float NdotV, float alpha, const std::vector<vec2> &seq) // :49-80
{
    const float two_pi = 6.28318530717958647692f;
    vec3 V(sqrtf(1.0f - (NdotV * NdotV)), 0.0f, NdotV);
    float result = 0.0f;
    for (size_t i = 0; i < seq.size(); ++i) {
        // importanceSampleGGX :34-47
        float a2 = alpha * alpha;
        const float cosTheta = sqrtf(std::max(0.0f, (1.0f - seq[i].x) / ((a2 - 1.0f) * seq[i].x + 1.0f)));
        const float sinTheta = sqrtf(std::max(0.0f, 1.0f - sq(cosTheta)));
        const float phi = two_pi * seq[i].y;
        float sn, cs;
        sincos_(phi, &sn, &cs);
        vec3 H(sinTheta * cs, sinTheta * sn, cosTheta);
        H = H * (1.0f / sqrtf(dot(H, H))); // glm::normalize
        const vec3 L = 2.0f * dot(V, H) * H - V;
        float NdotL = clamp_(L.z, 0.0f, 1.0f);
        if (NdotL > 0.0f) {
            const float VdotH = clamp_(dot(V, H), 0.0f, 1.0f);
            const float NdotH = clamp_(H.z, 0.0f, 1.0f);
            const float G = (lutG1(NdotL, alpha) * lutG1(NdotV, alpha) * VdotH) / (NdotV * NdotH);
            result += G;
        }
    }
    return result / (float)seq.size();
}
void multiscatterLUT(float *out, int dim, int samples) // :91-124, row = roughness, col = NdotV
{
    std::vector<vec2> seq(samples);
    qmcSequence(HR_SAMPLE_SOBOL, seq.data(), (uint32_t)samples, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int row = 0; row < dim; ++row) {
        const float roughness = clamp_(((float)row + 0.5f) / (float)dim, 0.0f, 1.0f);
        const float alpha = roughness * roughness;
        for (int col = 0; col < dim; ++col) {
            const float NdotV = clamp_(((float)col + 0.5f) / (float)dim, 0.0f, 1.0f);
            float value = lutValue(NdotV, alpha, seq);
            out[row * dim + col] = (1.0f - value) / value;
        }
    }
}

} // namespace ora
