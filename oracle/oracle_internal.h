// oracle_internal.h — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
#pragma once

#include "hrcore.h"
#include "oracle_math.h"

#include <cstdint>
#include <string>
#include <vector>

namespace ora {

// ---- QMC (oracle_qmc.cpp) ----
uint32_t burleyHash(uint32_t x);
uint32_t laineKarrasPermutation(uint32_t x, uint32_t seed);
uint32_t reverseBits(uint32_t v);
uint32_t nestedUniformScramble(uint32_t x, uint32_t seed);
void qmcSequence(int mode, vec2 *results, uint32_t count, uint32_t sequenceIndex);
void radialSobol(vec2 *results, uint32_t count, uint32_t sequenceIndex);
void blueNoise(vec2 *results, uint32_t count, int sequenceIndex);
void uniformRandomFloats(vec2 *results, uint32_t count, uint32_t seed);
void randomPolygonal(vec2 *results, uint32_t numEdges, uint32_t count, uint32_t seed);
void multiscatterLUT(float *out, int dim, int samples);

// ---- scene (oracle_scene.cpp) ----
struct Texture {
    int w = 0, h = 0, c = 0;
    int wrapS = HR_WRAP_REPEAT, wrapT = HR_WRAP_REPEAT, filter = HR_FILTER_LINEAR;
    bool alive = false;
    std::vector<float> px; // w*h*c, u8 data converted as float(byte)/255.0f
    // HR_TEXTURE_LOD_CONE: levels 1 .. nLevels-1 one after the other (2x2 box filter of the level below), nLevels == 0: not built yet
    int nLevels = 0;
    std::vector<float> mips;
    float lodScale = 0.0f; // 0.5 * log2(w * h)
};

struct vec4 {
    float x, y, z, w;
};
vec4 sampleTexture(const Texture &t, float u, float v);
vec4 sampleTextureLod(const Texture &t, float u, float v, float lambda); // trilinear over the mip chain (HR_TEXTURE_LOD_CONE)

struct Geom {
    bool alive = false;
    int nVerts = 0;
    std::vector<float> pos, nrm, uv, tan, bit, col; // tightly packed copies
    std::vector<uint32_t> idx;
    int mode = HR_TRIANGLES;
    float world[16];
    int frontFaceCW = 0, isOccluder = 1, material = 0;
};

// Per-triangle records produced by commit (world space).
struct Tri {
    vec3 v0, e1, e2;
};
enum : uint32_t { TF_FRONT_CW = 1u, TF_NON_OCCLUDER = 2u, TF_HAS_UV = 4u, TF_HAS_TANGENTS = 8u, TF_HAS_COLORS = 16u };
struct TriAttr {
    vec3 n[3];
    vec2 uv[3];
    vec3 tan[3], bit[3], col[3];
    int material;
    uint32_t flags;
};

// Importance table of the environment map for HR_ESTIMATOR_ENV_MIS (include/hrcore.h): a piecewise-constant distribution over the
// texels, weight = (luminosity + maxLuminosity / 65536) x cos(elevation of the row), quantised to integers so that sums — hence the
// tables — do not depend on the order of summation (the HIP build is a parallel scan and must produce the same bits).
struct EnvTable {
    int w = 0, h = 0;
    int tex = -2;                 // texture id the table was built from (-2: none)
    std::vector<float> rowCdf;    // h + 1: P(row < j)
    std::vector<float> colCdf;    // h x (w + 1): P(col < i | row j)
    std::vector<float> prob;      // h x w: probability of texel (i, j)
    float meanLum = 0.0f;         // solid-angle-weighted mean of the (dilated) luminosity: the light pick's estimate of the map's power
};
vec4 texelOf(const Texture &t, int x, int y);

// ---- BVH (oracle_bvh.cpp) ----
struct BvhNode { // binary node holding both child boxes
    float lo[2][3], hi[2][3];
    int child[2]; // >= 0 internal node index; < 0 leaf: ~(first | (count-1) << 28)
};
struct Bvh {
    std::vector<BvhNode> nodes; // node 0 = root (empty when nTris <= leaf size)
    std::vector<uint32_t> order; // sorted position -> triangle id
    std::vector<Tri> tris;       // in sorted order
    int rootLeafCount = 0;       // >0 when the whole scene is a single leaf
    float lo[3], hi[3];
};
struct Hit {
    int prim = -1;
    float t = 0, u = 0, v = 0;
};
struct TraceCounters {
    uint64_t nodeVisits = 0, triTests = 0;
};

struct Context;
void buildLBVH(const std::vector<Tri> &tris, Bvh &bvh);
// closest hit with (t, prim) lexicographic minimum over t in (tmin, tmax); skip = source triangle or -1
Hit traceClosest(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, TraceCounters *tc, bool brute);
// occlusion query honouring non-occluder (alpha masked) triangles
bool traceOccluded(const Context &ctx, vec3 o, vec3 d, float tmin, float tmax, int skip, TraceCounters *tc, bool brute);

struct Context {
    int W = 0, H = 0;
    int rank = 0, world = 1, tile = 32;
    std::vector<float> fb; // RGBA32F, row 0 = bottom
    std::vector<Geom> geoms;
    std::vector<Texture> textures;
    std::vector<hr_material> materials;
    hr_lights lights{};
    int nSeq = 0, seqLen = 0;
    std::vector<vec2> seq, aperture;
    std::vector<vec2> seqOffsets;
    int blockNx = 0, blockNy = 0, blockCoords[32] = {0}; // interactive-mode block table (0: the unshuffled list)
    EnvTable env;
    std::vector<float> texDensity; // HR_TEXTURE_LOD_CONE: 0.5 * log2(uv area / world area) per triangle (prim id)
    // committed scene
    bool committed = false;
    std::vector<Tri> tris;       // submission order (prim id)
    std::vector<TriAttr> attrs;  // submission order
    Bvh bvh;
    float rayEps = 0.0f;
    float hitPad = 0.0f; // half the leaf padding (oracle_bvh.cpp: hitInTriBox)
    float aabbLo[3], aabbHi[3];
    bool brute = false; // brute-force intersection instead of the BVH (validation of the traversal)
    hr_pass_stats stats{};
    std::string err;
};

void commitScene(Context &ctx);
void buildEnvTable(Context &ctx); // (re)builds ctx.env when the environment texture changed
void buildTextureLod(Context &ctx); // HR_TEXTURE_LOD_CONE: missing mip chains + the per-triangle level offsets
bool alphaPasses(const Context &ctx, int prim, float u, float v); // alpha-mask test for occlusion rays

// ---- shading (oracle_shade.cpp) ----
void renderPass(Context &ctx, const hr_pass_params &pp, int nThreads);

// oracle_display.cpp
void displayResolve(const Context &ctx, const hr_display_params &P, int format, void *out);

} // namespace ora
