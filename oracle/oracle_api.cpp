// oracle_api.cpp — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product.
//
// C entry points of liboracle.so.  They mirror include/hrcore.h one-to-one with the
// prefix ora_ so that the parity tests drive the oracle and libhrcore with the same
// calls and the same POD inputs.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library.
#include "oracle_internal.h"

#include <chrono>
#include <cstring>
#include <omp.h>

using namespace ora;

struct hr_ctx {
    Context c;
    int nThreads = 0;
    std::vector<float> display; // ora_display_readback's buffer (16 bytes per pixel)
    uint32_t passesSinceClear = 0;
};

#define ORA_FAIL(ctx, code, msg) \
    do {                         \
        (ctx)->c.err = (msg);    \
        return (code);           \
    } while (0)

// ---- tile-shard exchange helpers: same pixel enumeration as libhrcore (tile by tile, 8x8 blocks inside a tile)
static uint64_t packedSlots(const Context &c, int rank, int world)
{
    const int tile = c.tile > 0 ? c.tile : 32;
    const int tilesX = (c.W + tile - 1) / tile, tilesY = (c.H + tile - 1) / tile, nTiles = tilesX * tilesY;
    const int owned = nTiles > rank ? (nTiles - rank + world - 1) / world : 0;
    return (uint64_t)owned * (uint64_t)(tile * tile);
}
template <class F> static void forEachPackedSlot(const Context &c, int rank, int world, F f)
{
    const int tile = c.tile > 0 ? c.tile : 32, tilesX = (c.W + tile - 1) / tile, bpr = tile >> 3;
    const uint64_t n = packedSlots(c, rank, world);
    for (uint64_t gid = 0; gid < n; ++gid) {
        const uint32_t perTile = (uint32_t)(tile * tile), slot = (uint32_t)(gid / perTile), within = (uint32_t)(gid % perTile);
        const int tileId = rank + (int)slot * world, tx = tileId % tilesX, ty = tileId / tilesX;
        const int blk = (int)(within >> 6), l = (int)(within & 63u);
        const int x = tx * tile + (blk % bpr) * 8 + (l & 7), y = ty * tile + (blk / bpr) * 8 + (l >> 3);
        f(gid, x < c.W && y < c.H, (size_t)y * c.W + x);
    }
}

namespace ora {
Context &contextOf(hr_ctx *ctx) { return ctx->c; } // for the probes in oracle_shade.cpp
}

extern "C" {

uint32_t ora_abi_version(void) { return HR_ABI_VERSION; }
int ora_ctx_create(const hr_ctx_desc *desc, hr_ctx **out)
{
    if (!out) return HR_ERR_INVALID;
    hr_ctx *ctx = new hr_ctx();
    if (desc) {
        ctx->c.rank = desc->rank;
        ctx->c.world = desc->world > 0 ? desc->world : 1;
        ctx->c.tile = desc->tile_size > 0 ? desc->tile_size : 32;
    }
    *out = ctx;
    return HR_OK;
}
int ora_ctx_destroy(hr_ctx *ctx)
{
    delete ctx;
    return HR_OK;
}
const char *ora_last_error(const hr_ctx *ctx) { return ctx ? ctx->c.err.c_str() : "null ctx"; }
// oracle-only knobs
int ora_set_threads(hr_ctx *ctx, int n)
{
    ctx->nThreads = n;
    return HR_OK;
}
int ora_set_brute_force(hr_ctx *ctx, int on)
{
    ctx->c.brute = on != 0;
    return HR_OK;
}

int ora_frame_resize(hr_ctx *ctx, int32_t w, int32_t h)
{
    if (w <= 0 || h <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "bad frame size");
    ctx->c.W = w, ctx->c.H = h;
    ctx->c.fb.assign((size_t)w * h * 4, 0.0f);
    return HR_OK;
}

static void copyAttr(std::vector<float> &dst, const float *src, int stride, int comps, int n)
{
    dst.clear();
    if (!src) return;
    if (stride == 0) stride = comps * (int)sizeof(float);
    dst.resize((size_t)n * comps);
    for (int i = 0; i < n; ++i) std::memcpy(&dst[(size_t)i * comps], (const char *)src + (size_t)i * stride, comps * sizeof(float));
}

int ora_geom_add(hr_ctx *ctx, const hr_mesh_desc *d, hr_geom_id *out)
{
    if (!d || !d->positions || !d->normals || !d->indices || d->n_vertices <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "mesh needs positions, normals, indices");
    for (int i = 0; i < d->n_indices; ++i)
        if (d->indices[i] >= (uint32_t)d->n_vertices) ORA_FAIL(ctx, HR_ERR_INVALID, "index out of range");
    {
        const int sb = d->position_stride == 0 ? 12 : d->position_stride;
        for (int i = 0; i < d->n_vertices; ++i) {
            float p[3];
            std::memcpy(p, (const char *)d->positions + (size_t)i * (size_t)sb, 12);
            for (int k = 0; k < 3; ++k)
                if (!(fabsf(p[k]) <= 3.0e37f)) ORA_FAIL(ctx, HR_ERR_INVALID, "vertex positions must be finite");
        }
        for (int k = 0; k < 16; ++k)
            if (!(fabsf(d->world_from_entity[k]) <= 3.0e37f)) ORA_FAIL(ctx, HR_ERR_INVALID, "world_from_entity must be finite");
    }
    Geom g;
    g.alive = true;
    g.nVerts = d->n_vertices;
    copyAttr(g.pos, d->positions, d->position_stride, 3, d->n_vertices);
    copyAttr(g.nrm, d->normals, d->normal_stride, 3, d->n_vertices);
    copyAttr(g.uv, d->uvs, d->uv_stride, 2, d->n_vertices);
    copyAttr(g.tan, d->tangents, d->tangent_stride, 3, d->n_vertices);
    copyAttr(g.bit, d->bitangents, d->bitangent_stride, 3, d->n_vertices);
    copyAttr(g.col, d->colors, d->color_stride, 3, d->n_vertices);
    g.idx.assign(d->indices, d->indices + d->n_indices);
    g.mode = d->mode;
    std::memcpy(g.world, d->world_from_entity, sizeof(g.world));
    g.frontFaceCW = d->front_face_cw;
    g.isOccluder = d->is_occluder;
    g.material = d->material_id;
    ctx->c.geoms.push_back(std::move(g));
    ctx->c.committed = false;
    if (out) *out = (hr_geom_id)ctx->c.geoms.size() - 1;
    return HR_OK;
}
int ora_geom_remove(hr_ctx *ctx, hr_geom_id id)
{
    if (id < 0 || id >= (int)ctx->c.geoms.size() || !ctx->c.geoms[id].alive) ORA_FAIL(ctx, HR_ERR_INVALID, "bad geom id");
    ctx->c.geoms[id] = Geom();
    ctx->c.committed = false;
    return HR_OK;
}
int ora_geom_set_transform(hr_ctx *ctx, hr_geom_id id, const float m[16])
{
    if (id < 0 || id >= (int)ctx->c.geoms.size() || !ctx->c.geoms[id].alive || !m) ORA_FAIL(ctx, HR_ERR_INVALID, "bad geom id");
    for (int k = 0; k < 16; ++k)
        if (!(fabsf(m[k]) <= 3.0e37f)) ORA_FAIL(ctx, HR_ERR_INVALID, "world_from_entity must be finite");
    std::memcpy(ctx->c.geoms[id].world, m, 16 * sizeof(float));
    ctx->c.committed = false;
    return HR_OK;
}
int ora_scene_clear(hr_ctx *ctx)
{
    ctx->c.geoms.clear();
    ctx->c.committed = false;
    return HR_OK;
}
int ora_scene_commit(hr_ctx *ctx)
{
    auto t0 = std::chrono::steady_clock::now();
    commitScene(ctx->c);
    (void)t0;
    return HR_OK;
}
int ora_scene_cache(hr_ctx *, const char *) { return HR_OK; } // the oracle always builds

int ora_scene_get_info(hr_ctx *ctx, hr_scene_info *out)
{
    if (!ctx->c.committed) ORA_FAIL(ctx, HR_ERR_INVALID, "scene not committed");
    std::memset(out, 0, sizeof(*out));
    out->n_triangles = ctx->c.tris.size();
    out->n_nodes = ctx->c.bvh.nodes.size();
    for (int k = 0; k < 3; ++k) out->aabb_min[k] = ctx->c.aabbLo[k], out->aabb_max[k] = ctx->c.aabbHi[k];
    out->ray_epsilon = ctx->c.rayEps;
    return HR_OK;
}

int ora_texture_create(hr_ctx *ctx, const hr_texture_desc *d, const void *pixels, hr_tex_id *out)
{
    if (!d || !pixels || d->width <= 0 || d->height <= 0 || (d->channels != 1 && d->channels != 3 && d->channels != 4))
        ORA_FAIL(ctx, HR_ERR_INVALID, "bad texture descriptor");
    Texture t;
    t.w = d->width, t.h = d->height, t.c = d->channels;
    t.wrapS = d->wrap_s, t.wrapT = d->wrap_t, t.filter = d->filter;
    t.alive = true;
    const size_t n = (size_t)t.w * t.h * t.c;
    t.px.resize(n);
    if (d->dtype == HR_TEX_U8) {
        const uint8_t *p = (const uint8_t *)pixels;
        for (size_t i = 0; i < n; ++i) t.px[i] = (float)p[i] / 255.0f;
    } else {
        std::memcpy(t.px.data(), pixels, n * sizeof(float));
    }
    ctx->c.textures.push_back(std::move(t));
    if (out) *out = (hr_tex_id)ctx->c.textures.size() - 1;
    return HR_OK;
}
int ora_texture_destroy(hr_ctx *ctx, hr_tex_id id)
{
    if (id < 0 || id >= (int)ctx->c.textures.size()) ORA_FAIL(ctx, HR_ERR_INVALID, "bad texture id");
    ctx->c.textures[id] = Texture();
    return HR_OK;
}

int ora_material_set(hr_ctx *ctx, int32_t id, const hr_material *m)
{
    if (id < 0 || id > (1 << 20) || !m) ORA_FAIL(ctx, HR_ERR_INVALID, "bad material id");
    if ((int)ctx->c.materials.size() <= id) {
        hr_material none{};
        none.type = -1;
        ctx->c.materials.resize(id + 1, none);
    }
    ctx->c.materials[id] = *m;
    return HR_OK;
}
int ora_lights_set(hr_ctx *ctx, const hr_lights *l)
{
    if (!l || l->n_directional > HR_MAX_DIRECTIONAL_LIGHTS || l->n_point > HR_MAX_POINT_LIGHTS || l->n_spot > HR_MAX_SPOT_LIGHTS)
        ORA_FAIL(ctx, HR_ERR_INVALID, "bad light block");
    ctx->c.lights = *l;
    return HR_OK;
}

int ora_interactive_blocks_set(hr_ctx *ctx, const int32_t *coords, int32_t nx, int32_t ny)
{
    if (!coords) {
        ctx->c.blockNx = ctx->c.blockNy = 0;
        return HR_OK;
    }
    if (nx <= 0 || ny <= 0 || nx * ny > 16) ORA_FAIL(ctx, HR_ERR_INVALID, "block table: nx*ny must be 1..16");
    ctx->c.blockNx = nx, ctx->c.blockNy = ny;
    std::memcpy(ctx->c.blockCoords, coords, sizeof(int32_t) * 2 * (size_t)(nx * ny));
    return HR_OK;
}

int ora_sequences_set(hr_ctx *ctx, const float *seq, const float *ap, int32_t nSeq, int32_t len)
{
    if (!seq || !ap || nSeq <= 0 || len <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "bad sequence table");
    ctx->c.nSeq = nSeq, ctx->c.seqLen = len;
    ctx->c.seq.assign((const vec2 *)seq, (const vec2 *)seq + (size_t)nSeq * len);
    ctx->c.aperture.assign((const vec2 *)ap, (const vec2 *)ap + (size_t)nSeq * len);
    return HR_OK;
}
int ora_seq_offsets_set(hr_ctx *ctx, const float *off, int32_t n)
{
    if (!off || n <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "bad offsets table");
    ctx->c.seqOffsets.assign((const vec2 *)off, (const vec2 *)off + n);
    return HR_OK;
}
int ora_qmc_generate(hr_ctx *ctx, int32_t mode, uint32_t seqIndex, uint32_t count, int32_t radial, float *out)
{
    if (!out) ORA_FAIL(ctx, HR_ERR_INVALID, "null output");
    vec2 *r = (vec2 *)out;
    switch (mode) {
    case HR_SAMPLE_SOBOL:
        if (radial)
            radialSobol(r, count, seqIndex);
        else
            qmcSequence(mode, r, count, seqIndex);
        return HR_OK;
    case HR_SAMPLE_HALTON:
    case HR_SAMPLE_HAMMERSLEY:
        qmcSequence(mode, r, count, seqIndex);
        return HR_OK;
    case HR_SAMPLE_BLUE_NOISE:
        blueNoise(r, count, (int)seqIndex);
        return HR_OK;
    case HR_SAMPLE_RANDOM:
        uniformRandomFloats(r, count, seqIndex);
        return HR_OK;
    }
    ORA_FAIL(ctx, HR_ERR_INVALID, "bad sample mode");
}
// oracle-only: randomPolygonal (Random.h:293-355)
int ora_qmc_polygon(hr_ctx *, uint32_t edges, uint32_t seqIndex, uint32_t count, float *out)
{
    randomPolygonal((vec2 *)out, edges, count, seqIndex);
    return HR_OK;
}
// one aperture table (PassGenerator.cpp:653-676): the checker's side of hr_aperture_generate
int ora_aperture_generate(hr_ctx *ctx, int32_t bokeh, uint32_t seqIndex, uint32_t count, float *out)
{
    if (!out || count == 0) ORA_FAIL(ctx, HR_ERR_INVALID, "bad count / output");
    switch (bokeh) {
    case HR_BOKEH_CIRCULAR: radialSobol((vec2 *)out, count, seqIndex); return HR_OK;
    case HR_BOKEH_PENTAGON: randomPolygonal((vec2 *)out, 5, count, seqIndex); return HR_OK;
    case HR_BOKEH_HEXAGON: randomPolygonal((vec2 *)out, 6, count, seqIndex); return HR_OK;
    case HR_BOKEH_OCTAGON: randomPolygonal((vec2 *)out, 8, count, seqIndex); return HR_OK;
    }
    ORA_FAIL(ctx, HR_ERR_INVALID, "unknown bokeh shape");
}
// generateRandomSequences (PassGenerator.cpp:603-684)
int ora_sequences_generate(hr_ctx *ctx, int32_t sampleMode, int32_t bokeh, int32_t len)
{
    const int nSeq = HR_NUM_RANDOM_SEQUENCES;
    std::vector<vec2> seq((size_t)nSeq * len), ap((size_t)nSeq * len);
    for (int s = 0; s < nSeq; ++s) {
        ora_qmc_generate(ctx, sampleMode, (uint32_t)s, (uint32_t)len, 0, (float *)&seq[(size_t)s * len]);
        switch (bokeh) {
        case HR_BOKEH_CIRCULAR: radialSobol(&ap[(size_t)s * len], len, s); break;
        case HR_BOKEH_PENTAGON: randomPolygonal(&ap[(size_t)s * len], 5, len, s); break;
        case HR_BOKEH_HEXAGON: randomPolygonal(&ap[(size_t)s * len], 6, len, s); break;
        default: randomPolygonal(&ap[(size_t)s * len], 8, len, s); break;
        }
    }
    return ora_sequences_set(ctx, (const float *)seq.data(), (const float *)ap.data(), nSeq, len);
}
// generateSequenceOffsets (PassGenerator.cpp:150-159)
int ora_seq_offsets_generate(hr_ctx *ctx)
{
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    std::vector<vec2> off((size_t)ctx->c.W * ctx->c.H);
    qmcSequence(HR_SAMPLE_SOBOL, off.data(), (uint32_t)off.size(), 0);
    ctx->c.seqOffsets = std::move(off);
    return HR_OK;
}
int ora_multiscatter_lut_generate(hr_ctx *ctx, float *out, hr_tex_id *outTex)
{
    std::vector<float> lut(128 * 128);
    multiscatterLUT(lut.data(), 128, 4096);
    if (out) std::memcpy(out, lut.data(), lut.size() * sizeof(float));
    if (outTex) {
        hr_texture_desc d{128, 128, 1, HR_TEX_F32, HR_WRAP_CLAMP_TO_EDGE, HR_WRAP_CLAMP_TO_EDGE, HR_FILTER_LINEAR};
        return ora_texture_create(ctx, &d, lut.data(), outTex);
    }
    return HR_OK;
}

int ora_clear(hr_ctx *ctx)
{
    std::fill(ctx->c.fb.begin(), ctx->c.fb.end(), 0.0f);
    ctx->c.stats = hr_pass_stats{};
    ctx->passesSinceClear = 0;
    return HR_OK;
}
int ora_render_pass(hr_ctx *ctx, const hr_pass_params *pp)
{
    Context &c = ctx->c;
    if (!pp) ORA_FAIL(ctx, HR_ERR_INVALID, "null params");
    if (c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    if (!c.committed) ORA_FAIL(ctx, HR_ERR_INVALID, "scene not committed");
    if (c.nSeq <= 0 || c.seqOffsets.empty()) ORA_FAIL(ctx, HR_ERR_INVALID, "sample tables not set");
    auto t0 = std::chrono::steady_clock::now();
    renderPass(c, *pp, ctx->nThreads);
    ctx->passesSinceClear++;
    c.stats.ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HR_OK;
}
int ora_get_stats(hr_ctx *ctx, hr_pass_stats *out)
{
    *out = ctx->c.stats;
    return HR_OK;
}
int ora_readback(hr_ctx *ctx, const float **rgba, int32_t *w, int32_t *h)
{
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    *rgba = ctx->c.fb.data();
    if (w) *w = ctx->c.W;
    if (h) *h = ctx->c.H;
    return HR_OK;
}
int ora_frame_packed_slots(hr_ctx *ctx, int32_t rank, int32_t world, uint64_t *n)
{
    if (!n || world <= 0 || rank < 0 || rank >= world) ORA_FAIL(ctx, HR_ERR_INVALID, "bad rank / world");
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    *n = packedSlots(ctx->c, rank, world);
    return HR_OK;
}
int ora_frame_pack_owned(hr_ctx *ctx, void *out, void *)
{
    if (!out) ORA_FAIL(ctx, HR_ERR_INVALID, "null output");
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    float *o = (float *)out;
    const float *fb = ctx->c.fb.data();
    forEachPackedSlot(ctx->c, ctx->c.rank, ctx->c.world, [&](uint64_t gid, bool in, size_t pixel) {
        for (int k = 0; k < 4; ++k) o[4 * gid + k] = in ? fb[4 * pixel + k] : 0.0f;
    });
    return HR_OK;
}
int ora_frame_unpack(hr_ctx *ctx, int32_t rank, int32_t world, const void *packed, void *full, void *)
{
    if (!packed || !full) ORA_FAIL(ctx, HR_ERR_INVALID, "null argument");
    if (world <= 0 || rank < 0 || rank >= world) ORA_FAIL(ctx, HR_ERR_INVALID, "bad rank / world");
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    const float *p = (const float *)packed;
    float *f = (float *)full;
    forEachPackedSlot(ctx->c, rank, world, [&](uint64_t gid, bool in, size_t pixel) {
        if (in)
            for (int k = 0; k < 4; ++k) f[4 * pixel + k] = p[4 * gid + k];
    });
    return HR_OK;
}

int ora_display(hr_ctx *ctx, const hr_display_params *params, int32_t format, void *out, uint32_t *passes_shown)
{
    if (!params || !out) ORA_FAIL(ctx, HR_ERR_INVALID, "null argument");
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    format &= ~HR_DISPLAY_PROGRESSIVE; // the oracle has no pipeline
    if (format < HR_DISPLAY_RGBA8 || format > HR_DISPLAY_HDR_RGBA32F) ORA_FAIL(ctx, HR_ERR_INVALID, "unknown display format");
    displayResolve(ctx->c, *params, format, out);
    if (passes_shown) *passes_shown = ctx->passesSinceClear;
    return HR_OK;
}
int ora_display_readback(hr_ctx *ctx, const hr_display_params *params, int32_t format, const void **pixels, int32_t *w, int32_t *h, uint32_t *passes_shown)
{
    if (!pixels) ORA_FAIL(ctx, HR_ERR_INVALID, "null output");
    if (ctx->c.W <= 0) ORA_FAIL(ctx, HR_ERR_INVALID, "no frame");
    ctx->display.resize((size_t)ctx->c.W * ctx->c.H * 4);
    int rc = ora_display(ctx, params, format, ctx->display.data(), passes_shown);
    if (rc) return rc;
    *pixels = ctx->display.data();
    if (w) *w = ctx->c.W;
    if (h) *h = ctx->c.H;
    return HR_OK;
}
int ora_get_step_log(hr_ctx *ctx, hr_step_record *, int32_t, int32_t *n)
{
    if (!n) ORA_FAIL(ctx, HR_ERR_INVALID, "null output");
    *n = 0; // the oracle has no pipeline
    return HR_OK;
}
int ora_frame_passes_resolved(hr_ctx *ctx, uint64_t *passes)
{
    if (!passes) ORA_FAIL(ctx, HR_ERR_INVALID, "null output");
    *passes = ctx->passesSinceClear;
    return HR_OK;
}
int ora_readback_progressive(hr_ctx *ctx, const float **rgba, int32_t *w, int32_t *h, uint32_t *passes)
{
    int rc = ora_readback(ctx, rgba, w, h); // the oracle has no pipeline: every requested pass is in the buffer
    if (rc == HR_OK && passes) *passes = ctx->passesSinceClear;
    return rc;
}
int ora_frame_pass_batch(hr_ctx *, int32_t, int32_t *batch)
{
    if (batch) *batch = 1;
    return HR_OK;
}
int ora_synchronize(hr_ctx *) { return HR_OK; }
int ora_flush(hr_ctx *) { return HR_OK; }

int ora_debug_trace(hr_ctx *ctx, int32_t n, const float *o, const float *d, const float *tmax, const int32_t *skip, int32_t anyHit, hr_hit *out)
{
    Context &c = ctx->c;
    if (!c.committed) ORA_FAIL(ctx, HR_ERR_INVALID, "scene not committed");
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; ++i) {
        vec3 ro(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        float tm = tmax ? tmax[i] : INFINITY;
        int sk = skip ? skip[i] : -1;
        if (anyHit) {
            bool occ = traceOccluded(c, ro, rd, c.rayEps, tm, sk, nullptr, c.brute);
            out[i] = hr_hit{occ ? 0 : -1, 0, 0, 0};
        } else {
            Hit h = traceClosest(c, ro, rd, c.rayEps, tm, sk, nullptr, c.brute);
            out[i] = hr_hit{h.prim, h.prim >= 0 ? h.t : 0.0f, h.prim >= 0 ? h.u : 0.0f, h.prim >= 0 ? h.v : 0.0f};
        }
    }
    return HR_OK;
}

// oracle-only: expose the spec'd transcendental functions and integer hashes for unit tests
float ora_sin(float x) { return sin_(x); }
float ora_cos(float x) { return cos_(x); }
float ora_atan2(float y, float x) { return atan2_(y, x); }
float ora_exp(float x) { return exp_(x); }
uint32_t ora_burley_hash(uint32_t x) { return burleyHash(x); }
uint32_t ora_reverse_bits(uint32_t x) { return reverseBits(x); }
uint32_t ora_laine_karras(uint32_t x, uint32_t seed) { return laineKarrasPermutation(x, seed); }
uint32_t ora_nested_scramble(uint32_t x, uint32_t seed) { return nestedUniformScramble(x, seed); }

} // extern "C"
