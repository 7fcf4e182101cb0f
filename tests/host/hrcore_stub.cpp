// hrcore_stub.cpp — TEST INFRASTRUCTURE.  A do-nothing implementation of include/hrcore.h for the CPU-only sanitizer runs of
// the C++ drop-in layer (ThreadSanitizer / AddressSanitizer cannot run on the GPU boxes of this pool, and libhrcore needs a
// device).  It lets tests/host/host_threading_test.cpp drive PassGenerator's job queue, callbacks and object lifetimes without
// a GPU.  It renders nothing, is linked ONLY into that test executable, and is never part of libheatrayhost or the product.
#include "hrcore.h"

#include <cstdlib>
#include <cstring>
#include <vector>

struct hr_ctx {
    int w = 0, h = 0;
    std::vector<float> frame;
    unsigned passes = 0;
    int nTextures = 0, nGeoms = 0;
};

extern "C" {
uint32_t hr_abi_version(void) { return HR_ABI_VERSION; }
int hr_ctx_create(const hr_ctx_desc *, hr_ctx **out) { *out = new hr_ctx(); return HR_OK; }
int hr_ctx_destroy(hr_ctx *c) { delete c; return HR_OK; }
const char *hr_last_error(const hr_ctx *) { return "stub"; }
int hr_ctx_set_stream(hr_ctx *, void *) { return HR_OK; }
int hr_frame_resize(hr_ctx *c, int32_t w, int32_t h) { c->w = w, c->h = h; c->frame.assign((size_t)w * h * 4, 0.0f); c->passes = 0; return HR_OK; }
int hr_frame_bind_external(hr_ctx *, void *) { return HR_OK; }
int hr_frame_device_ptr(hr_ctx *, void **p) { *p = nullptr; return HR_OK; }
int hr_geom_add(hr_ctx *c, const hr_mesh_desc *, hr_geom_id *out) { if (out) *out = c->nGeoms; c->nGeoms++; return HR_OK; }
int hr_geom_remove(hr_ctx *, hr_geom_id) { return HR_OK; }
int hr_geom_set_transform(hr_ctx *, hr_geom_id, const float[16]) { return HR_OK; }
int hr_scene_clear(hr_ctx *) { return HR_OK; }
int hr_scene_commit(hr_ctx *) { return HR_OK; }
int hr_scene_cache(hr_ctx *, const char *) { return HR_OK; }
int hr_scene_get_info(hr_ctx *, hr_scene_info *o) { std::memset(o, 0, sizeof(*o)); return HR_OK; }
int hr_texture_create(hr_ctx *c, const hr_texture_desc *, const void *, hr_tex_id *out) { if (out) *out = c->nTextures; c->nTextures++; return HR_OK; }
int hr_texture_destroy(hr_ctx *, hr_tex_id) { return HR_OK; }
int hr_material_set(hr_ctx *, int32_t, const hr_material *) { return HR_OK; }
int hr_lights_set(hr_ctx *, const hr_lights *) { return HR_OK; }
int hr_sequences_set(hr_ctx *, const float *, const float *, int32_t, int32_t) { return HR_OK; }
int hr_seq_offsets_set(hr_ctx *, const float *, int32_t) { return HR_OK; }
int hr_qmc_generate(hr_ctx *, int32_t, uint32_t, uint32_t count, int32_t, float *out) { if (out) std::memset(out, 0, sizeof(float) * 2 * count); return HR_OK; }
int hr_aperture_generate(hr_ctx *, int32_t, uint32_t, uint32_t count, float *out) { if (out) std::memset(out, 0, sizeof(float) * 2 * count); return HR_OK; }
int hr_sequences_generate(hr_ctx *, int32_t, int32_t, int32_t) { return HR_OK; }
int hr_seq_offsets_generate(hr_ctx *) { return HR_OK; }
int hr_multiscatter_lut_generate(hr_ctx *c, float *out, hr_tex_id *tex) { if (out) std::memset(out, 0, sizeof(float) * 128 * 128); if (tex) *tex = c->nTextures++; return HR_OK; }
int hr_interactive_blocks_set(hr_ctx *, const int32_t *, int32_t, int32_t) { return HR_OK; }
int hr_clear(hr_ctx *c) { std::fill(c->frame.begin(), c->frame.end(), 0.0f); c->passes = 0; return HR_OK; }
int hr_render_pass(hr_ctx *c, const hr_pass_params *) { for (size_t i = 3; i < c->frame.size(); i += 4) c->frame[i] += 1.0f; c->passes++; return HR_OK; }
int hr_frame_pass_batch(hr_ctx *, int32_t, int32_t *b) { *b = 1; return HR_OK; }
int hr_flush(hr_ctx *) { return HR_OK; }
int hr_get_stats(hr_ctx *, hr_pass_stats *o) { std::memset(o, 0, sizeof(*o)); return HR_OK; }
int hr_get_kernel_times(hr_ctx *, hr_kernel_times *o) { std::memset(o, 0, sizeof(*o)); return HR_OK; }
int hr_readback(hr_ctx *c, const float **rgba, int32_t *w, int32_t *h) { *rgba = c->frame.data(); if (w) *w = c->w; if (h) *h = c->h; return HR_OK; }
int hr_readback_progressive(hr_ctx *c, const float **rgba, int32_t *w, int32_t *h, uint32_t *p) { if (p) *p = c->passes; return hr_readback(c, rgba, w, h); }
int hr_synchronize(hr_ctx *) { return HR_OK; }
int hr_display(hr_ctx *, const hr_display_params *, int32_t, void *, uint32_t *) { return HR_OK; }
int hr_frame_passes_resolved(hr_ctx *, uint64_t *n) { if (n) *n = 0; return HR_OK; }
int hr_get_step_log(hr_ctx *, hr_step_record *, int32_t, int32_t *n) { if (n) *n = 0; return HR_OK; }
int hr_display_readback(hr_ctx *c, const hr_display_params *, int32_t, const void **px, int32_t *w, int32_t *h, uint32_t *) { *px = c->frame.data(); if (w) *w = c->w; if (h) *h = c->h; return HR_OK; }
int hr_frame_packed_slots(hr_ctx *, int32_t, int32_t, uint64_t *n) { *n = 0; return HR_OK; }
int hr_frame_pack_owned(hr_ctx *, void *, void *) { return HR_OK; }
int hr_frame_unpack(hr_ctx *, int32_t, int32_t, const void *, void *, void *) { return HR_OK; }
int hr_debug_trace(hr_ctx *, int32_t, const float *, const float *, const float *, const int32_t *, int32_t, hr_hit *) { return HR_OK; }
}
