/*
 * hrcore.h — C-ABI of libhrcore, the MI355X-native per-pass ray kernel.
 *
 * This is the drop-in boundary.  The reference talks to its
 * ray engine through the OpenRL C API (3rdParty/OpenRL/rl.h:383-528) wrapped
 * by the headers in Source/RLWrapper/; this header declares the entry points a
 * re-implemented PassGenerator / Scene / Mesh / Material / Light layer binds
 * instead.  Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", opaque handle, plain pointers and sizes, int status
 *     (HR_OK == 0).  No exceptions cross the boundary.  After a non-zero
 *     status hr_last_error() returns a human-readable message
 *     (reference: RLFunc/checkError, Source/RLWrapper/Error.h:18-41).
 *   - The caller owns every input array; the library copies at call time
 *     (reference semantics: rlBufferData copies, Source/RLWrapper/Buffer.h:46-55).
 *   - All calls on one ctx must come from one thread at a time (the
 *     reference's "OpenRL thread", Source/HeatrayRenderer/PassGenerator.h:229-231).
 *   - Matrices are column-major float[16] (glm::mat4 memory layout).
 *   - The accumulation buffer is RGBA32F, row-major, row 0 = BOTTOM scanline
 *     (OpenRL frame origin, Resources/shaders/perspective.rlsl:73), A = sample
 *     count (Resources/shaders/perspective.rlsl:60).
 *
 * The same POD structs are the input format of the CPU oracle (oracle/), whose
 * entry points mirror these with the prefix ora_ instead of hr_.
 */
#ifndef HRCORE_H
#define HRCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HR_OK 0
#define HR_ERR_INVALID 1   /* bad argument / bad state            */
#define HR_ERR_DEVICE 2    /* HIP error (no device, OOM, fault)   */
#define HR_ERR_UNSUPPORTED 3

/* Source/HeatrayRenderer/Lights/ShaderLightingDefines.h:18-20 */
#define HR_MAX_DIRECTIONAL_LIGHTS 5
#define HR_MAX_POINT_LIGHTS 5
#define HR_MAX_SPOT_LIGHTS 5
/* Source/HeatrayRenderer/PassGenerator.h:193 */
#define HR_NUM_RANDOM_SEQUENCES 16

typedef struct hr_ctx hr_ctx;

/* Version of this interface: bumped whenever a struct grows, a signature changes or an entry point is added (6: hr_aperture_generate and
 * device generators behind every sample mode and bokeh shape; 5: hr_ctx_desc.memory_budget, hr_step_record.group,
 * hr_abi_version itself; 4 was round 4's hr_display(..., passes_shown) and the grown hr_scene_info / hr_kernel_times).  A caller compares
 * hr_abi_version() of the library it loaded with the HR_ABI_VERSION it was compiled against BEFORE anything else and refuses a
 * mismatch — a library that writes a longer struct than the caller allocated corrupts memory silently otherwise.  The Python
 * binding (heatray_amd/_ffi.py) and the C++ layer (heatray_amd/host/HeatrayRenderer/PassGenerator.cpp) do. */
#define HR_ABI_VERSION 6u
uint32_t hr_abi_version(void);

/* ------------------------------------------------------------------ context */

typedef struct hr_ctx_desc {
    int32_t device_id;  /* HIP device ordinal                                              */
    int32_t rank;       /* pixel-tile shard owned by this ctx: tiles t with t % world==rank */
    int32_t world;      /* number of shards (1 = whole frame)                              */
    int32_t tile_size;  /* tile edge in pixels (0 -> 32), SURVEY §8e                        */
    void *stream;       /* hipStream_t to launch on, or NULL for the null stream           */
    uint32_t flags;     /* HR_CTX_*                                                        */
    /* Device memory the pass pipeline may hold for rays and pass buffers, in bytes (0: no limit — 31 GB for a 1080p render of 1 M
     * triangles at the default 16 passes per step).  Fewer passes are then injected per pipeline step: first as many as fit when
     * every ray queue is as long as it can possibly get, later as many as the lengths the render really shows allow.  Scene,
     * textures and the frame itself are not counted.  A budget too small for ONE pass per step fails hr_render_pass (HR_ERR_INVALID). */
    uint64_t memory_budget;
} hr_ctx_desc;

#define HR_CTX_COLLECT_STATS 1u /* count node visits / triangle tests per pass (slower) */
#define HR_CTX_TIME_KERNELS 2u  /* bracket every kernel launch with HIP events (hr_get_kernel_times) */

/* replaces OpenRLCreateContext / OpenRLSetCurrentContext (PassGenerator.cpp:164-165) */
int hr_ctx_create(const hr_ctx_desc *desc, hr_ctx **out);
/* replaces OpenRLDestroyContext (PassGenerator.cpp:432) */
int hr_ctx_destroy(hr_ctx *ctx);
const char *hr_last_error(const hr_ctx *ctx);
/* change the stream later launches go to (e.g. torch's current stream) */
int hr_ctx_set_stream(hr_ctx *ctx, void *stream);

/* replaces the FBO texture + PixelPackBuffer (re)allocation and rlViewport
 * (PassGenerator.cpp:174-196, 301-323). Clears the accumulation buffer. */
int hr_frame_resize(hr_ctx *ctx, int32_t width, int32_t height);
/* Use caller-owned device memory (width*height*4 floats) as the accumulation
 * buffer, e.g. a torch tensor handed to RCCL.  NULL returns to internal memory. */
int hr_frame_bind_external(hr_ctx *ctx, void *device_rgba);
int hr_frame_device_ptr(hr_ctx *ctx, void **device_rgba);

/* --------------------------------------------------------------- geometry */

#define HR_TRIANGLES 0      /* RL_TRIANGLES      (Mesh.cpp:134-136) */
#define HR_TRIANGLE_STRIP 1 /* RL_TRIANGLE_STRIP (Mesh.cpp:137-139) */

/* One submesh == one RL primitive (Mesh.cpp:55-153).  Attribute pointers are
 * planar float arrays addressed with a byte stride like rlVertexAttribBuffer
 * (Mesh.cpp:104-132); NULL = attribute absent. */
typedef struct hr_mesh_desc {
    const float *positions;  /* 3 floats / vertex, required */
    const float *normals;    /* 3 floats / vertex, required */
    const float *uvs;        /* 2 floats / vertex           */
    const float *tangents;   /* 3 floats / vertex           */
    const float *bitangents; /* 3 floats / vertex           */
    const float *colors;     /* 3 floats / vertex           */
    int32_t position_stride; /* bytes; 0 -> tightly packed  */
    int32_t normal_stride;
    int32_t uv_stride;
    int32_t tangent_stride;
    int32_t bitangent_stride;
    int32_t color_stride;
    int32_t n_vertices;
    const uint32_t *indices; /* RL_UNSIGNED_INT (Mesh.cpp:152) */
    int32_t n_indices;       /* elementCount                   */
    int32_t mode;            /* HR_TRIANGLES / HR_TRIANGLE_STRIP */
    float world_from_entity[16]; /* localTransform * transform (Mesh.cpp:85,102) */
    int32_t front_face_cw;   /* rlFrontFace(RL_CW) when det < 0 (Mesh.cpp:86-91) */
    int32_t is_occluder;     /* RL_PRIMITIVE_IS_OCCLUDER (Mesh.cpp:95-100)       */
    int32_t material_id;     /* row set with hr_material_set                     */
} hr_mesh_desc;

typedef int32_t hr_geom_id;

/* replaces Buffer::create + rlVertexAttribBuffer + rlDrawElements (Mesh.cpp:29-152).  Positions and the transform must be finite
 * (HR_ERR_INVALID otherwise: a NaN box has no order, and the tree builders and slab tests assume one). */
int hr_geom_add(hr_ctx *ctx, const hr_mesh_desc *desc, hr_geom_id *out);
/* replaces Scene::removeMesh / primitive destruction (Scene.h:52) */
int hr_geom_remove(hr_ctx *ctx, hr_geom_id id);
/* replaces setMatrix4fv("worldFromEntity") (Scene.cpp:38-49) */
int hr_geom_set_transform(hr_ctx *ctx, hr_geom_id id, const float world_from_entity[16]);
/* replaces Scene::clearMeshesAndMaterials (Scene.h:58) */
int hr_scene_clear(hr_ctx *ctx);
/* Acceleration-structure build: what OpenRL does behind rlDrawElements /
 * the next rlRenderFrame.  World transform of every
 * vertex (vertex.rlsl:25-43), Morton codes, radix sort, LBVH, refit. */
int hr_scene_commit(hr_ctx *ctx);

/* Acceleration-structure cache.  With a path set, a commit that would build the tree first looks for the file: if
 * it holds the tree of exactly this scene (a content hash of every mesh block — computed on the device — the transforms, modes and the
 * node format), the tree is read and uploaded instead of built, and only the triangles are re-assembled; otherwise the tree is built and
 * the file (re)written.  NULL or "" switches the cache off.  Worth it for large scenes (a 30 M-triangle build takes about a second); for
 * a million triangles the 5 ms device build is as fast as reading the file. */
int hr_scene_cache(hr_ctx *ctx, const char *path);

typedef struct hr_scene_info {
    uint64_t n_triangles;
    uint64_t n_nodes;
    float aabb_min[3];
    float aabb_max[3];
    float ray_epsilon; /* self-intersection t_min, 1e-4 * |aabb diagonal| */
    float build_ms;
    uint32_t bvh_levels; /* levels of inner nodes of the acceleration structure (0 for the oracle's brute force / a leaf root) */
    uint32_t refitted;   /* 1: the last commit kept the tree's topology and refitted its boxes (transform-only edits);
                          * 2: the tree came from the cache file (hr_scene_cache) */
    float box_area_ratio; /* quality of a refitted tree: (sum of its node boxes' areas / sum of its triangles' areas) relative to the
                           * value right after the last full build (1 after a build; a refit whose ratio would exceed 1.25 rebuilds
                           * instead); 0: unknown */
    uint32_t builder;     /* which binary tree the 4-wide tree was collapsed from: 0 the radix tree over Morton codes (LBVH), 1 PLOC */
    float cost_radix;     /* summed surface area of the 4-wide nodes / the root's, for the radix tree ...                              */
    float cost_ploc;      /* ... and for the PLOC tree (expected node visits of a random ray; the cheaper one is kept; 0: not built)    */
} hr_scene_info;
int hr_scene_get_info(hr_ctx *ctx, hr_scene_info *out);

/* --------------------------------------------------------------- textures */

/* values mirror the RL_* enums used in openrl::Texture::Descriptor / Sampler
 * (Source/RLWrapper/Texture.h:26-55) */
#define HR_TEX_U8 0  /* RL_UNSIGNED_BYTE */
#define HR_TEX_F32 1 /* RL_FLOAT         */
#define HR_WRAP_REPEAT 0
#define HR_WRAP_CLAMP_TO_EDGE 1
#define HR_FILTER_NEAREST 0
#define HR_FILTER_LINEAR 1 /* also RL_LINEAR_MIPMAP_LINEAR: sampled at LOD 0 */

typedef struct hr_texture_desc {
    int32_t width, height;
    int32_t channels; /* 1 (LUMINANCE), 3 (RGB), 4 (RGBA) */
    int32_t dtype;    /* HR_TEX_U8 (normalised /255) or HR_TEX_F32 */
    int32_t wrap_s, wrap_t;
    int32_t filter;
} hr_texture_desc;

typedef int32_t hr_tex_id;
#define HR_TEX_NONE (-1)

/* replaces openrl::Texture::create (Texture.h:69-93); row 0 = bottom row (GL) */
int hr_texture_create(hr_ctx *ctx, const hr_texture_desc *desc, const void *pixels, hr_tex_id *out);
int hr_texture_destroy(hr_ctx *ctx, hr_tex_id id);

/* --------------------------------------------------------------- materials */

#define HR_MAT_PBR 0   /* physicallyBased.rlsl */
#define HR_MAT_GLASS 1 /* glass.rlsl           */

/* shader permutation #defines become flag bits
 * (PhysicallyBasedMaterial.cpp:57-110, GlassMaterial.cpp:50-77) */
#define HR_MF_HAS_BASE_COLOR_TEXTURE (1u << 0)
#define HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE (1u << 1)
#define HR_MF_HAS_EMISSIVE_TEXTURE (1u << 2)
#define HR_MF_HAS_NORMALMAP (1u << 3)
#define HR_MF_HAS_CLEARCOAT_TEXTURE (1u << 4)
#define HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE (1u << 5)
#define HR_MF_HAS_CLEARCOAT_NORMALMAP (1u << 6)
#define HR_MF_DOUBLE_SIDED (1u << 7)
#define HR_MF_ALPHA_MASK (1u << 8)
#define HR_MF_VERTEX_COLORS (1u << 9)

/* One material row == the ShaderParams uniform block AFTER host-side baking
 * (PhysicallyBasedMaterial.cpp:16-36,127-191; GlassMaterial.cpp:15-28,88-126). */
typedef struct hr_material {
    int32_t type;
    uint32_t flags;
    hr_tex_id base_color_texture;
    hr_tex_id metallic_roughness_texture;
    hr_tex_id emissive_texture;
    hr_tex_id normalmap;
    hr_tex_id clear_coat_texture;
    hr_tex_id clear_coat_roughness_texture;
    hr_tex_id clear_coat_normalmap;
    hr_tex_id multiscatter_lut;
    float base_color[3];
    float emissive_color[3];
    float metallic;
    float roughness;
    float specular_f0;
    float roughness_alpha;
    float clear_coat;
    float clear_coat_roughness;
    float clear_coat_roughness_alpha;
    float ior;     /* glass */
    float density; /* glass */
} hr_material;

/* replaces Buffer::modify on the "Material" uniform block
 * (PhysicallyBasedMaterial.cpp:190, GlassMaterial.cpp:125) */
int hr_material_set(hr_ctx *ctx, int32_t material_id, const hr_material *m);

/* ------------------------------------------------------------------ lights */

/* Packed light blocks, mirroring ShaderLightingDefines.h:33-64 and
 * lightDefines.rlsl:16-47 (RL primitive handles dropped: a light is
 * addressed by (type,index)).  Colours are radiometric, directional
 * `directions` point TO the light, spot angles are cosines (x inner, y outer). */
typedef struct hr_lights {
    int32_t n_directional;
    float directional_directions[HR_MAX_DIRECTIONAL_LIGHTS][3];
    float directional_colors[HR_MAX_DIRECTIONAL_LIGHTS][3];
    int32_t n_point;
    float point_positions[HR_MAX_POINT_LIGHTS][3];
    float point_colors[HR_MAX_POINT_LIGHTS][3];
    int32_t n_spot;
    float spot_positions[HR_MAX_SPOT_LIGHTS][3];
    float spot_directions[HR_MAX_SPOT_LIGHTS][3];
    float spot_colors[HR_MAX_SPOT_LIGHTS][3];
    float spot_angles[HR_MAX_SPOT_LIGHTS][2];
    int32_t env_enabled;        /* EnvironmentLight.lightPrimitive != rl_NullPrimitive */
    hr_tex_id env_texture;      /* lat/long map                                        */
    float env_exposure;         /* 2^exposureCompensation (EnvironmentLight.cpp:95)    */
    float env_theta_rotation;   /* radians                                             */
} hr_lights;

/* replaces Lighting::update* buffer maps (Scene/Lighting.cpp:192-381) */
int hr_lights_set(hr_ctx *ctx, const hr_lights *lights);

/* --------------------------------------------------------------- QMC tables */

/* Every mode and shape has a device generator.  The three that the reference builds on its C++ library's <random> (RANDOM, and the polygonal
 * apertures) follow libstdc++'s distributions (GCC 11+; heatray_amd/csrc/hr_tables.h states them) — an application built against
 * another standard library that wants ITS tables uploads them with hr_sequences_set. */
#define HR_SAMPLE_RANDOM 0     /* PassGenerator.h:103 — util::uniformRandomFloats: std::mt19937 + uniform_real_distribution, Random.h:113-130 */
#define HR_SAMPLE_HALTON 1
#define HR_SAMPLE_HAMMERSLEY 2
#define HR_SAMPLE_BLUE_NOISE 3 /* util::blueNoise: best of 30 hashed candidates per point, BlueNoise.h:52-100 */
#define HR_SAMPLE_SOBOL 4

#define HR_BOKEH_CIRCULAR 0 /* radialSobol, Random.h:268-289 */
#define HR_BOKEH_PENTAGON 1 /* randomPolygonal(5 / 6 / 8 edges): std::mt19937 + uniform_int / uniform_real, Random.h:293-355 */
#define HR_BOKEH_HEXAGON 2
#define HR_BOKEH_OCTAGON 3

/* Upload host-generated tables.  replaces the RandomSequences /
 * RandomSequenceMetadata / ApertureSamples uniform blocks
 * (PassGenerator.cpp:603-684).  seq_xy, aperture_xy: n_seq*len vec2. */
int hr_sequences_set(hr_ctx *ctx, const float *seq_xy, const float *aperture_xy, int32_t n_seq, int32_t len);
/* replaces the SequenceOffsets uniform block (PassGenerator.cpp:150-159); n vec2 */
int hr_seq_offsets_set(hr_ctx *ctx, const float *offsets_xy, int32_t n);

/* Device-side generators of one table: util::sobol / halton / hammersley / blueNoise / uniformRandomFloats(seed = sequence_index)
 * (Random.h:85-265); radial != 0 (Sobol only) = util::radialSobol (Random.h:268-289).  out_xy: host, count vec2. */
int hr_qmc_generate(hr_ctx *ctx, int32_t mode, uint32_t sequence_index, uint32_t count, int32_t radial,
                    float *out_xy);
/* One aperture table: util::radialSobol (HR_BOKEH_CIRCULAR) or util::randomPolygonal(edges, count, seed = sequence_index)
 * (Random.h:268-355), as PassGenerator.cpp:653-676 calls them.  out_xy: host, count vec2. */
int hr_aperture_generate(hr_ctx *ctx, int32_t bokeh_shape, uint32_t sequence_index, uint32_t count, float *out_xy);
/* generateRandomSequences(P, mode, bokeh) entirely on device (PassGenerator.cpp:603-684): every mode, every shape. */
int hr_sequences_generate(hr_ctx *ctx, int32_t sample_mode, int32_t bokeh_shape, int32_t len);
/* generateSequenceOffsets(W,H) on device: sobol(W*H points, sequence 0) (PassGenerator.cpp:150-159) */
int hr_seq_offsets_generate(hr_ctx *ctx);
/* generateMultiScatterTexture on device (MultiScatterUtil.cpp:91-139): 128x128 R32F, out may be NULL.
 * Returns the texture id holding the LUT in *out_tex (may be NULL). */
int hr_multiscatter_lut_generate(hr_ctx *ctx, float *out_128x128, hr_tex_id *out_tex);

/* ------------------------------------------------------------------- pass */

/* Per-pass uniforms: the frame-program uniforms set in
 * PassGenerator::runRenderFrameJob (PassGenerator.cpp:349-369) plus the
 * Globals block (PassGenerator.h:269-294 / globalData.rlsl:9-34). */
typedef struct hr_pass_params {
    int32_t sample_index;      /* Globals.sampleIndex                           */
    int32_t max_ray_depth;     /* Globals.maxRayDepth                           */
    float max_channel_value;   /* Globals.maxChannelValue                       */
    float fov_tan;             /* tan(fovY/2), PassGenerator.cpp:341-357        */
    float aspect_ratio;
    float focus_distance;
    float aperture_radius;
    float view_matrix[16];     /* camera -> world, column-major                 */
    int32_t interactive_mode;  /* 3x3 block mode (perspective.rlsl:42-57)       */
    int32_t block_size[2];
    int32_t current_block_pixel[2];
    float max_sample_index;    /* float(maxRenderPasses)                        */
    int32_t enable_visualizer; /* Globals.enableVisualizer                      */
    int32_t visualizer_mode;   /* HR_VIS_* (one-hot show* flags of Globals)     */
    int32_t enable_accumulator_visualizer;
    int32_t show_nans;
    int32_t show_inf;
    int32_t estimator;         /* HR_ESTIMATOR_*: how next-event rays towards the environment are sampled */
    int32_t texture_lod;       /* HR_TEXTURE_LOD_*: which mip level material textures are read at        */
} hr_pass_params;

/* HR_ESTIMATOR_REFERENCE: the reference's estimator, bit for bit: when the light pick falls on the environment, the occlusion ray
 * is BRDF-sampled (microfacet.rlsl:25-50,100-151) — the shaders leave importance sampling of the IBL and MIS as TODOs
 * (lightSampling.rlsl:75-77, microfacet.rlsl:94-96).
 * HR_ESTIMATOR_ENV_MIS: that one ray is drawn either from the BRDF lobe or from a luminance x solid-angle
 * distribution over the environment map's texels (half the time each) and weighted with the balance heuristic (the one-sample
 * MIS estimator): same expectation, far less variance under small bright sources.  PBR materials, and the reflection branch of
 * glass (its next-event ray towards the map: glass.rlsl:83-129; BRDF x cos as :104-109 has it for analytic lights); everything
 * else — ray budget, light pick, lobe pick, analytic lights, glass transmission — is the reference's. */
#define HR_ESTIMATOR_REFERENCE 0
#define HR_ESTIMATOR_ENV_MIS 1
/* HR_ESTIMATOR_ALL_LIGHTS: HR_ESTIMATOR_ENV_MIS without the reference's remaining large variance term, the random choice of ONE
 * light per path vertex (lightSampling.rlsl:11-161: a scene lit by a sun and a sky of equal weight sees each at half the
 * vertices, at twice the value).  Every PBR vertex sends
 *   - one ray to an analytic light picked among the analytic lights only (directional / point / spot; a single such light is
 *     always sampled), carrying the WHOLE BSDF (diffuse + specular + clearcoat: no choice of lobe for a single direction), and
 *   - the MIS-weighted environment ray with probability 1 (the map's sampler 7 times out of 8 for the diffuse lobe); the hit of a
 *     camera ray sends three, a third of the value each, each with its own choice of lobe and its own sequence values.
 * No two rays of a launch may write the same pixel, so the pass's sample is kept as four partial sums (everything of the
 * reference path + the first environment ray; the analytic-light ray; the second and third environment ray), added in this
 * order when the pass resolves.  Up to three more occlusion rays per path.  Glass vertices send the analytic-light ray and the
 * MIS-weighted environment ray as well.  Own oracle
 * contract (bit-exact) and known-answer tests, like ENV_MIS. */
#define HR_ESTIMATOR_ALL_LIGHTS 2

/* HR_TEXTURE_LOD_BASE: every lookup reads level 0 (bilinear), as in rounds 1-2: OpenRL's level selection inside ray shaders (no
 * screen-space derivatives) is closed, so the reference-faithful path does not guess one.
 * HR_TEXTURE_LOD_CONE: the reference creates its textures with RL_LINEAR_MIPMAP_LINEAR and generated mips
 * (RLWrapper/Texture.h:51,68,85-87).  With this mode the core builds the mip chain on the device (2x2 box filter, levels >= 1 kept
 * as f32) the first time it is asked for, carries a ray cone along every path (width at the ray origin + spread angle: the pixel's
 * angle for camera rays, widened by 0.25 x roughness at every scattering event) and reads MATERIAL textures trilinearly at the
 * level whose texel matches the cone's footprint on the triangle (its world-to-uv area ratio, over |cos| of the incidence).
 * Environment, LUT and occlusion-ray alpha lookups stay at level 0.  Not the reference's estimator: it has its own oracle
 * contract (oracle/oracle_shade.cpp, bit-exact) and known-answer tests. */
#define HR_TEXTURE_LOD_BASE 0
#define HR_TEXTURE_LOD_CONE 1

/* one-hot show* flags of GlobalData (PassGenerator.h:275-289) as an enum */
#define HR_VIS_NONE 0
#define HR_VIS_GEOMETRIC_NORMALS 1
#define HR_VIS_UVS 2
#define HR_VIS_TANGENTS 3
#define HR_VIS_BITANGENTS 4
#define HR_VIS_NORMALMAP 5
#define HR_VIS_FINAL_NORMALS 6
#define HR_VIS_BASE_COLOR 7
#define HR_VIS_ROUGHNESS 8
#define HR_VIS_METALLIC 9
#define HR_VIS_EMISSIVE 10
#define HR_VIS_CLEARCOAT 11
#define HR_VIS_CLEARCOAT_ROUGHNESS 12
#define HR_VIS_CLEARCOAT_NORMALMAP 13
#define HR_VIS_SHADER 14

/* Device-side counters.  OpenRL exposes the same kind of statistics
 * (rl.h:346-355: RL_RENDER_FRAME_TIME, RL_EMITTED_RAY_COUNT). */
typedef struct hr_pass_stats {
    float ms;               /* hipEvent time of the pass (0 unless requested)       */
    uint64_t paths;         /* primary rays generated                                */
    uint64_t rays_closest;  /* closest-hit traversals launched                       */
    uint64_t rays_any;      /* occlusion traversals launched                         */
    uint64_t shaded_hits;   /* material shader invocations                           */
    uint64_t accumulates;   /* accumulate() calls with RGB payload                   */
    uint64_t node_visits;   /* BVH nodes fetched, all rays   (HR_CTX_COLLECT_STATS only) */
    uint64_t tri_tests;     /* triangle tests, all rays      (HR_CTX_COLLECT_STATS only) */
    uint64_t node_visits_any; /* ... of which by occlusion rays                          */
    uint64_t tri_tests_any;
} hr_pass_stats;

/* Per-kernel device time, summed over every launch since the last hr_clear: ms[] / launches[] from HIP event pairs
 * around the kernels (HR_CTX_TIME_KERNELS contexts only, else zero: every record is a packet of a few microseconds on
 * the stream, which a small tile shard's dependent launches feel), trace_clock_* from the constant 100 MHz device clock
 * read inside k_trace itself (first workgroup's start to last workgroup's end; always on, nothing on the stream).
 * OpenRL's counterpart: RL_RENDER_FRAME_TIME / RL_PROFILE (rl.h:346-355), which Heatray never queries.
 * camera_packets / packet_union: the camera rays (perspective.rlsl:39-93) of the passes injected together are either traced like
 * every other ray or, where a probe finds the rays of a few neighbouring pixels in these passes walking nearly the same nodes, as one
 * packet per wave (one node fetch and one stack for 64 rays; ray generation and this traversal are then one kernel, timed under
 * HR_KERNEL_RAYGEN — where rays walk far it runs BESIDE k_trace on a second stream, and that bucket holds only the time it outlasts
 * k_trace); the probe runs beside the pipeline after a commit, a resize, a change of camera and every 64th batch.
 * The hits — and so the image — are the same bits either way, by construction: whether a ray hits a triangle is decided by the
 * triangle test alone (Moeller-Trumbore plus "the hit point lies in the triangle's half-padded box"), never by which
 * triangles a traversal happens to test; HR_TUNE packets=0|1 pins the mode all the same. */
#define HR_KERNEL_RAYGEN 0  /* ray generation; with camera_packets, the camera rays' traversal too */
#define HR_KERNEL_TRACE 1   /* closest-hit + occlusion traversal (one kernel) */
#define HR_KERNEL_SHADE 2
#define HR_KERNEL_RESOLVE 3 /* finished pass samples -> accumulation buffer */
#define HR_KERNEL_COUNT 4
typedef struct hr_kernel_times {
    float ms[HR_KERNEL_COUNT];
    uint32_t launches[HR_KERNEL_COUNT];
    float trace_clock_ms;          /* k_trace (with the packet kernel in front of it, when camera rays go that way), device clock */
    uint32_t trace_clock_launches;
    /* how the camera rays are traced at the moment (chosen by a measurement, see below) and the measurement itself */
    uint32_t camera_packets;       /* n > 0: 64 camera rays — n passes of 64 / n neighbouring pixels — walk the tree as one packet; 0: one ray per lane like every other ray */
    float packet_union;            /* last probe: node tests a packet makes per ray / node tests the ray's own traversal needs (0: no probe yet) */
} hr_kernel_times;

/* The pipeline's own timeline — what the reference shows as "pass time" in its UI (HeatrayRenderer.cpp:960, the passTime argument of
 * PassCompleteCallback, PassGenerator.h:161) and OpenRL would report through RL_RENDER_FRAME_TIME (rl.h:346-355), per step instead of
 * per pass because passes overlap here.  One record per macro step of the pass pipeline since the last hr_clear (the newest 4096): when its k_trace launch started and how
 * long it ran, by the device clock, how many passes were in flight in it and how many of them it injected.  A step in which the
 * pipeline is full (passes_in_flight = stages x batch) does one pass's worth of work per injected pass: the period between such
 * steps / passes_injected is the steady-state time per pass; the steps before and behind are the pipeline's fill and drain. */
typedef struct hr_step_record {
    double start_ms;   /* relative to the first record */
    float trace_ms;
    int32_t passes_in_flight;
    int32_t passes_injected;
    int32_t group;     /* pipeline group (worker stream) the step ran on; records are sorted by start_ms, whichever group they belong to */
} hr_step_record;
int hr_get_step_log(hr_ctx *ctx, hr_step_record *out, int32_t capacity, int32_t *n_records);

/* Interactive 3x3 mode (perspective.rlsl:42-57): which pixel of a block is sampled in a sub-pass is looked up in a small table,
 * the reference's interactiveBlockSamplesTexture (PassGenerator.cpp:267-294: the nx x ny list of (row, col) pairs, shuffled
 * with std::random_device, uploaded as an RL_NEAREST / RL_REPEAT RGB texture).  coords_xy holds nx*ny integer pairs in the
 * texture's memory order (texel (tx, ty) at index ty*nx + tx); the pair is what the shader reads as `.xy`.  NULL restores
 * the unshuffled list (pair (ty, tx) at texel (tx, ty)), which is also the state of a new context.  nx*ny <= 16. */
int hr_interactive_blocks_set(hr_ctx *ctx, const int32_t *coords_xy, int32_t nx, int32_t ny);

/* replaces rlClear(RL_COLOR_BUFFER_BIT) (PassGenerator.cpp:439) */
int hr_clear(hr_ctx *ctx);
/* replaces rlRenderFrame() (PassGenerator.cpp:386): one sample per owned pixel.
 * Asynchronous, and PIPELINED: passes are kept in flight at different bounce stages
 * (max_ray_depth+2 stages per pass; on a small tile shard several passes are collected
 * and injected together; the ray kernels run on internal streams ordered against the
 * ctx stream by events).  A pass's sample reaches the accumulation buffer when its last
 * stage has run — on the ctx stream, always in pass order — so the buffer only ever
 * holds complete passes and its alpha channel says how many.
 * hr_readback / hr_display / hr_get_stats / hr_clear / hr_synchronize / hr_flush and
 * every call that changes scene state complete all enqueued passes first. */
int hr_render_pass(hr_ctx *ctx, const hr_pass_params *params);
/* How many passes hr_render_pass collects before it launches them together at this frame size and depth (1 = every call
 * launches).  The accumulation buffer changes once per batch, so this is also the cadence at which a progressive display or a
 * tile exchange sees new samples.  hr_flush / hr_readback launch whatever is pending. */
int hr_frame_pass_batch(hr_ctx *ctx, int32_t max_ray_depth, int32_t *batch);
/* enqueue the remaining stages of every pass in flight (asynchronous): after it, work
 * the caller puts on the ctx stream sees every requested sample in the buffer */
int hr_flush(hr_ctx *ctx);
/* synchronises the stream, then copies the counters */
int hr_get_stats(hr_ctx *ctx, hr_pass_stats *out);
/* completes the enqueued passes, synchronises, then sums the recorded event pairs and the device-clock totals */
int hr_get_kernel_times(hr_ctx *ctx, hr_kernel_times *out);
/* replaces PixelPackBuffer::setPixelData + mapPixelData (PixelPackBuffer.h:39-60):
 * synchronous copy to a pinned host buffer owned by the ctx; the pointer stays
 * valid until the next hr_readback / hr_frame_resize / hr_ctx_destroy. */
int hr_readback(hr_ctx *ctx, const float **rgba, int32_t *width, int32_t *height);
/* Progressive variant for a viewer that refreshes while passes accumulate: does NOT complete the passes still in the
 * pipeline (hr_readback does, which costs the pipeline's depth in latency for every displayed frame); copies the
 * accumulation buffer as it stands — complete passes only, `passes_in_buffer` of them since the last hr_clear (the alpha
 * channel holds the same number).  Snapshots rotate through three pinned buffers and, while passes are in flight, the one
 * handed out is the previous call's (so the host never waits for work it has only just enqueued); a pointer stays valid for
 * two further calls.  Returns passes_in_buffer = 0 while the first passes after a clear are still in flight: call
 * hr_readback then (PixelPackBuffer::setPixelData does). */
int hr_readback_progressive(hr_ctx *ctx, const float **rgba, int32_t *width, int32_t *height, uint32_t *passes_in_buffer);
int hr_synchronize(hr_ctx *ctx);

/* ------------------------------------------------------------------ display resolve
 * The step right after the path: the reference copies the RGBA32F accumulation buffer to the host
 * (mapPixelData), uploads it through a PBO (HeatrayRenderer.cpp:328-344, 33 MB per displayed frame at 1080p)
 * and runs Resources/shaders/displayGL.frag:74-151 on the GL device.  hr_display runs that fragment shader
 * on the MI355X instead, straight from the accumulation buffer: divide by the sample count in alpha,
 * optional ACES tonemapping, brightness / contrast, hue / saturation / vibrance, RGB levels, vignette,
 * exposure, sRGB encoding — and hands over display-ready pixels (RGBA8: 4 x fewer bytes over PCIe). */
typedef struct hr_display_params {            /* HeatrayRenderer.h:104-117 PostProcessingParams, as uploaded by */
    int32_t tonemapping_enabled;              /* DisplayProgram::bind (HeatrayRenderer.h:222-248)               */
    float camera_exposure;                    /* = 2^exposure (bind computes std::powf(2, exposure))             */
    float brightness, contrast, hue, saturation, vibrance;
    float red, green, blue;
    float vignette_intensity, vignette_falloff;
} hr_display_params;

#define HR_DISPLAY_RGBA8 0       /* uint8 x 4, sRGB-encoded, alpha 255: what the GL framebuffer would hold        */
#define HR_DISPLAY_RGBA32F 1     /* float x 4, the shader's fragColor before framebuffer conversion               */
#define HR_DISPLAY_HDR_RGBA32F 2 /* float x 4, rgb * (1 / a), a unchanged: saveScreenshot's HDR path (.cpp:1624-1645) */
#define HR_DISPLAY_PROGRESSIVE 0x100 /* OR into `format`: show the passes that are complete already instead of completing
                                      * the ones still in the pipeline (see hr_readback_progressive) */

/* Completes all enqueued passes (unless HR_DISPLAY_PROGRESSIVE), then writes width*height pixels (row 0 = bottom, like the accumulation
 * buffer) to device memory `device_out` (asynchronous on the ctx stream). Pixels without samples (a == 0) give
 * colour 0.  Pixels owned by other ranks of a tile-sharded frame are written as 0.
 * `passes_shown` (may be NULL) receives the number of complete passes since the last hr_clear that the image holds — with
 * HR_DISPLAY_PROGRESSIVE that is whatever had been resolved when the call was made, and only the call itself can say which
 * (a later hr_readback_progressive may already see more). */
int hr_display(hr_ctx *ctx, const hr_display_params *params, int32_t format, void *device_out, uint32_t *passes_shown);
/* Same, into a pinned host buffer owned by the ctx (synchronous); valid until the next hr_display_readback /
 * hr_frame_resize / hr_ctx_destroy.  Replaces the mapPixelData -> glBufferSubData upload of the HDR buffer. */
int hr_display_readback(hr_ctx *ctx, const hr_display_params *params, int32_t format, const void **pixels, int32_t *width,
                        int32_t *height, uint32_t *passes_shown);
/* Passes whose sample has been added to the accumulation buffer since the last hr_clear, as enqueued on the ctx stream so far
 * (a host-side counter: no synchronisation).  Work enqueued on the ctx stream after this call sees exactly that many passes.
 * The reference's counterpart is the passIndex its PassCompleteCallback hands the viewer (PassGenerator.h:161,
 * PassGenerator.cpp:390-400: one callback per pass, because rlRenderFrame is synchronous there). */
int hr_frame_passes_resolved(hr_ctx *ctx, uint64_t *passes);

/* ------------------------------------------------------------------ tile-shard exchange
 * The "RCCL reduce of the HDR accumulation buffer" of the north-star, done as a gather of the tiles each rank
 * owns (world x fewer bytes than summing full frames, and exact: pixels are copied, never added).  These two
 * calls are the device-side ends of it; the collective in between is the caller's (bench.py: RCCL gather).
 * Pixel order = the order libhrcore enumerates a rank's pixels in (tile by tile, 8x8 blocks inside a tile);
 * both ends use it, nobody else needs to know it.  Slots of pixels outside a cropped edge tile hold zeros. */
/* number of float4 slots hr_frame_pack_owned writes for `rank` of `world` at the ctx's frame size and tile size */
int hr_frame_packed_slots(hr_ctx *ctx, int32_t rank, int32_t world, uint64_t *n_slots);
/* this context's owned pixels -> device_out (n_slots x RGBA32F); asynchronous on `stream` (a hipStream_t; NULL = the
 * ctx stream).  Either way the copy is ordered after the passes resolved so far and before the next resolve (on a
 * foreign stream through events) — the pipeline is NOT drained: progressive display */
int hr_frame_pack_owned(hr_ctx *ctx, void *device_out, void *stream);
/* rank `src_rank`'s packed pixels -> their place in a full-frame RGBA32F buffer (width*height*16 B); asynchronous on
 * `stream` (NULL = the ctx stream), e.g. the side stream the collective ran on */
int hr_frame_unpack(hr_ctx *ctx, int32_t src_rank, int32_t world, const void *device_packed, void *device_full_frame, void *stream);

/* ------------------------------------------------------------------ debug */

typedef struct hr_hit {
    int32_t prim; /* global triangle index in submission order, -1 = miss */
    float t, u, v;
} hr_hit;

/* Trace n caller-supplied rays (closest hit; any_hit != 0 -> occlusion query,
 * prim = 0 blocked / -1 clear).  Used by the parity tests of the traversal
 * against the oracle's brute-force intersector. */
int hr_debug_trace(hr_ctx *ctx, int32_t n, const float *origins_xyz, const float *dirs_xyz, const float *tmax,
                   const int32_t *skip_prim, int32_t any_hit, hr_hit *out);

#ifdef __cplusplus
}
#endif
#endif /* HRCORE_H */
