"""ctypes view of include/hrcore.h.

`Engine` drives any shared library that exports the C-ABI of include/hrcore.h under
a given symbol prefix.  The product binds it to libhrcore.so (prefix ``hr_``, see
heatray_amd.core); the test-suite binds the same class to the CPU oracle (prefix
``ora_``) so both are fed identical POD inputs.  Nothing in this package loads the
oracle.
"""
import ctypes as C

import numpy as np

HR_OK = 0
HR_MAX_LIGHTS = 5
HR_NUM_RANDOM_SEQUENCES = 16

HR_CTX_COLLECT_STATS = 1
HR_CTX_TIME_KERNELS = 2
HR_KERNEL_NAMES = ("raygen", "trace", "shade", "resolve")

HR_TRIANGLES, HR_TRIANGLE_STRIP = 0, 1
HR_TEX_U8, HR_TEX_F32 = 0, 1
HR_WRAP_REPEAT, HR_WRAP_CLAMP_TO_EDGE = 0, 1
HR_FILTER_NEAREST, HR_FILTER_LINEAR = 0, 1
HR_MAT_PBR, HR_MAT_GLASS = 0, 1

HR_MF_HAS_BASE_COLOR_TEXTURE = 1 << 0
HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE = 1 << 1
HR_MF_HAS_EMISSIVE_TEXTURE = 1 << 2
HR_MF_HAS_NORMALMAP = 1 << 3
HR_MF_HAS_CLEARCOAT_TEXTURE = 1 << 4
HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE = 1 << 5
HR_MF_HAS_CLEARCOAT_NORMALMAP = 1 << 6
HR_MF_DOUBLE_SIDED = 1 << 7
HR_MF_ALPHA_MASK = 1 << 8
HR_MF_VERTEX_COLORS = 1 << 9

HR_SAMPLE_RANDOM, HR_SAMPLE_HALTON, HR_SAMPLE_HAMMERSLEY, HR_SAMPLE_BLUE_NOISE, HR_SAMPLE_SOBOL = range(5)
HR_BOKEH_CIRCULAR, HR_BOKEH_PENTAGON, HR_BOKEH_HEXAGON, HR_BOKEH_OCTAGON = range(4)

HR_ESTIMATOR_REFERENCE, HR_ESTIMATOR_ENV_MIS, HR_ESTIMATOR_ALL_LIGHTS = 0, 1, 2
HR_TEXTURE_LOD_BASE, HR_TEXTURE_LOD_CONE = 0, 1
(HR_VIS_NONE, HR_VIS_GEOMETRIC_NORMALS, HR_VIS_UVS, HR_VIS_TANGENTS, HR_VIS_BITANGENTS, HR_VIS_NORMALMAP,
 HR_VIS_FINAL_NORMALS, HR_VIS_BASE_COLOR, HR_VIS_ROUGHNESS, HR_VIS_METALLIC, HR_VIS_EMISSIVE, HR_VIS_CLEARCOAT,
 HR_VIS_CLEARCOAT_ROUGHNESS, HR_VIS_CLEARCOAT_NORMALMAP, HR_VIS_SHADER) = range(15)

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class CtxDesc(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("tile_size", C.c_int32),
                ("stream", C.c_void_p), ("flags", C.c_uint32), ("memory_budget", C.c_uint64)]


class MeshDesc(C.Structure):
    _fields_ = [("positions", f32p), ("normals", f32p), ("uvs", f32p), ("tangents", f32p), ("bitangents", f32p),
                ("colors", f32p), ("position_stride", C.c_int32), ("normal_stride", C.c_int32),
                ("uv_stride", C.c_int32), ("tangent_stride", C.c_int32), ("bitangent_stride", C.c_int32),
                ("color_stride", C.c_int32), ("n_vertices", C.c_int32), ("indices", u32p), ("n_indices", C.c_int32),
                ("mode", C.c_int32), ("world_from_entity", C.c_float * 16), ("front_face_cw", C.c_int32),
                ("is_occluder", C.c_int32), ("material_id", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("n_nodes", C.c_uint64), ("aabb_min", C.c_float * 3),
                ("aabb_max", C.c_float * 3), ("ray_epsilon", C.c_float), ("build_ms", C.c_float),
                ("bvh_levels", C.c_uint32), ("refitted", C.c_uint32),
                ("box_area_ratio", C.c_float), ("builder", C.c_uint32), ("cost_radix", C.c_float), ("cost_ploc", C.c_float)]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("dtype", C.c_int32),
                ("wrap_s", C.c_int32), ("wrap_t", C.c_int32), ("filter", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("flags", C.c_uint32), ("base_color_texture", C.c_int32),
                ("metallic_roughness_texture", C.c_int32), ("emissive_texture", C.c_int32), ("normalmap", C.c_int32),
                ("clear_coat_texture", C.c_int32), ("clear_coat_roughness_texture", C.c_int32),
                ("clear_coat_normalmap", C.c_int32), ("multiscatter_lut", C.c_int32), ("base_color", C.c_float * 3),
                ("emissive_color", C.c_float * 3), ("metallic", C.c_float), ("roughness", C.c_float),
                ("specular_f0", C.c_float), ("roughness_alpha", C.c_float), ("clear_coat", C.c_float),
                ("clear_coat_roughness", C.c_float), ("clear_coat_roughness_alpha", C.c_float), ("ior", C.c_float),
                ("density", C.c_float)]


V3x5 = (C.c_float * 3) * HR_MAX_LIGHTS
V2x5 = (C.c_float * 2) * HR_MAX_LIGHTS


class Lights(C.Structure):
    _fields_ = [("n_directional", C.c_int32), ("directional_directions", V3x5), ("directional_colors", V3x5),
                ("n_point", C.c_int32), ("point_positions", V3x5), ("point_colors", V3x5), ("n_spot", C.c_int32),
                ("spot_positions", V3x5), ("spot_directions", V3x5), ("spot_colors", V3x5), ("spot_angles", V2x5),
                ("env_enabled", C.c_int32), ("env_texture", C.c_int32), ("env_exposure", C.c_float),
                ("env_theta_rotation", C.c_float)]


class PassParams(C.Structure):
    _fields_ = [("sample_index", C.c_int32), ("max_ray_depth", C.c_int32), ("max_channel_value", C.c_float),
                ("fov_tan", C.c_float), ("aspect_ratio", C.c_float), ("focus_distance", C.c_float),
                ("aperture_radius", C.c_float), ("view_matrix", C.c_float * 16), ("interactive_mode", C.c_int32),
                ("block_size", C.c_int32 * 2), ("current_block_pixel", C.c_int32 * 2), ("max_sample_index", C.c_float),
                ("enable_visualizer", C.c_int32), ("visualizer_mode", C.c_int32),
                ("enable_accumulator_visualizer", C.c_int32), ("show_nans", C.c_int32), ("show_inf", C.c_int32),
                ("estimator", C.c_int32), ("texture_lod", C.c_int32)]


class PassStats(C.Structure):
    _fields_ = [("ms", C.c_float), ("paths", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("accumulates", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("node_visits_any", C.c_uint64), ("tri_tests_any", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class StepRecord(C.Structure):
    _fields_ = [("start_ms", C.c_double), ("trace_ms", C.c_float), ("passes_in_flight", C.c_int32), ("passes_injected", C.c_int32), ("group", C.c_int32)]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_float * 4), ("launches", C.c_uint32 * 4), ("trace_clock_ms", C.c_float), ("trace_clock_launches", C.c_uint32),
                ("camera_packets", C.c_uint32), ("packet_union", C.c_float)]


class DisplayParams(C.Structure):
    """hr_display_params: HeatrayRenderer.h:104-117 PostProcessingParams as DisplayProgram::bind uploads them."""
    _fields_ = [("tonemapping_enabled", C.c_int32), ("camera_exposure", C.c_float), ("brightness", C.c_float),
                ("contrast", C.c_float), ("hue", C.c_float), ("saturation", C.c_float), ("vibrance", C.c_float),
                ("red", C.c_float), ("green", C.c_float), ("blue", C.c_float), ("vignette_intensity", C.c_float),
                ("vignette_falloff", C.c_float)]


def display_params(tonemapping_enabled=False, exposure=0.0, brightness=0.0, contrast=1.0, hue=1.0, saturation=1.0,
                   vibrance=0.0, red=1.0, green=1.0, blue=1.0, vignette_intensity=0.0, vignette_falloff=1.0):
    """Defaults = the reference's PostProcessingParams defaults; camera_exposure = 2^exposure (computed in binary32 by
    repeated doubling / halving for integral exposures, else through numpy's float32 power, as std::powf would)."""
    return DisplayParams(int(bool(tonemapping_enabled)), float(np.float32(2.0) ** np.float32(exposure)), brightness, contrast, hue,
                         saturation, vibrance, red, green, blue, vignette_intensity, vignette_falloff)


HR_DISPLAY_RGBA8, HR_DISPLAY_RGBA32F, HR_DISPLAY_HDR_RGBA32F = 0, 1, 2
HR_DISPLAY_PROGRESSIVE = 0x100


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("u", C.c_float), ("v", C.c_float)]


HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("u", np.float32), ("v", np.float32)])

# every symbol include/hrcore.h declares (suffix after the prefix)
HR_ABI_VERSION = 6  # include/hrcore.h: checked against the loaded library before the first call (Engine.__init__)
ABI_SYMBOLS = [
    "abi_version", "ctx_create", "ctx_destroy", "last_error", "ctx_set_stream", "frame_resize", "frame_bind_external",
    "frame_device_ptr", "geom_add", "geom_remove", "geom_set_transform", "scene_clear", "scene_commit",
    "scene_get_info", "texture_create", "texture_destroy", "material_set", "lights_set", "sequences_set",
    "seq_offsets_set", "qmc_generate", "aperture_generate", "sequences_generate", "seq_offsets_generate", "multiscatter_lut_generate",
    "clear", "render_pass", "flush", "get_stats", "get_kernel_times", "readback", "synchronize", "debug_trace",
    "display", "display_readback", "frame_passes_resolved", "get_step_log", "readback_progressive", "frame_packed_slots", "frame_pack_owned", "frame_unpack",
    "frame_pass_batch", "interactive_blocks_set", "scene_cache",
]


class EngineError(RuntimeError):
    pass


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, typ=f32p):
    return a.ctypes.data_as(typ) if a is not None else typ()


class Engine:
    """Thin object wrapper over one hr_ctx (or ora_ctx)."""

    def __init__(self, lib, prefix, device_id=0, rank=0, world=1, tile_size=32, stream=None, flags=0, memory_budget=0):
        self._lib = lib
        self._p = prefix
        self._ctx = C.c_void_p()
        ver = getattr(lib, prefix + "abi_version", None)
        if ver is None:
            raise EngineError(f"{prefix}abi_version missing: the library predates this binding (ABI {HR_ABI_VERSION})")
        ver.restype = C.c_uint32
        if ver() != HR_ABI_VERSION:
            raise EngineError(f"{prefix}abi_version() = {ver()}, this binding was written against {HR_ABI_VERSION}: rebuild the library")
        desc = CtxDesc(device_id, rank, world, tile_size, stream, flags, int(memory_budget))
        rc = self._fn("ctx_create")(C.byref(desc), C.byref(self._ctx))
        if rc != HR_OK:
            self._ctx = C.c_void_p()
            raise EngineError(f"{prefix}ctx_create failed with status {rc} (no usable HIP device?)")
        self.width = self.height = 0

    # -- plumbing
    def _fn(self, name):
        fn = getattr(self._lib, self._p + name)
        fn.restype = C.c_int
        return fn

    def _call(self, name, *args):
        rc = self._fn(name)(self._ctx, *args)
        if rc != HR_OK:
            le = getattr(self._lib, self._p + "last_error")
            le.restype = C.c_char_p
            msg = le(self._ctx)
            raise EngineError(f"{self._p}{name}: status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if self._ctx:
            self._fn("ctx_destroy")(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- frame
    def resize(self, w, h):
        self._call("frame_resize", C.c_int32(w), C.c_int32(h))
        self.width, self.height = w, h

    def bind_external_frame(self, device_ptr):
        self._call("frame_bind_external", C.c_void_p(device_ptr))

    def frame_device_ptr(self):
        p = C.c_void_p()
        self._call("frame_device_ptr", C.byref(p))
        return p.value

    def set_stream(self, stream):
        self._call("ctx_set_stream", C.c_void_p(stream))

    # -- geometry
    def add_mesh(self, positions, normals, indices, uvs=None, tangents=None, bitangents=None, colors=None,
                 mode=HR_TRIANGLES, world=None, front_face_cw=None, is_occluder=True, material_id=0):
        pos, nrm = _f32(positions).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        opt = [None if a is None else _f32(a) for a in (uvs, tangents, bitangents, colors)]
        m = np.eye(4, dtype=np.float32) if world is None else _f32(world).reshape(4, 4)
        if front_face_cw is None:  # Mesh.cpp:86-91: determinant of the full 4x4
            front_face_cw = bool(np.linalg.det(m.astype(np.float64)) < 0)
        d = MeshDesc()
        d.positions, d.normals = _ptr(pos), _ptr(nrm)
        d.uvs, d.tangents, d.bitangents, d.colors = (_ptr(a) for a in opt)
        d.n_vertices = pos.shape[0]
        d.indices, d.n_indices, d.mode = _ptr(idx, u32p), idx.size, mode
        # numpy holds the matrix as m[row][col]; the ABI wants column-major storage
        d.world_from_entity = (C.c_float * 16)(*m.T.reshape(-1))
        d.front_face_cw, d.is_occluder, d.material_id = int(front_face_cw), int(is_occluder), material_id
        gid = C.c_int32()
        self._call("geom_add", C.byref(d), C.byref(gid))
        return gid.value

    def add_mesh_strided(self, buf, pos_off, nrm_off, uv_off, stride_floats, indices, mode=HR_TRIANGLES, world=None, is_occluder=True,
                         material_id=0):
        """One interleaved float buffer (n, stride_floats) holding positions / normals / uvs at the given float offsets."""
        buf = np.ascontiguousarray(buf, dtype=np.float32)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        d = MeshDesc()
        base = buf.ctypes.data
        d.positions = C.cast(base + 4 * pos_off, f32p)
        d.normals = C.cast(base + 4 * nrm_off, f32p)
        if uv_off is not None:
            d.uvs = C.cast(base + 4 * uv_off, f32p)
        d.position_stride = d.normal_stride = d.uv_stride = 4 * stride_floats
        d.n_vertices = buf.shape[0]
        d.indices, d.n_indices, d.mode = _ptr(idx, u32p), idx.size, mode
        m = np.eye(4, dtype=np.float32) if world is None else _f32(world).reshape(4, 4)
        d.world_from_entity = (C.c_float * 16)(*m.T.reshape(-1))
        d.front_face_cw, d.is_occluder, d.material_id = int(np.linalg.det(m.astype(np.float64)) < 0), int(is_occluder), material_id
        gid = C.c_int32(-1)
        self._call("geom_add", C.byref(d), C.byref(gid))
        return gid.value

    def remove_mesh(self, gid):
        self._call("geom_remove", C.c_int32(gid))

    def set_transform(self, gid, world):
        m = _f32(world).reshape(4, 4)
        self._call("geom_set_transform", C.c_int32(gid), (C.c_float * 16)(*m.T.reshape(-1)))

    def clear_scene(self):
        self._call("scene_clear")

    def set_scene_cache(self, path):
        self._call("scene_cache", None if not path else str(path).encode())

    def commit(self):
        self._call("scene_commit")

    def scene_info(self):
        s = SceneInfo()
        self._call("scene_get_info", C.byref(s))
        return s

    # -- textures / materials / lights
    def create_texture(self, pixels, wrap=HR_WRAP_REPEAT, filter=HR_FILTER_LINEAR):
        a = np.ascontiguousarray(pixels)
        if a.ndim == 2:
            a = a[:, :, None]
        dtype = HR_TEX_U8 if a.dtype == np.uint8 else HR_TEX_F32
        if dtype == HR_TEX_F32:
            a = _f32(a)
        d = TextureDesc(a.shape[1], a.shape[0], a.shape[2], dtype, wrap, wrap, filter)
        tid = C.c_int32()
        self._call("texture_create", C.byref(d), a.ctypes.data_as(C.c_void_p), C.byref(tid))
        return tid.value

    def destroy_texture(self, tid):
        self._call("texture_destroy", C.c_int32(tid))

    def set_material(self, material_id, material):
        self._call("material_set", C.c_int32(material_id), C.byref(material))

    def set_lights(self, lights):
        self._call("lights_set", C.byref(lights))

    # -- sample tables
    def set_sequences(self, seq_xy, aperture_xy):
        s, a = _f32(seq_xy), _f32(aperture_xy)
        assert s.shape == a.shape and s.ndim == 3 and s.shape[2] == 2
        self._call("sequences_set", _ptr(s), _ptr(a), C.c_int32(s.shape[0]), C.c_int32(s.shape[1]))

    def set_seq_offsets(self, offsets_xy):
        o = _f32(offsets_xy).reshape(-1, 2)
        self._call("seq_offsets_set", _ptr(o), C.c_int32(o.shape[0]))

    def qmc_generate(self, mode, sequence_index, count, radial=False):
        out = np.empty((count, 2), dtype=np.float32)
        self._call("qmc_generate", C.c_int32(mode), C.c_uint32(sequence_index), C.c_uint32(count),
                   C.c_int32(int(radial)), _ptr(out))
        return out

    def aperture_generate(self, bokeh, sequence_index, count):
        """One aperture table as PassGenerator.cpp:653-676 makes it: radialSobol or randomPolygonal(5 / 6 / 8 edges, seed = sequence)."""
        out = np.empty((count, 2), dtype=np.float32)
        self._call("aperture_generate", C.c_int32(bokeh), C.c_uint32(sequence_index), C.c_uint32(count), _ptr(out))
        return out

    def generate_sequences(self, sample_mode=HR_SAMPLE_SOBOL, bokeh=HR_BOKEH_CIRCULAR, length=32):
        self._call("sequences_generate", C.c_int32(sample_mode), C.c_int32(bokeh), C.c_int32(length))

    def generate_seq_offsets(self):
        self._call("seq_offsets_generate")

    def generate_multiscatter_lut(self, want_host=True):
        out = np.empty((128, 128), dtype=np.float32) if want_host else None
        tid = C.c_int32()
        self._call("multiscatter_lut_generate", _ptr(out), C.byref(tid))
        return out, tid.value

    # -- rendering
    def clear(self):
        self._call("clear")

    def render_pass(self, params):
        self._call("render_pass", C.byref(params))

    def set_interactive_blocks(self, coords_xy):
        """coords_xy: (ny, nx, 2) integer array in texture memory order, or None for the unshuffled list."""
        if coords_xy is None:
            self._call("interactive_blocks_set", None, C.c_int32(0), C.c_int32(0))
            return
        a = np.ascontiguousarray(coords_xy, dtype=np.int32)
        self._call("interactive_blocks_set", a.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(a.shape[1]), C.c_int32(a.shape[0]))

    def pass_batch(self, max_ray_depth):
        b = C.c_int32(0)
        self._call("frame_pass_batch", C.c_int32(max_ray_depth), C.byref(b))
        return int(b.value)

    def flush(self):
        self._call("flush")

    def stats(self):
        s = PassStats()
        self._call("get_stats", C.byref(s))
        return s

    def kernel_times(self):
        """{kernel: (total_ms, launches)} since the last clear: HIP-event times (zero unless the context was created with
        HR_CTX_TIME_KERNELS), under "trace_clock" k_trace's time by the device clock (always), under "camera_packets" whether camera
        rays are traced as packets at the moment and the union factor the selector's last probe measured."""
        t = KernelTimes()
        self._call("get_kernel_times", C.byref(t))
        d = {n: (t.ms[i], t.launches[i]) for i, n in enumerate(HR_KERNEL_NAMES)}
        d["trace_clock"] = (t.trace_clock_ms, t.trace_clock_launches)
        d["camera_packets"] = (bool(t.camera_packets), t.packet_union, t.camera_packets)  # (camera rays as packets now?, the probe's union factor, passes per packet)
        return d

    def step_log(self, capacity=4096):
        """[(start_ms, trace_ms, passes_in_flight, passes_injected, group)] of the macro steps since the last clear (newest 4096), sorted by start."""
        recs = (StepRecord * capacity)()
        n = C.c_int32()
        self._call("get_step_log", recs, C.c_int32(capacity), C.byref(n))
        return [(r.start_ms, r.trace_ms, r.passes_in_flight, r.passes_injected, r.group) for r in recs[: n.value]]

    def synchronize(self):
        self._call("synchronize")

    def readback(self, copy=True):
        p = f32p()
        w, h = C.c_int32(), C.c_int32()
        self._call("readback", C.byref(p), C.byref(w), C.byref(h))
        a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4))
        return a.copy() if copy else a

    def display(self, params=None, fmt=HR_DISPLAY_RGBA8, with_passes=False):
        """Display resolve of the accumulation buffer -> numpy: uint8 [H, W, 4] (RGBA8) or float32 [H, W, 4];
        with_passes: (image, number of complete passes it shows)."""
        params = params if params is not None else display_params()
        dt, ch = (np.uint8, 4) if (fmt & 0xFF) == HR_DISPLAY_RGBA8 else (np.float32, 4)
        p = C.c_void_p()
        w, h = C.c_int32(), C.c_int32()
        shown = C.c_uint32()
        self._call("display_readback", C.byref(params), C.c_int32(fmt), C.byref(p), C.byref(w), C.byref(h), C.byref(shown))
        n = w.value * h.value * ch
        buf = (C.c_uint8 * n).from_address(p.value) if dt is np.uint8 else (C.c_float * n).from_address(p.value)
        img = np.frombuffer(buf, dtype=dt).reshape(h.value, w.value, ch).copy()
        return (img, int(shown.value)) if with_passes else img

    def passes_resolved(self):
        """Passes added to the accumulation buffer since the last clear, as enqueued so far (no synchronisation)."""
        n = C.c_uint64()
        self._call("frame_passes_resolved", C.byref(n))
        return int(n.value)

    # -- tile-shard exchange (device pointers for the HIP core, host pointers for the oracle)
    def packed_slots(self, rank, world):
        n = C.c_uint64()
        self._call("frame_packed_slots", C.c_int32(rank), C.c_int32(world), C.byref(n))
        return int(n.value)

    def pack_owned(self, out_ptr, stream=None):
        self._call("frame_pack_owned", C.c_void_p(int(out_ptr)), C.c_void_p(stream or 0))

    def unpack(self, src_rank, world, packed_ptr, full_ptr, stream=None):
        self._call("frame_unpack", C.c_int32(src_rank), C.c_int32(world), C.c_void_p(int(packed_ptr)), C.c_void_p(int(full_ptr)),
                   C.c_void_p(stream or 0))

    def display_device(self, device_ptr, params=None, fmt=HR_DISPLAY_RGBA8):
        """Asynchronous display resolve into device memory (e.g. a torch tensor or a GL-interop buffer)."""
        params = params if params is not None else display_params()
        self._call("display", C.byref(params), C.c_int32(fmt), C.c_void_p(int(device_ptr)), None)

    def readback_progressive(self, copy=True):
        """(buffer, complete passes in it) without completing the passes still in the pipeline.  copy=False returns a view
        of the context's pinned buffer (valid until the next readback)."""
        p = f32p()
        w, h, n = C.c_int32(), C.c_int32(), C.c_uint32()
        self._call("readback_progressive", C.byref(p), C.byref(w), C.byref(h), C.byref(n))
        a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4))
        return (a.copy() if copy else a), int(n.value)

    def debug_trace(self, origins, dirs, tmax=None, skip_prim=None, any_hit=False):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        tm = None if tmax is None else _f32(tmax).reshape(-1)
        sk = None if skip_prim is None else np.ascontiguousarray(skip_prim, dtype=np.int32).reshape(-1)
        out = np.empty(n, dtype=HIT_DTYPE)
        self._call("debug_trace", C.c_int32(n), _ptr(o), _ptr(d), _ptr(tm), _ptr(sk, i32p), C.c_int32(int(any_hit)),
                   out.ctypes.data_as(C.POINTER(Hit)))
        return out
