"""Host-side parameter baking: the arithmetic the reference does on the CPU before
anything reaches the ray engine.  The drop-in C++ layer (heatray_amd/host/*.h) does
the same in C++; this module is its Python twin for bench.py and the tests.

All paths below are relative to /root/reference/Source/HeatrayRenderer.
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _ffi as ffi

F = np.float32

WATTS_TO_LUMENS = F(683.0)
LUMENS_TO_WATTS = F(1.0) / F(683.0)  # Lights/DirectionalLight.cpp:16


def _sat(x):
    return np.minimum(np.maximum(np.asarray(x, dtype=F), F(0)), F(1))


def bake_pbr(base_color=(1, 1, 1), emissive_color=(0, 0, 0), roughness=1.0, metallic=0.0, specular_f0=0.5,
             clear_coat=0.0, clear_coat_roughness=0.0, double_sided=True, alpha_mask=False, vertex_colors=False,
             base_color_texture=-1, metallic_roughness_texture=-1, emissive_texture=-1, normalmap=-1,
             clear_coat_texture=-1, clear_coat_roughness_texture=-1, clear_coat_normalmap=-1, multiscatter_lut=-1,
             force_enable_all_textures=False):
    """PhysicallyBasedMaterial::modify + the shader permutation flags of ::build
    (Materials/PhysicallyBasedMaterial.cpp:127-191, 57-110; defaults from .h:22-41)."""
    m = ffi.Material()
    m.type = ffi.HR_MAT_PBR
    k_min_roughness, k_max_f0, k_max_cc = F(0.01), F(0.08), F(0.2)
    m.base_color = (C.c_float * 3)(*_sat(base_color))
    m.emissive_color = (C.c_float * 3)(*_sat(emissive_color))
    m.metallic = float(_sat(metallic))
    r = np.maximum(_sat(roughness), k_min_roughness)
    m.roughness = float(r)
    m.specular_f0 = float(F(specular_f0) * k_max_f0)
    m.roughness_alpha = float(F(r) * F(r))
    m.clear_coat = float(F(clear_coat) * k_max_cc)
    ccr = np.maximum(_sat(clear_coat_roughness), k_min_roughness)
    m.clear_coat_roughness = float(ccr)
    m.clear_coat_roughness_alpha = float(F(ccr) * F(ccr))
    flags = 0
    fa = force_enable_all_textures
    if base_color_texture >= 0 or fa:
        flags |= ffi.HR_MF_HAS_BASE_COLOR_TEXTURE
    if metallic_roughness_texture >= 0 or fa:
        flags |= ffi.HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE
    if emissive_texture >= 0:
        flags |= ffi.HR_MF_HAS_EMISSIVE_TEXTURE
    if normalmap >= 0:
        flags |= ffi.HR_MF_HAS_NORMALMAP
    if clear_coat_texture >= 0 or fa:
        flags |= ffi.HR_MF_HAS_CLEARCOAT_TEXTURE
    if clear_coat_roughness_texture >= 0 or fa:
        flags |= ffi.HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE
    if clear_coat_normalmap >= 0:
        flags |= ffi.HR_MF_HAS_CLEARCOAT_NORMALMAP
    if double_sided:
        flags |= ffi.HR_MF_DOUBLE_SIDED
    if alpha_mask:
        flags |= ffi.HR_MF_ALPHA_MASK
    if vertex_colors:
        flags |= ffi.HR_MF_VERTEX_COLORS
    m.flags = flags
    (m.base_color_texture, m.metallic_roughness_texture, m.emissive_texture, m.normalmap, m.clear_coat_texture,
     m.clear_coat_roughness_texture, m.clear_coat_normalmap, m.multiscatter_lut) = (
        base_color_texture, metallic_roughness_texture, emissive_texture, normalmap, clear_coat_texture,
        clear_coat_roughness_texture, clear_coat_normalmap, multiscatter_lut)
    return m


def bake_glass(base_color=(1, 1, 1), roughness=1.0, ior=1.57, density=0.05, base_color_texture=-1, normalmap=-1,
               metallic_roughness_texture=-1, vertex_colors=False, force_enable_all_textures=False):
    """GlassMaterial::modify + ::build flags (Materials/GlassMaterial.cpp:88-126, 50-77; defaults .h:21-31)."""
    m = ffi.Material()
    m.type = ffi.HR_MAT_GLASS
    m.base_color = (C.c_float * 3)(*_sat(base_color))
    r = np.maximum(_sat(roughness), F(0.01))
    m.roughness = float(r)
    m.roughness_alpha = float(F(r) * F(r))
    m.density = float(F(density))
    i = np.maximum(F(0), F(ior))
    m.ior = float(i)
    f0 = np.abs((F(1) - i) / (F(1) + i))
    m.specular_f0 = float(F(f0) * F(f0))
    flags = 0
    if base_color_texture >= 0 or force_enable_all_textures:
        flags |= ffi.HR_MF_HAS_BASE_COLOR_TEXTURE
    if metallic_roughness_texture >= 0 or force_enable_all_textures:
        flags |= ffi.HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE
    if normalmap >= 0:
        flags |= ffi.HR_MF_HAS_NORMALMAP
    if vertex_colors:
        flags |= ffi.HR_MF_VERTEX_COLORS
    m.flags = flags
    m.base_color_texture, m.normalmap, m.metallic_roughness_texture = base_color_texture, normalmap, metallic_roughness_texture
    m.emissive_texture = m.clear_coat_texture = m.clear_coat_roughness_texture = m.clear_coat_normalmap = -1
    m.multiscatter_lut = -1
    return m


# ---- orientation helpers (glm::angleAxis / mat4_cast, as used by OrbitCamera.h:32-45,
# Lights/DirectionalLight.cpp:64-78, Lights/SpotLight.cpp:76-91) ----
def _quat_axis(angle, axis):
    s = math.sin(angle * 0.5)
    return np.array([math.cos(angle * 0.5), axis[0] * s, axis[1] * s, axis[2] * s])  # w, x, y, z


def _quat_mul(p, q):
    pw, px, py, pz = p
    qw, qx, qy, qz = q
    return np.array([pw * qw - px * qx - py * qy - pz * qz, pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy + py * qw + pz * qx - px * qz, pw * qz + pz * qw + px * qy - py * qx])


def _quat_to_mat4(q):
    w, x, y, z = q
    m = np.eye(4)
    m[0, 0], m[1, 0], m[2, 0] = 1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)
    m[0, 1], m[1, 1], m[2, 1] = 2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)
    m[0, 2], m[1, 2], m[2, 2] = 2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)
    return m  # m[row, col]


def _orientation_matrix(phi, theta):
    q = _quat_mul(_quat_axis(theta, (1, 0, 0)), _quat_axis(phi, (0, 1, 0)))
    inv = np.array([q[0], -q[1], -q[2], -q[3]]) / np.dot(q, q)
    return _quat_to_mat4(inv)


def orbit_view_matrix(distance, phi, theta, target=(0, 0, 0)):
    """OrbitCamera::createViewMatrix (OrbitCamera.h:32-45): camera -> world, returned as m[row, col]."""
    t = np.eye(4)
    t[:3, 3] = np.asarray(target, dtype=np.float64) + np.array([0, 0, distance])
    return (_orientation_matrix(phi, theta) @ t).astype(F)


def light_direction_to(phi, theta):
    """DirectionalLight::calculateDirection (Lights/DirectionalLight.cpp:64-78): direction TO the light."""
    d = _orientation_matrix(phi, theta)[:3, 2]
    return (d / np.linalg.norm(d)).astype(F)


@dataclass
class LightRig:
    """Collects lights and bakes the packed blocks of ShaderLightingDefines.h:33-64."""
    directional: list = field(default_factory=list)
    point: list = field(default_factory=list)
    spot: list = field(default_factory=list)
    env_texture: int = -1
    env_enabled: bool = False
    env_exposure_compensation: float = 0.0
    env_theta_rotation: float = 0.0

    def add_directional(self, color=(1, 1, 1), illuminance=683.0 * math.pi, phi=0.0, theta=math.pi / 2):
        # DirectionalLight::copyToLightBuffer (DirectionalLight.cpp:42-51)
        c = np.asarray(color, dtype=F) * (F(illuminance) * LUMENS_TO_WATTS)
        self.directional.append((light_direction_to(phi, theta), c))

    def add_point(self, position, color=(1, 1, 1), luminous_intensity=683.0 * 4 * math.pi):
        # PointLight::copyToLightBuffer (PointLight.cpp:41-50)
        watts = (F(luminous_intensity) * LUMENS_TO_WATTS) * (F(4.0) * F(math.pi))
        self.point.append((np.asarray(position, dtype=F), np.asarray(color, dtype=F) * watts))

    def add_spot(self, position, color=(1, 1, 1), luminous_intensity=683.0 * math.pi * math.pi, phi=0.0,
                 theta=math.pi / 2, inner_angle=0.0, outer_angle=math.radians(40.0)):
        # SpotLight::setParams + copyToLightBuffer (SpotLight.cpp:44-69); direction FROM the light (:76-91)
        if inner_angle > outer_angle:
            inner_angle = max(0.0, outer_angle - math.radians(1.0))
        if inner_angle > 0.0 and inner_angle == outer_angle:
            inner_angle -= math.radians(1.0)
        watts = (F(luminous_intensity) * LUMENS_TO_WATTS) * F(math.pi)
        d = -light_direction_to(phi, theta)
        self.spot.append((np.asarray(position, dtype=F), d, np.asarray(color, dtype=F) * watts,
                          np.array([math.cos(inner_angle), math.cos(outer_angle)], dtype=F)))

    def set_environment(self, texture_id, exposure_compensation=0.0, theta_rotation=0.0):
        self.env_texture, self.env_enabled = texture_id, True
        self.env_exposure_compensation, self.env_theta_rotation = exposure_compensation, theta_rotation

    def bake(self):
        L = ffi.Lights()
        L.n_directional = len(self.directional)
        for i, (d, c) in enumerate(self.directional):
            L.directional_directions[i] = (C.c_float * 3)(*d)
            L.directional_colors[i] = (C.c_float * 3)(*c)
        L.n_point = len(self.point)
        for i, (p, c) in enumerate(self.point):
            L.point_positions[i] = (C.c_float * 3)(*p)
            L.point_colors[i] = (C.c_float * 3)(*c)
        L.n_spot = len(self.spot)
        for i, (p, d, c, a) in enumerate(self.spot):
            L.spot_positions[i] = (C.c_float * 3)(*p)
            L.spot_directions[i] = (C.c_float * 3)(*d)
            L.spot_colors[i] = (C.c_float * 3)(*c)
            L.spot_angles[i] = (C.c_float * 2)(*a)
        L.env_enabled = int(self.env_enabled)
        L.env_texture = self.env_texture
        L.env_exposure = float(F(2.0) ** F(self.env_exposure_compensation))  # EnvironmentLight.cpp:95
        L.env_theta_rotation = self.env_theta_rotation
        return L


FSTOP_DISABLED = float(np.finfo(np.float32).max)  # PassGenerator.h:79-81


@dataclass
class RenderOptions:
    """The subset of PassGenerator::RenderOptions (PassGenerator.h:49-150) that reaches the ray engine."""
    max_render_passes: int = 32
    max_ray_depth: int = 10
    max_channel_value: float = float(F(math.pi))
    aspect_ratio: float = -1.0
    focus_distance: float = 1.0
    focal_length: float = 50.0
    fstop: float = 32.0
    view_matrix: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    sample_mode: int = ffi.HR_SAMPLE_SOBOL
    bokeh_shape: int = ffi.HR_BOKEH_CIRCULAR
    enable_interactive_mode: bool = False
    visualizer_mode: int = ffi.HR_VIS_NONE
    show_nans: bool = False
    show_inf: bool = False
    estimator: int = ffi.HR_ESTIMATOR_REFERENCE
    texture_lod: int = ffi.HR_TEXTURE_LOD_BASE

    def pass_params(self, sample_index, current_block_pixel=(0, 0)):
        """Uniforms of one pass: PassGenerator::runRenderFrameJob (PassGenerator.cpp:341-369)."""
        p = ffi.PassParams()
        p.sample_index = sample_index
        p.max_ray_depth = self.max_ray_depth
        p.max_channel_value = self.max_channel_value
        fov_y = F(2.0) * F(math.atan2(24.0, 2.0 * self.focal_length))  # 35 mm film, :341-343
        p.fov_tan = float(F(math.tan(float(fov_y * F(0.5)))))
        p.aspect_ratio = self.aspect_ratio
        p.focus_distance = self.focus_distance
        p.aperture_radius = float((F(self.focal_length) / F(self.fstop)) / F(1000.0))  # PassGenerator.h:92-94
        p.view_matrix = (C.c_float * 16)(*np.asarray(self.view_matrix, dtype=F).T.reshape(-1))
        p.interactive_mode = int(self.enable_interactive_mode)
        p.block_size = (C.c_int32 * 2)(3, 3)
        p.current_block_pixel = (C.c_int32 * 2)(*current_block_pixel)
        p.max_sample_index = float(self.max_render_passes)
        p.enable_visualizer = int(self.visualizer_mode != ffi.HR_VIS_NONE)
        p.visualizer_mode = self.visualizer_mode
        p.enable_accumulator_visualizer = int(self.show_nans or self.show_inf)
        p.show_nans, p.show_inf = int(self.show_nans), int(self.show_inf)
        p.estimator = int(self.estimator)
        p.texture_lod = int(self.texture_lod)
        return p
