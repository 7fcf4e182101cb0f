"""Known-answer tests of the oracle's surface inputs (physicallyBased.rlsl:55-205, accumulator.rlsl:12-28): textures, vertex colours,
normal maps, alpha masks, single-sided faces, emission and the per-channel clamp — each against a closed-form image of a scene in which
only the piece under test differs from a Lambert plane under one directional light, L = Cdiff / pi * N.L * E (max_ray_depth 0: direct
lighting only, no sampling noise).  The HIP path is held to the oracle bit for bit elsewhere; these pin the oracle to the shader text."""
import math

import numpy as np

import oracle_lib
from heatray_amd import _ffi as ffi
from heatray_amd import host, scenes

F = np.float32
LIGHT = dict(color=(1.0, 0.9, 0.8), illuminance=683.0 * 2.0, phi=0.4, theta=0.9)


def _quad_mesh(y=0.0, s=50.0, up=True, **kw):
    """A square in the plane y = const, uv (0..1) along +x and -z; `up`: front face (CCW) seen from above."""
    pts = [(-s, y, s), (s, y, s), (s, y, -s), (-s, y, -s)]
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    if not up:
        pts, uv = pts[::-1], uv[::-1].copy()
    p, n, i = scenes._quad(*pts)
    return scenes.MeshData(p, n, i, uvs=uv, **kw)


def _scene(w=64, h=40, depth=0, lit=True):
    sc = scenes.Scene("kat_surface", width=w, height=h, use_multiscatter_lut=False)
    if lit:
        sc.lights.add_directional(**LIGHT)
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop = depth, w / h, host.FSTOP_DISABLED
    o.view_matrix = host.orbit_view_matrix(5.0, 0.0, 0.9)  # on +z, looking down at the origin: image x = world x
    return sc


def _render(sc, passes=2):
    eng = oracle_lib.engine()
    sc.apply(eng)
    for s in range(passes):
        eng.render_pass(sc.options.pass_params(s))
    img = eng.readback()
    return img[..., :3] / img[..., 3:4]


def _lambert(cdiff, normal=(0.0, 1.0, 0.0)):
    to_light = host.light_direction_to(LIGHT["phi"], LIGHT["theta"]).astype(np.float64)
    ndl = max(0.0, float(np.dot(np.asarray(normal, dtype=np.float64), to_light)))
    return np.asarray(cdiff, dtype=np.float64) / math.pi * ndl * np.array(LIGHT["color"]) * (LIGHT["illuminance"] / 683.0)


def test_emission_and_the_per_channel_clamp():
    # :205 performAccumulate(weight * emissive); accumulator.rlsl:12-28 value = min(vec3(maxChannelValue), colour): per channel
    sc = _scene(lit=False)
    sc.materials[0] = host.bake_pbr(base_color=(0, 0, 0), emissive_color=(0.3, 0.7, 1.0), roughness=1.0, specular_f0=0.0)
    sc.meshes.append(_quad_mesh(material_id=0))
    sc.options.max_channel_value = 0.5
    rgb = _render(sc)
    assert np.array_equal(rgb, np.broadcast_to(np.array([0.3, 0.5, 0.5], dtype=F), rgb.shape))
    sc.options.max_channel_value = math.pi
    assert np.array_equal(_render(sc), np.broadcast_to(np.array([0.3, 0.7, 1.0], dtype=F), rgb.shape))


def test_base_colour_texture_times_vertex_colour():
    # :57-69 baseColor = Material.baseColor * texture.rgb * vertexColor (HAS_BASE_COLOR_TEXTURE, VERTEX_COLORS)
    sc = _scene()
    base, vcol = np.array([0.8, 0.6, 0.9]), np.array([0.5, 1.0, 0.25])
    tex = np.array([[[0.5, 0.25, 1.0, 1.0], [1.0, 0.5, 0.25, 1.0]]], dtype=F)          # 2 x 1 texels: left / right half of the square
    sc.textures.append((tex, ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=base, roughness=1.0, specular_f0=0.0, vertex_colors=True, base_color_texture=0)
    m = _quad_mesh(material_id=0)
    m.colors = np.tile(vcol.astype(F), (4, 1))
    sc.meshes.append(m)
    rgb = _render(sc)
    h, w = rgb.shape[:2]
    for col, texel in ((w // 4, tex[0, 0, :3]), (3 * w // 4, tex[0, 1, :3])):
        assert np.allclose(rgb[h // 2, col], _lambert(base * texel * vcol), rtol=3e-5), (col, rgb[h // 2, col])


def test_metallic_roughness_texture_channels():
    # :134-139 metallicRoughness = texture(...).bg: metallic *= BLUE, roughness *= GREEN, red is ignored
    def image(tex_rgb):
        sc = _scene()
        if tex_rgb is not None:
            sc.textures.append((np.array([[tex_rgb]], dtype=F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
        sc.materials[0] = host.bake_pbr(base_color=(0.7, 0.5, 0.3), roughness=1.0, metallic=1.0, specular_f0=0.0,
                                        metallic_roughness_texture=0 if tex_rgb is not None else -1)
        sc.meshes.append(_quad_mesh(material_id=0))
        return _render(sc)
    want = _lambert((0.7, 0.5, 0.3))
    got = image((0.123, 1.0, 0.0))                 # blue 0: a dielectric, Cdiff = base, Cspec = mix(specularF0 = 0, base, 0) = 0
    assert np.allclose(got, want, rtol=3e-5), got[0, 0]
    metal = image(None)                            # metallic 1 stays: Cdiff = 0, everything comes from the GGX lobe
    assert not np.allclose(metal, want, rtol=0.05)
    assert np.allclose(image((0.9, 1.0, 1.0)), metal, rtol=1e-6)   # blue 1, green 1: the material's own values


def test_normal_map_tilts_the_shading_normal():
    # :112-118 N = normalize(mat3(T, B, N) * (texture.xyz * 2 - 1)) (HAS_NORMALMAP with the mesh's tangent space)
    alpha = 0.3
    nts = np.array([math.sin(alpha), 0.0, math.cos(alpha)])
    def image(with_map):
        sc = _scene()
        if with_map:
            sc.textures.append((np.array([[(nts + 1.0) / 2.0]], dtype=F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
        sc.materials[0] = host.bake_pbr(base_color=(0.6, 0.6, 0.6), roughness=1.0, specular_f0=0.0, normalmap=0 if with_map else -1)
        m = _quad_mesh(material_id=0)
        m.tangents = np.tile(np.array([1, 0, 0], dtype=F), (4, 1))
        m.bitangents = np.tile(np.array([0, 0, -1], dtype=F), (4, 1))   # T x B = +y = N
        sc.meshes.append(m)
        return _render(sc)
    tilted = np.array([math.sin(alpha), math.cos(alpha), 0.0])          # T sin a + N cos a
    assert np.allclose(image(True), _lambert((0.6, 0.6, 0.6), tilted), rtol=3e-5)
    assert np.allclose(image(False), _lambert((0.6, 0.6, 0.6)), rtol=3e-5)
    assert abs(_lambert((1, 1, 1), tilted)[0] / _lambert((1, 1, 1))[0] - 1.0) > 0.05     # (the map does change the answer)


def test_alpha_mask_lets_camera_and_shadow_rays_through_its_holes():
    # :70-91 a hit with alpha < 1 re-emits the ray (an occlusion ray goes on towards the light); Mesh.cpp:95-100 such primitives are
    # not occluders, so shadow rays run this shader.  A sheet above a floor: its left half (u < 0.5) is holes, its right half opaque.
    floor, sheet = np.array([0.5, 0.6, 0.7]), np.array([0.9, 0.2, 0.2])
    sc = _scene(w=96, h=48)
    tex = np.array([[[1, 1, 1, 0.0], [1, 1, 1, 1.0]]], dtype=F)
    sc.textures.append((tex, ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=floor, roughness=1.0, specular_f0=0.0)
    sc.materials[1] = host.bake_pbr(base_color=sheet, roughness=1.0, specular_f0=0.0, alpha_mask=True, base_color_texture=0)
    sc.meshes.append(_quad_mesh(y=0.0, material_id=0))
    sc.meshes.append(_quad_mesh(y=1.5, s=1.0, material_id=1, is_occluder=False))
    rgb = _render(sc)
    h, w = rgb.shape[:2]
    to_light = host.light_direction_to(LIGHT["phi"], LIGHT["theta"]).astype(np.float64)
    is_floor = np.isclose(rgb, _lambert(floor), rtol=3e-5).all(axis=-1)
    is_sheet = np.isclose(rgb, _lambert(sheet), rtol=3e-5).all(axis=-1)
    is_dark = (rgb == 0).all(axis=-1)
    # every pixel is the lit floor (also seen THROUGH the holes), the opaque half's own Lambert term, or the floor in that half's shadow
    # (edge pixels mix two of them)
    assert (is_floor | is_sheet | is_dark).mean() > 0.95
    assert is_sheet.sum() > 30 and is_floor.sum() > 1000
    # the shadow of the opaque half lies on the floor beside it (1.5 above the floor, the light is oblique): black pixels ...
    assert is_dark.sum() > 20
    # ... fewer than the same sheet without holes leaves (its other half would cast a shadow and hide floor too)
    sc.textures[0] = (np.array([[[1, 1, 1, 1.0], [1, 1, 1, 1.0]]], dtype=F), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST)
    opaque = _render(sc)
    assert (opaque == 0).all(axis=-1).sum() > 1.3 * is_dark.sum()
    assert np.isclose(opaque, _lambert(sheet), rtol=3e-5).all(axis=-1).sum() > 1.7 * is_sheet.sum()
    sc.textures[0] = (np.array([[[1, 1, 1, 0.0], [1, 1, 1, 0.0]]], dtype=F), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST)
    all_holes = _render(sc)
    assert np.allclose(all_holes, _lambert(floor), rtol=3e-5)      # an all-hole sheet is not there at all
    assert to_light[1] > 0.3


def test_single_sided_back_faces_are_invisible_but_still_cast_shadows():
    # :99-108 (no DOUBLE_SIDED): a back-face hit re-emits the ray — the camera sees through; but the primitive is an occluder
    # (only alpha-masked ones are not, Mesh.cpp:95-100), so OpenRL's occlusion test stops at it without running any shader
    floor = np.array([0.5, 0.6, 0.7])
    def image(with_sheet, double_sided=False):
        sc = _scene(w=96, h=48)
        sc.materials[0] = host.bake_pbr(base_color=floor, roughness=1.0, specular_f0=0.0)
        sc.materials[1] = host.bake_pbr(base_color=(0.9, 0.1, 0.1), roughness=1.0, specular_f0=0.0, double_sided=double_sided)
        sc.meshes.append(_quad_mesh(y=0.0, material_id=0))
        if with_sheet:
            sc.meshes.append(_quad_mesh(y=1.5, s=1.0, up=False, material_id=1))   # front face points DOWN: the camera above sees its back
        return _render(sc)
    bare, seen_from_behind, two_sided = image(False), image(True), image(True, double_sided=True)
    assert np.allclose(bare, _lambert(floor), rtol=3e-5)
    lit = np.isclose(seen_from_behind, _lambert(floor), rtol=3e-5).all(axis=-1)
    dark = (seen_from_behind == 0).all(axis=-1)
    assert (lit | dark).mean() > 0.97                 # nothing of the sheet's own colour anywhere (edge pixels mix lit and dark samples)
    assert dark.sum() > 20 and lit.sum() > 20         # ... yet its shadow lies on the floor
    red = two_sided[..., 0] > 2.0 * two_sided[..., 2] + 1e-6
    assert red.sum() > 20                             # the double-sided twin of the same sheet IS visible (normal flipped, :95-98)


def test_thin_lens_depth_of_field_blur_disc():
    # perspective.rlsl:72-91 — the ray starts at the aperture sample (ApertureSamples * 2 - 1) * apertureRadius in the lens plane and
    # aims at focusDistance * dirCS, apertureRadius = focalLength / fstop / 1000 (PassGenerator.h:92-94), fovTan = 12 / focalLength
    # for the 36 x 24 mm film (PassGenerator.cpp:341-343).  A straight emissive edge at distance z then images as the edge-spread of
    # a uniform disc of radius rho = R |1 - z / F| in the object plane: E(u) = 1/2 + (asin u + u sqrt(1 - u^2)) / pi, u = x / rho.
    R_mm, F_dist, focal = 100.0, 4.0, 50.0
    def profile(z_obj, passes=256, w=192, h=4):
        sc = scenes.Scene("dof", width=w, height=h, use_multiscatter_lut=False)
        sc.materials[0] = host.bake_pbr(base_color=(0, 0, 0), emissive_color=(1, 1, 1), roughness=1.0, specular_f0=0.0)
        zc = 10.0 - z_obj
        p, n, i = scenes._quad((-50, -50, zc), (0, -50, zc), (0, 50, zc), (-50, 50, zc))     # covers x < 0, faces +z (the camera)
        sc.meshes.append(scenes.MeshData(p, n, i, material_id=0))
        o = sc.options
        o.max_ray_depth, o.aspect_ratio, o.max_render_passes = 0, 1.0, passes            # (aspect 1: the 192 pixels of a row span +-fovTan)
        o.focal_length, o.fstop, o.focus_distance = focal, focal / R_mm, F_dist          # aperture radius = 50 / 0.5 / 1000 = 0.1
        o.view_matrix = host.orbit_view_matrix(10.0, 0.0, 0.0)                           # camera at (0, 0, 10), looking down -z
        eng = oracle_lib.engine()
        sc.apply(eng)
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
        img = eng.readback()
        v = (img[..., 0] / img[..., 3]).mean(axis=0)
        fov_tan = 12.0 / focal
        x0 = ((np.arange(w) + 0.5) / w * 2.0 - 1.0) * fov_tan * z_obj                       # where a pixel's pinhole ray meets the plane
        return x0, v
    def width_10_90(x0, v):   # v falls from 1 (x < 0) to 0
        xs = [np.interp(-level, -v, x0) for level in (0.9, 0.1)]
        return xs[1] - xs[0]
    us = np.linspace(-1, 1, 20001)
    esf = 0.5 + (np.arcsin(us) + us * np.sqrt(1 - us * us)) / math.pi
    u90 = np.interp(0.9, esf, us)
    R = focal / (focal / R_mm) / 1000.0
    assert abs(R - 0.1) < 1e-9
    for z_obj in (8.0, 2.0, 6.0):
        rho = R * abs(1.0 - z_obj / F_dist)
        x0, v = profile(z_obj)
        assert v[:10].min() > 0.999 and v[-10:].max() < 1e-3
        got, want = width_10_90(x0, v), 2.0 * u90 * rho
        assert abs(got - want) < 0.08 * want + 0.5 * (x0[1] - x0[0]), (z_obj, got, want)
        # the whole profile follows the disc's edge spread (pixel footprint and sample count blur it a little)
        inside = np.abs(x0) < 0.8 * rho
        assert np.abs(v[inside] - np.interp(-x0[inside] / rho, us, esf)).max() < 0.06, z_obj
    x0, v = profile(F_dist)                                  # in focus: the edge is as sharp as the pixel grid allows
    assert width_10_90(x0, v) < 1.5 * (x0[1] - x0[0])


def test_emissive_texture_replaces_the_emissive_colour():
    # :156-159 emissive = texture2D(Material.emissiveTexture, texCoord).rgb (replaces, does not multiply)
    sc = _scene(lit=False)
    tex = np.array([[[0.2, 0.4, 0.6], [0.9, 0.1, 0.3]]], dtype=F)
    sc.textures.append((tex, ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=(0, 0, 0), emissive_color=(1.0, 1.0, 1.0), roughness=1.0, specular_f0=0.0, emissive_texture=0)
    sc.meshes.append(_quad_mesh(material_id=0))
    rgb = _render(sc)
    h, w = rgb.shape[:2]
    assert np.array_equal(rgb[h // 2, w // 4], tex[0, 0]) and np.array_equal(rgb[h // 2, 3 * w // 4], tex[0, 1])


def test_material_baking_clamps():
    # PhysicallyBasedMaterial.cpp:133-145: roughness >= 0.01, specularF0 * 0.08, clearCoat * 0.2, alphas = roughness^2;
    # GlassMaterial.cpp:88-126 keeps the same roughness floor
    m = host.bake_pbr(roughness=0.0, specular_f0=1.0, clear_coat=1.0, clear_coat_roughness=0.0, metallic=2.0, base_color=(2, -1, 0.5))
    assert m.roughness == F(0.01) and m.roughness_alpha == F(0.01) * F(0.01)
    assert m.specular_f0 == F(0.08) and m.clear_coat == F(0.2)
    assert m.clear_coat_roughness == F(0.01) and m.clear_coat_roughness_alpha == F(0.01) * F(0.01)
    assert m.metallic == 1.0 and list(m.base_color) == [1.0, 0.0, 0.5]
    g = host.bake_glass(base_color=(0.5, 0.5, 0.5), roughness=0.0, ior=1.5, density=0.5)
    assert g.roughness == F(0.01)


# ---- the scenes above, for the GPU suite: the HIP path must give the oracle's image of each, bit for bit
def gpu_parity_scenes():
    out = []
    sc = _scene(lit=False)
    sc.materials[0] = host.bake_pbr(base_color=(0, 0, 0), emissive_color=(0.3, 0.7, 1.0), roughness=1.0, specular_f0=0.0)
    sc.meshes.append(_quad_mesh(material_id=0))
    sc.options.max_channel_value = 0.5
    out.append(("emission and clamp", sc))
    sc = _scene(depth=3)
    sc.textures.append((np.array([[[0.5, 0.25, 1.0, 1.0], [1.0, 0.5, 0.25, 1.0]]], dtype=F), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.textures.append((np.array([[(0.123, 0.8, 0.3)]], dtype=F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
    sc.textures.append((np.array([[((math.sin(0.3) + 1) / 2, 0.5, (math.cos(0.3) + 1) / 2)]], dtype=F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
    sc.textures.append((np.array([[[0.2, 0.4, 0.6], [0.9, 0.1, 0.3]]], dtype=F), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=(0.8, 0.6, 0.9), roughness=0.7, metallic=0.9, specular_f0=0.5, vertex_colors=True, base_color_texture=0,
                                    metallic_roughness_texture=1, normalmap=2, emissive_texture=3)
    m = _quad_mesh(material_id=0)
    m.colors = np.tile(np.array([0.5, 1.0, 0.25], dtype=F), (4, 1))
    m.tangents = np.tile(np.array([1, 0, 0], dtype=F), (4, 1))
    m.bitangents = np.tile(np.array([0, 0, -1], dtype=F), (4, 1))
    sc.meshes.append(m)
    out.append(("base colour, metallic-roughness, normal and emissive textures with vertex colours", sc))
    sc = _scene(w=96, h=48, depth=2)
    sc.textures.append((np.array([[[1, 1, 1, 0.0], [1, 1, 1, 1.0]]], dtype=F), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=(0.5, 0.6, 0.7), roughness=1.0, specular_f0=0.0)
    sc.materials[1] = host.bake_pbr(base_color=(0.9, 0.2, 0.2), roughness=1.0, specular_f0=0.0, alpha_mask=True, base_color_texture=0)
    sc.materials[2] = host.bake_pbr(base_color=(0.9, 0.1, 0.1), roughness=1.0, specular_f0=0.0, double_sided=False)
    sc.meshes.append(_quad_mesh(y=0.0, material_id=0))
    sc.meshes.append(_quad_mesh(y=1.5, s=1.0, material_id=1, is_occluder=False))
    mesh = _quad_mesh(y=0.8, s=0.7, up=False, material_id=2)
    mesh.positions = (mesh.positions + np.array([2.5, 0, 0.5], dtype=F)).astype(F)
    sc.meshes.append(mesh)
    out.append(("alpha-masked sheet and a single-sided sheet seen from behind, over a floor", sc))
    return out
