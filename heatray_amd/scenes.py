"""Synthetic benchmark / test scenes.

Everything is generated from a seeded splitmix64 stream (seed 0x48454154, "HEAT") so the
CPU oracle and the HIP core are fed bit-identical inputs.  A `Scene` is plain data; `apply`
pushes it through an `Engine` (either library), exactly the calls the C++ Mesh / Material /
Lighting classes make.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import _ffi as ffi
from . import host

SEED = 0x48454154
F = np.float32
_M64 = (1 << 64) - 1


class SplitMix64:
    """splitmix64 -> uniform float32 in [0,1) from the top 24 bits."""

    def __init__(self, seed=SEED):
        self.state = np.uint64(seed)

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = z[-1] if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, shape, lo=0.0, hi=1.0):
        n = int(np.prod(shape))
        u = (self.u64(n) >> np.uint64(40)).astype(np.float32) * F(1.0 / (1 << 24))
        return (F(lo) + u * F(hi - lo)).astype(F).reshape(shape)


@dataclass
class MeshData:
    positions: np.ndarray
    normals: np.ndarray
    indices: np.ndarray
    uvs: np.ndarray = None
    tangents: np.ndarray = None
    bitangents: np.ndarray = None
    colors: np.ndarray = None
    mode: int = ffi.HR_TRIANGLES
    world: np.ndarray = None
    material_id: int = 0
    is_occluder: bool = True


@dataclass
class Scene:
    name: str
    meshes: list = field(default_factory=list)
    materials: dict = field(default_factory=dict)  # id -> ffi.Material (baked)
    textures: list = field(default_factory=list)   # (pixels, wrap, filter); ids are list positions + tex_base
    lights: host.LightRig = field(default_factory=host.LightRig)
    env_pixels: np.ndarray = None                   # lat/long RGB32F, row 0 = bottom
    env_exposure_compensation: float = 0.0
    use_multiscatter_lut: bool = True
    options: host.RenderOptions = field(default_factory=host.RenderOptions)
    width: int = 256
    height: int = 256

    @property
    def n_triangles(self):
        n = 0
        for m in self.meshes:
            n += (m.indices.size - 2) if m.mode == ffi.HR_TRIANGLE_STRIP else m.indices.size // 3
        return n

    def apply(self, eng, lut=None, tables=None):
        """Upload the scene.  `lut`: 128x128 multiscatter LUT (host array) or None to let the engine
        generate it; `tables`: (seq, aperture, offsets) host arrays or None to let the engine
        generate them (device-side on the HIP core)."""
        eng.resize(self.width, self.height)
        lut_id = -1
        if self.use_multiscatter_lut:
            if lut is not None:
                lut_id = eng.create_texture(np.asarray(lut, dtype=F), wrap=ffi.HR_WRAP_CLAMP_TO_EDGE)
            else:
                _, lut_id = eng.generate_multiscatter_lut(want_host=False)
        tex_ids = [eng.create_texture(px, wrap=w, filter=f) for (px, w, f) in self.textures]
        for mid, mat in self.materials.items():
            m = ffi.Material.from_buffer_copy(mat)
            for fld in ("base_color_texture", "metallic_roughness_texture", "emissive_texture", "normalmap",
                        "clear_coat_texture", "clear_coat_roughness_texture", "clear_coat_normalmap"):
                t = getattr(m, fld)
                if t >= 0:
                    setattr(m, fld, tex_ids[t])
            if m.type == ffi.HR_MAT_PBR:
                m.multiscatter_lut = lut_id
            eng.set_material(mid, m)
        for me in self.meshes:
            eng.add_mesh(me.positions, me.normals, me.indices, uvs=me.uvs, tangents=me.tangents,
                         bitangents=me.bitangents, colors=me.colors, mode=me.mode, world=me.world,
                         is_occluder=me.is_occluder, material_id=me.material_id)
        eng.commit()
        rig = self.lights
        if self.env_pixels is not None:
            env_id = eng.create_texture(self.env_pixels, wrap=ffi.HR_WRAP_REPEAT)
            rig.set_environment(env_id, self.env_exposure_compensation, rig.env_theta_rotation)
        eng.set_lights(rig.bake())
        if tables is not None:
            seq, ap, off = tables
            eng.set_sequences(seq, ap)
            eng.set_seq_offsets(off)
        else:  # every sample mode and bokeh shape has a generator behind the C-ABI (PassGenerator.cpp:603-684)
            eng.generate_sequences(self.options.sample_mode, self.options.bokeh_shape, self.options.max_render_passes)
            eng.generate_seq_offsets()
        eng.clear()


def _quad(p0, p1, p2, p3):
    """Two CCW triangles (p0,p1,p2),(p0,p2,p3) with the flat normal of the winding."""
    p = np.array([p0, p1, p2, p3], dtype=F)
    n = np.cross(p[1] - p[0], p[2] - p[0])
    n = (n / np.linalg.norm(n)).astype(F)
    return p, np.tile(n, (4, 1)), np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)


def _merge(parts):
    pos, nrm, idx, base = [], [], [], 0
    for p, n, i in parts:
        pos.append(p), nrm.append(n), idx.append(i + base)
        base += p.shape[0]
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(idx).astype(np.uint32)


def _box(cx, cz, sx, sy, sz, angle):
    """Five-sided... no: full six-sided box minus the bottom = 5 quads = 10 triangles."""
    c, s = math.cos(angle), math.sin(angle)

    def P(x, y, z):
        return (cx + c * x * sx + s * z * sz, y * sy, cz - s * x * sx + c * z * sz)

    v = [P(-1, 0, 1), P(1, 0, 1), P(1, 0, -1), P(-1, 0, -1), P(-1, 1, 1), P(1, 1, 1), P(1, 1, -1), P(-1, 1, -1)]
    quads = [(4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
    return [_quad(*(v[i] for i in q)) for q in quads]


def cornell_box(width=256, height=256, bounces=4, passes=32):
    """C1: 32 triangles = 5 walls x 2 + light quad x 2 + 2 boxes x 10; white/red/green Lambert
    (roughness 1, specularF0 0), emissive ceiling quad, no environment."""
    sc = Scene("cornell", width=width, height=height, use_multiscatter_lut=True)
    lam = dict(roughness=1.0, metallic=0.0, specular_f0=0.0)
    sc.materials = {0: host.bake_pbr(base_color=(0.73, 0.73, 0.73), **lam),
                    1: host.bake_pbr(base_color=(0.65, 0.05, 0.05), **lam),
                    2: host.bake_pbr(base_color=(0.12, 0.45, 0.15), **lam),
                    3: host.bake_pbr(base_color=(0.0, 0.0, 0.0), emissive_color=(1, 1, 1), **lam)}
    white = [_quad((-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)),      # floor (normal +y)
             _quad((-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)),      # ceiling (normal -y)
             _quad((-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1))]    # back wall (normal +z)
    white += _box(-0.35, -0.3, 0.3, 1.2, 0.3, 0.3) + _box(0.4, 0.35, 0.3, 0.6, 0.3, -0.3)
    p, n, i = _merge(white)
    sc.meshes.append(MeshData(p, n, i, material_id=0))
    p, n, i = _quad((-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1))   # left wall, red (normal +x)
    sc.meshes.append(MeshData(p, n, i, material_id=1))
    p, n, i = _quad((1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1))       # right wall, green (normal -x)
    sc.meshes.append(MeshData(p, n, i, material_id=2))
    p, n, i = _quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3))  # light, faces down
    sc.meshes.append(MeshData(p, n, i, material_id=3))
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.aspect_ratio = width / height
    o.focal_length = 35.0
    o.fstop = host.FSTOP_DISABLED
    o.view_matrix = host.orbit_view_matrix(3.4, 0.0, 0.0, target=(0, 1, 0))
    o.focus_distance = 3.4
    return sc


def _material_palette(rng, n=16, glass_fraction=0.0, clearcoat_fraction=0.0, passthrough_fraction=0.0):
    """16 PBR rows: baseColor ~U(0.2,0.9)^3, roughness ~U(0.05,1), metallic in {0,1} p=0.25;
    C5 swaps in 25 % glass (ior 1.5, density 0.5, roughness 0.05) and 25 % clearcoat 1 / roughness 0.1."""
    mats = {}
    base = rng.uniform((n, 3), 0.2, 0.9)
    rough = rng.uniform((n,), 0.05, 1.0)
    metal = (rng.uniform((n,)) < 0.25).astype(F)
    n_glass, n_cc = int(round(n * glass_fraction)), int(round(n * clearcoat_fraction))
    for i in range(n):
        if i < n_glass:
            mats[i] = host.bake_glass(base_color=base[i], roughness=0.05, ior=1.5, density=0.5)
        elif i < n_glass + n_cc:
            mats[i] = host.bake_pbr(base_color=base[i], roughness=float(rough[i]), metallic=float(metal[i]),
                                    clear_coat=1.0, clear_coat_roughness=0.1)
        elif i >= n - int(round(n * passthrough_fraction)):
            # glTF-style assets: single-sided surfaces (AI_MATKEY_TWOSIDED defaults to false) and alpha-masked cut-outs, whose
            # back faces / holes let rays pass through (physicallyBased.rlsl:70-108); texture 0 is the scene's alpha mask
            masked = (i % 2) == 0
            mats[i] = host.bake_pbr(base_color=base[i], roughness=float(rough[i]), metallic=float(metal[i]), double_sided=False,
                                    alpha_mask=masked, base_color_texture=0 if masked else -1)
        else:
            mats[i] = host.bake_pbr(base_color=base[i], roughness=float(rough[i]), metallic=float(metal[i]))
    return mats


def synthetic_hdri(width=2048, height=1024):
    """Sky gradient + three Gaussian suns, peak 50; RGB32F, row 0 = bottom (v = 0)."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    el = (v - 0.5) * math.pi          # elevation
    az = (u - 0.5) * 2.0 * math.pi    # azimuth
    t = np.clip(v, 0, 1)[:, None]
    sky = np.stack([0.15 + 0.35 * t, 0.2 + 0.5 * t, 0.3 + 0.7 * t], axis=-1) * np.ones((1, width, 1))
    ground = np.array([0.12, 0.1, 0.08])
    img = np.where(t[..., None] < 0.5, ground * (0.5 + t[..., None]), sky)
    d = np.stack([np.cos(el)[:, None] * np.sin(az)[None, :], np.sin(el)[:, None] * np.ones((1, width)),
                  -np.cos(el)[:, None] * np.cos(az)[None, :]], axis=-1)
    for (saz, sel, sigma, col) in [(0.6, 0.9, 0.05, (50, 45, 38)), (-1.8, 0.4, 0.08, (20, 24, 30)),
                                   (2.6, 0.2, 0.12, (12, 8, 5))]:
        sd = np.array([math.cos(sel) * math.sin(saz), math.sin(sel), -math.cos(sel) * math.cos(saz)])
        ang = np.arccos(np.clip(d @ sd, -1, 1))
        img = img + np.exp(-0.5 * (ang / sigma) ** 2)[..., None] * np.array(col)
    return np.minimum(img, 50.0).astype(F)


def _camera_for(sc, lo, hi, phi=0.6, theta=0.3, dist=None):
    center = (lo + hi) * 0.5
    radius = float(np.linalg.norm(hi - lo)) * 0.5
    dist = 3.0 * radius if dist is None else dist  # OrbitCamera distance 3 x radius
    o = sc.options
    o.view_matrix = host.orbit_view_matrix(dist, phi, theta, target=center)
    o.focus_distance = dist
    o.focal_length = 50.0
    o.aspect_ratio = sc.width / sc.height


def triangle_soup(n_tris, width=1920, height=1080, bounces=8, passes=32, env=False, seed=SEED, glass_fraction=0.0,
                  clearcoat_fraction=0.0, n_materials=16, passthrough_fraction=0.0, room=False):
    """S-50k / S-1M: N random triangles, centroid ~U([-1,1]^3), two edge vectors ~U([-l,l]^3) with
    l = 0.5 N^(-1/3), de-indexed with flat normals, 16 materials, one directional light
    (theta 60 deg, phi 30 deg, illuminance 683 pi) and optionally the synthetic HDRI.

    room=True (workload c3d): the same soup and lights inside a closed, emissive-free room [-2,2]^3 of one more diffuse material,
    with the camera inside it and a 1.2 x 1.2 skylight in the ceiling as the only way in for the sun and the sky: every camera ray
    hits something and nearly every bounce does too, so paths run until Russian roulette ends them — the regime of
    physicallyBased.rlsl:277-330 the open soup (77 % of the rays leave the scene) hardly exercises."""
    rng = SplitMix64(seed)
    sc = Scene(f"soup{n_tris}", width=width, height=height)
    l = 0.5 * n_tris ** (-1.0 / 3.0)
    c = rng.uniform((n_tris, 3), -1.0, 1.0)
    e1 = rng.uniform((n_tris, 3), -l, l)
    e2 = rng.uniform((n_tris, 3), -l, l)
    third = F(1.0 / 3.0)
    v0 = (c - (e1 + e2) * third).astype(F)
    pos = np.stack([v0, v0 + e1, v0 + e2], axis=1).astype(F)  # [n,3,3]
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm = (nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)).astype(F)
    sc.materials = _material_palette(rng, n_materials, glass_fraction, clearcoat_fraction, passthrough_fraction)
    if passthrough_fraction > 0.0:  # the alpha mask: 32x32 RGBA, a quarter of the texels are holes (alpha 0)
        chk = ((np.add.outer(np.arange(32), np.arange(32)) // 4) % 4 == 0).astype(F)
        sc.textures.append((np.stack([np.ones_like(chk)] * 3 + [F(1.0) - chk], axis=-1).astype(F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
    for m in range(n_materials):  # triangle i -> material i % 16, one submesh per material
        sel = np.arange(m, n_tris, n_materials)
        if sel.size == 0:
            continue
        p = pos[sel].reshape(-1, 3)
        n = np.repeat(nrm[sel], 3, axis=0)
        mat = sc.materials[m]
        masked = mat.type == ffi.HR_MAT_PBR and bool(mat.flags & ffi.HR_MF_ALPHA_MASK)
        uv = np.tile(np.array([[0, 0], [1, 0], [0, 1]], dtype=F), (sel.size, 1)) if masked else None
        # alpha-masked primitives are not occluders: shadow rays run their shader (Mesh.cpp:95-100)
        sc.meshes.append(MeshData(p, n, np.arange(p.shape[0], dtype=np.uint32), uvs=uv, material_id=m, is_occluder=not masked))
    if room:
        h, k = 2.0, 0.6  # half size of the room, half size of the skylight
        walls = [_quad((-h, -h, h), (h, -h, h), (h, -h, -h), (-h, -h, -h)),       # floor (normal +y)
                 _quad((-h, -h, -h), (h, -h, -h), (h, h, -h), (-h, h, -h)),       # back (+z)
                 _quad((h, -h, h), (-h, -h, h), (-h, h, h), (h, h, h)),           # front (-z)
                 _quad((-h, -h, h), (-h, -h, -h), (-h, h, -h), (-h, h, h)),       # left (+x)
                 _quad((h, -h, -h), (h, -h, h), (h, h, h), (h, h, -h)),           # right (-x)
                 # ceiling (normal -y): four strips around the skylight
                 _quad((-h, h, -h), (h, h, -h), (h, h, -k), (-h, h, -k)), _quad((-h, h, k), (h, h, k), (h, h, h), (-h, h, h)),
                 _quad((-h, h, -k), (-k, h, -k), (-k, h, k), (-h, h, k)), _quad((k, h, -k), (h, h, -k), (h, h, k), (k, h, k))]
        p, n, i = _merge(walls)
        sc.materials[n_materials] = host.bake_pbr(base_color=(0.7, 0.7, 0.7), roughness=1.0, metallic=0.0)
        sc.meshes.append(MeshData(p, n, i, material_id=n_materials))
    sc.lights.add_directional(color=(1, 1, 1), illuminance=683.0 * math.pi, phi=math.radians(30.0), theta=math.radians(60.0))
    if env:
        sc.env_pixels = synthetic_hdri()
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.fstop = host.FSTOP_DISABLED
    # (room: the camera stands inside, 2.2 from the centre on the usual orbit: at (1.19, 0.65, 1.73), looking slightly down)
    _camera_for(sc, pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0), dist=2.2 if room else None)
    return sc


def terrain(nx=1000, ny=500, width=1920, height=1080, bounces=8, passes=32, env=False, seed=SEED):
    """Indexed shared-vertex grid mesh (coherent counterpart of the soup): nx x ny quads."""
    rng = SplitMix64(seed ^ 0x7E44A1)
    sc = Scene(f"terrain{nx}x{ny}", width=width, height=height)
    xs = np.linspace(-2.0, 2.0, nx + 1, dtype=np.float64)
    zs = np.linspace(-1.0, 1.0, ny + 1, dtype=np.float64)
    X, Z = np.meshgrid(xs, zs)
    Y = 0.15 * np.sin(3.1 * X) * np.cos(4.3 * Z) + 0.05 * np.sin(17.0 * X + 1.3) * np.sin(13.0 * Z)
    Y = Y + 0.01 * rng.uniform(Y.shape, -1, 1)
    pos = np.stack([X, Y, Z], axis=-1).astype(F)
    dYdx = np.gradient(Y, xs, axis=1)
    dYdz = np.gradient(Y, zs, axis=0)
    nrm = np.stack([-dYdx, np.ones_like(Y), -dYdz], axis=-1)
    nrm = (nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)).astype(F)
    uv = np.stack([(X + 2) / 4, (Z + 1) / 2], axis=-1).astype(F)
    i = np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]
    a, b, c, d = i, i + 1, i + nx + 2, i + nx + 1  # (x,z) (x+1,z) (x+1,z+1) (x,z+1)
    idx = np.stack([a, d, c, a, c, b], axis=-1).reshape(-1).astype(np.uint32)  # CCW seen from +y
    sc.materials = _material_palette(rng, 16)
    sc.meshes.append(MeshData(pos.reshape(-1, 3), nrm.reshape(-1, 3), idx, uvs=uv.reshape(-1, 2), material_id=3))
    sc.lights.add_directional(color=(1, 1, 1), illuminance=683.0 * math.pi, phi=math.radians(30.0), theta=math.radians(60.0))
    if env:
        sc.env_pixels = synthetic_hdri()
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.fstop = host.FSTOP_DISABLED
    _camera_for(sc, pos.reshape(-1, 3).min(0), pos.reshape(-1, 3).max(0))
    return sc


def uv_sphere(u_slices, v_slices, radius):
    """Lat/long sphere with the vertex / index layout of the reference's built-in sphere
    (Scene/SphereMeshProvider.h:24-177): (u+1)*(v+2) vertices, 2*u*v triangles."""
    vs = v_slices + 2
    ii, jj = np.meshgrid(np.arange(u_slices + 1), np.arange(vs), indexing="ij")
    u = ii.astype(F) / F(u_slices)
    v = jj.astype(F) / F(v_slices + 1)
    theta, phi = (u * F(2 * math.pi)).astype(F), (v * F(math.pi)).astype(F)
    p = np.stack([F(radius) * np.cos(theta) * np.sin(phi), F(radius) * np.cos(phi),
                  F(radius) * np.sin(theta) * np.sin(-phi)], axis=-1).astype(F).reshape(-1, 3)
    n = (p / np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-30)).astype(F)
    uv = np.stack([u, F(1.0) - v], axis=-1).astype(F).reshape(-1, 2)
    idx = []
    for i in range(u_slices):
        for j in range(vs - 1):
            if j == 0:
                idx += [i * vs, i * vs + 1, (i + 1) * vs + 1]
            elif j == vs - 2:
                idx += [(i + 1) * vs + j, i * vs + j, i * vs + j + 1]
            else:
                idx += [i * vs + j, i * vs + j + 1, (i + 1) * vs + j + 1, (i + 1) * vs + j + 1, (i + 1) * vs + j, i * vs + j]
    return p, n, uv, np.array(idx, dtype=np.uint32)


def plane_strip(width, length):
    """The reference's ground plane: 4 vertices, one 4-index triangle strip (Scene/PlaneMeshProvider.h:17-143)."""
    sx, sz = width * 0.5, length * 0.5
    p = np.array([[-sx, 0, sz], [sx, 0, sz], [sx, 0, -sz], [-sx, 0, -sz]], dtype=F)
    n = np.tile(np.array([0, 1, 0], dtype=F), (4, 1))
    uv = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], dtype=F)
    return p, n, uv, np.array([0, 1, 3, 2], dtype=np.uint32)


def _translate(x, y, z):
    m = np.eye(4, dtype=F)
    m[:3, 3] = (x, y, z)
    return m


def multi_material(width=320, height=180, bounces=8, passes=32, slices=24, env_color=(0.5, 0.5, 0.5), textured=False):
    """The reference's built-in "Multi-Material" scene (HeatrayRenderer.cpp:157-236): ground plane strip,
    a rough-metal sphere and a glass sphere, solid-colour environment; plus one of each analytic light so
    every light shader and NEE branch runs."""
    sc = Scene("multi_material", width=width, height=height)
    p, n, uv, i = plane_strip(15, 15)
    tex = {}
    if textured:
        rng = SplitMix64(SEED ^ 0x7E87)
        chk = ((np.add.outer(np.arange(64), np.arange(64)) // 8) % 2).astype(F)
        base = np.stack([0.3 + 0.6 * chk, 0.8 - 0.5 * chk, 0.4 + 0.2 * chk, np.ones_like(chk)], axis=-1).astype(F)
        sc.textures.append((base, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR))
        mr = (rng.uniform((32, 32, 3), 0.2, 1.0) * 255).astype(np.uint8)
        sc.textures.append((mr, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR))
        tex = dict(base_color_texture=0, metallic_roughness_texture=1)
    sc.materials[0] = host.bake_pbr(base_color=(0.9, 0.9, 0.9), roughness=1.0, metallic=0.0, specular_f0=0.0, **tex)
    sc.meshes.append(MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, world=_translate(0, -1.5, 0), material_id=0))
    sp, sn, suv, si = uv_sphere(slices, slices, 1.0)
    sc.materials[1] = host.bake_pbr(base_color=(0.4, 0.4, 0.4), roughness=0.1, metallic=1.0, specular_f0=0.3)
    sc.meshes.append(MeshData(sp, sn, si, uvs=suv, world=_translate(-0.9, -0.5, -0.8), material_id=1))
    sc.materials[2] = host.bake_glass(base_color=(0.9, 0.6, 0.6), roughness=0.1, ior=1.57, density=0.5)
    sc.meshes.append(MeshData(sp, sn, si, uvs=suv, world=_translate(1.2, -0.5, 0.8), material_id=2))
    sc.materials[3] = host.bake_pbr(base_color=(0.2, 0.5, 0.9), roughness=0.4, metallic=0.0, specular_f0=0.5,
                                    clear_coat=1.0, clear_coat_roughness=0.1)
    sc.meshes.append(MeshData(sp * F(0.5), sn, si, uvs=suv, world=_translate(0.2, -1.0, 1.8), material_id=3))
    sc.lights.add_directional(illuminance=683.0 * 2.0, phi=0.5, theta=1.0)
    sc.lights.add_point((0.0, 2.5, 1.5), color=(1.0, 0.9, 0.8), luminous_intensity=683.0 * 1.5)
    sc.lights.add_spot((-2.0, 3.0, 2.0), color=(0.7, 0.8, 1.0), luminous_intensity=683.0 * 6.0, phi=-0.7, theta=0.9,
                       inner_angle=math.radians(15), outer_angle=math.radians(35))
    sc.env_pixels = np.array(env_color, dtype=F).reshape(1, 1, 3)  # EnvironmentLight::enableSolidColor
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.aspect_ratio = width / height
    o.view_matrix = host.orbit_view_matrix(8.0, 0.5, 0.35, target=(0, -0.5, 0))
    o.focus_distance = 8.0
    o.fstop = host.FSTOP_DISABLED
    return sc


def stacked_sheets(n_sheets=120, width=64, height=36, bounces=3, passes=8, alpha_every=3):
    """`n_sheets` parallel single-sided quads seen from BEHIND, every `alpha_every`-th one an alpha-masked sheet full of holes
    instead; behind them a lit double-sided wall.  Every camera ray is re-emitted through every sheet (physicallyBased.rlsl:70-108:
    back faces of single-sided materials and alpha holes pass rays on without counting a bounce), so a path is more than
    `n_sheets` ray segments long whatever maxRayDepth says — the case the pass pipeline's per-stage counters must survive."""
    sc = Scene("stacked_sheets", width=width, height=height)
    chk = np.zeros((8, 8), dtype=F)  # alpha: holes everywhere but for one opaque texel in 64
    chk[3, 5] = 1.0
    rgba = np.stack([np.full_like(chk, 0.9), np.full_like(chk, 0.8), np.full_like(chk, 0.7), chk], axis=-1).astype(F)
    sc.textures.append((rgba, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=(0.8, 0.3, 0.2), roughness=0.6, double_sided=False)                       # back faces pass rays on
    sc.materials[1] = host.bake_pbr(base_color=(1.0, 1.0, 1.0), roughness=1.0, specular_f0=0.0, alpha_mask=True, base_color_texture=0)
    sc.materials[2] = host.bake_pbr(base_color=(0.7, 0.7, 0.75), roughness=0.9)                                          # the wall
    solid, holes = [], []
    for k in range(n_sheets):
        z = -0.02 * k
        # CCW seen from -z (normal -z): the camera at +z sees the back face
        q = _quad((-1.5, -1.0, z), (-1.5, 1.0, z), (1.5, 1.0, z), (1.5, -1.0, z))
        (holes if (alpha_every and k % alpha_every == alpha_every - 1) else solid).append(q)
    p, n, i = _merge(solid)
    sc.meshes.append(MeshData(p, n, i, material_id=0))
    if holes:
        p, n, i = _merge(holes)
        uv = np.tile(np.array([[0, 0], [0, 3], [3, 3], [3, 0]], dtype=F), (len(holes), 1))
        sc.meshes.append(MeshData(p, n, i, uvs=uv, material_id=1, is_occluder=False))  # Mesh.cpp:95-100
    zw = -0.02 * n_sheets - 0.5
    p, n, i = _quad((-3, -2, zw), (3, -2, zw), (3, 2, zw), (-3, 2, zw))  # faces +z
    sc.meshes.append(MeshData(p, n, i, material_id=2))
    sc.lights.add_directional(illuminance=683.0 * 2.0, phi=0.3, theta=1.2)
    sc.env_pixels = np.array((0.4, 0.5, 0.6), dtype=F).reshape(1, 1, 3)
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.aspect_ratio = width / height
    o.view_matrix = host.orbit_view_matrix(4.0, 0.0, 0.0, target=(0, 0, 0))
    o.focus_distance = 4.0
    o.fstop = host.FSTOP_DISABLED
    return sc


def instanced(n_side=4, slices=180, width=1920, height=1080, bounces=8, passes=32, spacing=1.0, radius=0.4):
    """n_side^2 rigid, non-overlapping objects (lat/long spheres with a ripple, one submesh and one material each) on a grid over a
    ground quad: the per-primitive transform model of OpenRL (rl.h:301-303, Scene.cpp:38-49) — each object has its own
    worldFromEntity, and an edit moves ONE of them.  4 x 4 x slices=180 is 1.05 M triangles."""
    rng = SplitMix64(SEED ^ 0x1257)
    sc = Scene(f"instanced{n_side}x{n_side}", width=width, height=height)
    sc.materials = _material_palette(rng, n_side * n_side + 1)
    p, n, uv, idx = uv_sphere(slices, slices, radius)
    ripple = (1.0 + 0.04 * np.sin(9.0 * p[:, 0:1] / radius) * np.cos(7.0 * p[:, 1:2] / radius)).astype(F)
    p = (p * ripple).astype(F)
    half = 0.5 * (n_side - 1) * spacing
    for k in range(n_side * n_side):
        gx, gz = k % n_side, k // n_side
        sc.meshes.append(MeshData(p, n, idx, uvs=uv, world=_translate(gx * spacing - half, radius * 1.1, gz * spacing - half), material_id=k))
    g = half + spacing
    gp, gn, gi = _quad((-g, 0, g), (g, 0, g), (g, 0, -g), (-g, 0, -g))
    sc.meshes.append(MeshData(gp, gn, gi, material_id=n_side * n_side))
    sc.lights.add_directional(color=(1, 1, 1), illuminance=683.0 * math.pi, phi=math.radians(30.0), theta=math.radians(60.0))
    sc.env_pixels = synthetic_hdri()
    o = sc.options
    o.max_ray_depth, o.max_render_passes = bounces, passes
    o.fstop = host.FSTOP_DISABLED
    _camera_for(sc, np.array([-g, 0.0, -g]), np.array([g, 2.2 * radius, g]))
    return sc
