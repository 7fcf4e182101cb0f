"""Randomised parity campaign: seeded random scenes over the whole parameter space of the path (every material flag and texture
slot, vertex colours / tangents, single-sided and alpha-masked surfaces, glass, all four light types in random numbers, DoF,
sample modes, strips and indexed meshes, non-uniform and mirrored transforms, odd frame sizes) — the HIP core must match the
oracle bit for bit on every one of them.  Small frames keep the oracle fast; the seeds are fixed, so failures reproduce."""
import math
import os

import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi
from heatray_amd import core, host, scenes

pytestmark = pytest.mark.gpu
F = np.float32


def random_scene(seed):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(17, 72)), int(rng.integers(9, 56))
    sc = scenes.Scene(f"fuzz{seed}", width=w, height=h)
    # ---- textures: RGBA / RGB / luminance, float and uint8, both wraps and filters
    for _ in range(int(rng.integers(2, 6))):
        tw, th, ch = int(rng.integers(1, 33)), int(rng.integers(1, 33)), int(rng.choice([1, 3, 4]))
        if rng.random() < 0.5:
            px = rng.uniform(0.0, 1.0, (th, tw, ch)).astype(F)
        else:
            px = rng.integers(0, 256, (th, tw, ch)).astype(np.uint8)
        if ch == 4 and rng.random() < 0.5:
            px[..., 3] = (rng.random((th, tw)) < 0.6) * (255 if px.dtype == np.uint8 else 1.0)   # alpha-mask material
        sc.textures.append((px, int(rng.choice([ffi.HR_WRAP_REPEAT, ffi.HR_WRAP_CLAMP_TO_EDGE])),
                            int(rng.choice([ffi.HR_FILTER_LINEAR, ffi.HR_FILTER_NEAREST]))))
    n_tex = len(sc.textures)

    def tex(p=0.5):
        return int(rng.integers(0, n_tex)) if rng.random() < p else -1

    # ---- materials
    n_mat = int(rng.integers(2, 7))
    for m in range(n_mat):
        if rng.random() < 0.3:
            sc.materials[m] = host.bake_glass(base_color=rng.uniform(0.2, 1.0, 3), roughness=float(rng.uniform(0, 1)), ior=float(rng.uniform(1.0, 2.2)),
                                              density=float(rng.uniform(0, 2)), base_color_texture=tex(0.3), normalmap=tex(0.3),
                                              metallic_roughness_texture=tex(0.3), vertex_colors=bool(rng.random() < 0.3))
        else:
            sc.materials[m] = host.bake_pbr(base_color=rng.uniform(0, 1, 3), emissive_color=rng.uniform(0, 1, 3) * (rng.random() < 0.3),
                                            roughness=float(rng.uniform(0, 1)), metallic=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])),
                                            specular_f0=float(rng.uniform(0, 1)), clear_coat=float(rng.uniform(0, 1) * (rng.random() < 0.5)),
                                            clear_coat_roughness=float(rng.uniform(0, 1)), double_sided=bool(rng.random() < 0.7),
                                            alpha_mask=bool(rng.random() < 0.3), vertex_colors=bool(rng.random() < 0.3),
                                            base_color_texture=tex(), metallic_roughness_texture=tex(0.4), emissive_texture=tex(0.2),
                                            normalmap=tex(0.4), clear_coat_texture=tex(0.2), clear_coat_roughness_texture=tex(0.2),
                                            clear_coat_normalmap=tex(0.2))
    # ---- meshes: spheres, strips, random triangle batches, with every optional attribute now and then
    for k in range(int(rng.integers(2, 6))):
        kind = rng.integers(0, 3)
        mode = ffi.HR_TRIANGLES
        if kind == 0:
            s = int(rng.integers(4, 12))
            p, n, uv, idx = scenes.uv_sphere(s, s, float(rng.uniform(0.3, 1.0)))
        elif kind == 1:
            p, n, uv, idx = scenes.plane_strip(float(rng.uniform(2, 8)), float(rng.uniform(2, 8)))
            mode = ffi.HR_TRIANGLE_STRIP
        else:
            nt = int(rng.integers(1, 40))
            p = rng.uniform(-1, 1, (nt * 3, 3)).astype(F)
            e1, e2 = p[1::3] - p[0::3], p[2::3] - p[0::3]
            fn = np.cross(e1, e2)
            fn = fn / np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-12)
            n = np.repeat(fn, 3, axis=0).astype(F)
            uv = rng.uniform(-1, 2, (nt * 3, 2)).astype(F)
            idx = np.arange(nt * 3, dtype=np.uint32)
        nv = p.shape[0]
        world = np.eye(4, dtype=F)
        world[:3, :3] = (np.diag(rng.uniform(0.5, 1.5, 3) * rng.choice([-1.0, 1.0], 3)) @ _rot(rng)).astype(F)   # may mirror
        world[:3, 3] = rng.uniform(-1.5, 1.5, 3)
        has_tb = rng.random() < 0.5
        t = rng.normal(size=(nv, 3)).astype(F)
        sc.meshes.append(scenes.MeshData(
            p.astype(F), n.astype(F), idx, uvs=uv.astype(F) if rng.random() < 0.85 else None,
            tangents=t if has_tb else None, bitangents=np.cross(n, t).astype(F) if has_tb else None,
            colors=rng.uniform(0, 1, (nv, 3)).astype(F) if rng.random() < 0.5 else None,
            mode=mode, world=world, material_id=int(rng.integers(0, n_mat + 1)),          # sometimes an unset material id
            is_occluder=bool(rng.random() < 0.8)))
    # ---- lights
    for _ in range(int(rng.integers(0, 3))):
        sc.lights.add_directional(color=rng.uniform(0.2, 1, 3), illuminance=float(rng.uniform(100, 3000)), phi=float(rng.uniform(-3, 3)),
                                  theta=float(rng.uniform(0.1, 3)))
    for _ in range(int(rng.integers(0, 3))):
        sc.lights.add_point(rng.uniform(-3, 3, 3), color=rng.uniform(0.2, 1, 3), luminous_intensity=float(rng.uniform(100, 5000)))
    for _ in range(int(rng.integers(0, 3))):
        inner = float(rng.uniform(0.05, 0.6))
        sc.lights.add_spot(rng.uniform(-3, 3, 3), color=rng.uniform(0.2, 1, 3), luminous_intensity=float(rng.uniform(100, 9000)),
                           phi=float(rng.uniform(-3, 3)), theta=float(rng.uniform(0.1, 3)), inner_angle=inner,
                           outer_angle=inner + float(rng.uniform(0.0, 0.6)))
    r = rng.random()
    if r < 0.4:
        sc.env_pixels = scenes.synthetic_hdri(64, 32)
        if rng.random() < 0.3:                                   # an 8-bit map (stays 8-bit on the device)
            sc.env_pixels = (np.clip(sc.env_pixels / 50.0, 0, 1) * 255).astype(np.uint8)
        sc.lights.env_theta_rotation = float(rng.uniform(0, 6))
        sc.env_exposure_compensation = float(rng.uniform(-2, 2))
    elif r < 0.7:
        sc.env_pixels = rng.uniform(0, 1, (1, 1, 3)).astype(F)
    # ---- camera and options
    o = sc.options
    o.max_ray_depth = int(rng.integers(0, 7))
    o.max_render_passes = 8
    o.aspect_ratio = w / h
    o.view_matrix = host.orbit_view_matrix(float(rng.uniform(3, 9)), float(rng.uniform(-3, 3)), float(rng.uniform(-1.2, 1.2)),
                                           target=tuple(rng.uniform(-0.5, 0.5, 3)))
    o.focus_distance = float(rng.uniform(2, 9))
    o.focal_length = float(rng.choice([24.0, 35.0, 50.0, 85.0]))
    o.fstop = float(rng.choice([host.FSTOP_DISABLED, 1.4, 2.8, 8.0]))
    o.sample_mode = int(rng.choice([ffi.HR_SAMPLE_SOBOL, ffi.HR_SAMPLE_HALTON, ffi.HR_SAMPLE_HAMMERSLEY]))
    o.max_channel_value = float(F(rng.choice([math.pi, 1.0, 50.0])))
    # round 5: the serial generators' tables too (std::mt19937 tables, blue noise, polygonal apertures: k_mt_tables / k_blue_noise against the
    # oracle's <random>).  Drawn from a generator of their own so that the scenes of earlier campaigns keep every other parameter.
    rng2 = np.random.default_rng(seed ^ 0x7AB1E5)
    if rng2.random() < 0.3:
        o.sample_mode = int(rng2.choice([ffi.HR_SAMPLE_RANDOM, ffi.HR_SAMPLE_BLUE_NOISE]))
    o.bokeh_shape = int(rng2.choice([ffi.HR_BOKEH_CIRCULAR, ffi.HR_BOKEH_PENTAGON, ffi.HR_BOKEH_HEXAGON, ffi.HR_BOKEH_OCTAGON], p=[0.55, 0.15, 0.15, 0.15]))
    # the importance-sampled environment + MIS estimator (include/hrcore.h) on a part of the scenes (a no-op without a map)
    o.estimator = int(rng.choice([ffi.HR_ESTIMATOR_REFERENCE, ffi.HR_ESTIMATOR_ENV_MIS, ffi.HR_ESTIMATOR_ALL_LIGHTS], p=[0.5, 0.25, 0.25]))
    # mip chain + ray-cone texture lookups (include/hrcore.h) on a third of the scenes
    o.texture_lod = ffi.HR_TEXTURE_LOD_CONE if rng.random() < 0.35 else ffi.HR_TEXTURE_LOD_BASE
    return sc


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


@pytest.mark.parametrize("seed", range(int(os.environ.get("HR_FUZZ_SEEDS", "40"))))   # HR_FUZZ_SEEDS=500 for a longer campaign
def test_random_scene_parity(golden, seed):
    sc = random_scene(1000 + seed)
    g, o = core.create_engine(), oracle_lib.engine()
    oracle_lib.load().ora_set_threads(o._ctx, 16)
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"])          # device- / oracle-generated tables (bit-identical generators)
    passes = 3
    for eng in (g, o):
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
    a, b = g.readback(), o.readback()
    nbad = int((a != b).any(axis=-1).sum())
    assert np.isfinite(b[..., 3]).all()
    assert a.tobytes() == b.tobytes(), f"seed {seed}: {nbad} of {a.shape[0] * a.shape[1]} pixels differ"
    sg, so = g.stats(), o.stats()
    for k in ("paths", "rays_closest", "rays_any", "shaded_hits", "accumulates"):
        assert getattr(sg, k) == getattr(so, k), k
    # and what the viewer would show of it
    P = ffi.display_params(tonemapping_enabled=bool(seed & 1), exposure=0.5 * (seed % 5 - 2), saturation=1.0 + 0.1 * (seed % 3))
    assert g.display(P, ffi.HR_DISPLAY_RGBA8).tobytes() == o.display(P, ffi.HR_DISPLAY_RGBA8).tobytes()


@pytest.mark.parametrize("seed", range(max(8, int(os.environ.get("HR_FUZZ_SEEDS", "40")) // 4)))
def test_random_scene_edit_sequences(golden, seed):
    # transform edits between passes: the HIP core refits (or, when the boxes degenerate, rebuilds) its tree, the oracle rebuilds;
    # every state must render bit-identically
    rng = np.random.default_rng(9000 + seed)
    sc = random_scene(3000 + seed)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"])
    n_pass = 0
    for round_ in range(4):
        for eng in (g, o):
            for s in range(2):
                eng.render_pass(sc.options.pass_params(n_pass + s))
        n_pass += 2
        a, b = g.readback(), o.readback()
        assert a.tobytes() == b.tobytes(), f"seed {seed} round {round_}: {int((a != b).any(axis=-1).sum())} pixels differ"
        for gid in rng.choice(len(sc.meshes), size=int(rng.integers(1, len(sc.meshes) + 1)), replace=False):
            m = np.eye(4, dtype=F)
            m[:3, :3] = (np.diag(rng.uniform(0.8, 1.25, 3)) @ _rot(rng)).astype(F) if rng.random() < 0.5 else np.eye(3, dtype=F)
            m[:3, 3] = rng.uniform(-0.4, 0.4, 3)
            base = sc.meshes[int(gid)].world if sc.meshes[int(gid)].world is not None else np.eye(4, dtype=F)
            w = (m @ base).astype(F)
            g.set_transform(int(gid), w), o.set_transform(int(gid), w)
        g.commit(), o.commit()
        assert list(g.scene_info().aabb_min) == list(o.scene_info().aabb_min) and g.scene_info().ray_epsilon == o.scene_info().ray_epsilon
        g.clear(), o.clear()


def _clustered_soup(rng, n_tris):
    """Triangles in tight clusters with log-uniform sizes over 5 decades: many identical Morton codes, deep and unbalanced
    trees, large triangles spanning many cells next to tiny ones."""
    n_cl = int(rng.integers(1, 12))
    centres = rng.uniform(-1, 1, (n_cl, 3))
    spread = 10.0 ** rng.uniform(-4, -0.5, n_cl)
    which = rng.integers(0, n_cl, n_tris)
    c = centres[which] + rng.normal(size=(n_tris, 3)) * spread[which, None]
    size = 10.0 ** rng.uniform(-5, 0, n_tris)
    e1 = rng.normal(size=(n_tris, 3)) * size[:, None]
    e2 = rng.normal(size=(n_tris, 3)) * size[:, None] * 10.0 ** rng.uniform(-2, 0, (n_tris, 1))   # some needles
    pos = np.stack([c, c + e1, c + e2], axis=1).astype(F)
    return pos


@pytest.mark.parametrize("seed", range(max(6, int(os.environ.get("HR_FUZZ_SEEDS", "40")) // 10)))
def test_random_traversal_vs_brute_force(seed):
    rng = np.random.default_rng(500 + seed)
    n_tris = int(rng.choice([7, 300, 2500, 9000, 20000]))
    pos = _clustered_soup(rng, n_tris)
    sc = scenes.Scene("clusters", width=16, height=16)
    nrm = np.tile(np.array([0, 1, 0], F), (n_tris * 3, 1))
    sc.materials[0] = host.bake_pbr()
    sc.meshes.append(scenes.MeshData(pos.reshape(-1, 3), nrm, np.arange(n_tris * 3, dtype=np.uint32), material_id=0))
    sc.use_multiscatter_lut = False
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng)
    oracle_lib.load().ora_set_brute_force(o._ctx, 1)
    n = 12000
    cent = pos.mean(axis=1)
    org = (cent[rng.integers(0, n_tris, n)] + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 0.5, (n, 1))).astype(F)
    aim = cent[rng.integers(0, n_tris, n)] + rng.normal(size=(n, 3)) * 1e-4 - org
    d = (aim / np.maximum(np.linalg.norm(aim, axis=1, keepdims=True), 1e-30)).astype(F)
    hg, ho = g.debug_trace(org, d), o.debug_trace(org, d)
    assert (ho["prim"] >= 0).sum() > n // 10
    assert hg.tobytes() == ho.tobytes(), f"{(hg != ho).sum()} of {n} closest hits differ ({n_tris} triangles)"
    tm = (10.0 ** rng.uniform(-3, 1, n)).astype(F)
    ag = g.debug_trace(org, d, tmax=tm, skip_prim=ho["prim"], any_hit=True)
    ao = o.debug_trace(org, d, tmax=tm, skip_prim=ho["prim"], any_hit=True)
    assert ag.tobytes() == ao.tobytes()


@pytest.mark.parametrize("seed", range(max(8, int(os.environ.get("HR_FUZZ_SEEDS", "40")) // 5)))
def test_random_scene_modes_and_shards(golden, seed):
    # the same random scenes under the per-pass modes (interactive 3x3 blocks, debug visualisers, NaN / Inf display) and as
    # tile shards: every shard must render exactly its pixels of the full frame
    rng = np.random.default_rng(7000 + seed)
    sc = random_scene(2000 + seed)
    sc.options.enable_interactive_mode = bool(rng.random() < 0.4)
    sc.options.visualizer_mode = int(rng.choice([ffi.HR_VIS_NONE, ffi.HR_VIS_NONE] + [v for k, v in vars(ffi).items() if k.startswith("HR_VIS_") and k != "HR_VIS_NONE"]))
    sc.options.show_nans, sc.options.show_inf = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    world = int(rng.integers(1, 5))
    tile = int(rng.choice([8, 16, 32, 64]))
    full = None
    acc = None
    for rank in [-1] + list(range(world)):
        kw = dict(tile_size=tile) if rank < 0 else dict(rank=rank, world=world, tile_size=tile)
        g, o = core.create_engine(**kw), oracle_lib.engine(**kw)
        for eng in (g, o):
            sc.apply(eng, lut=golden["multiscatter_lut"])
            for s in range(3):
                eng.render_pass(sc.options.pass_params(s, current_block_pixel=(s % 3, (s // 3) % 3)))
        a, b = g.readback(), o.readback()
        assert a.tobytes() == b.tobytes(), f"seed {seed} rank {rank}/{world}: {int((a != b).any(axis=-1).sum())} pixels differ"
        if rank < 0:
            full = a
        else:
            acc = a.copy() if acc is None else acc + a
    assert acc.tobytes() == full.tobytes()
