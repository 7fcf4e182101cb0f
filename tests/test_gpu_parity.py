"""GPU parity tests: libhrcore (hand-written HIP, through the C-ABI) against the CPU oracle on the same
seeded inputs.  The bar is BIT-EXACT HDR buffers (the arithmetic contract of DESIGN.md §Arithmetic);
BASELINE.json's tolerance (1e-4 relative L2) is asserted as well and reported on failure."""
import os
import re

import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi
from heatray_amd import core, host, scenes

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json: HDR output within 1e-4 relative L2


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


@pytest.fixture(scope="module")
def gpu():
    return core.create_engine()


_TABLE_CACHE = {}


def host_tables(sc):
    """Sample tables generated ONCE on the host (by the oracle's generators, themselves pinned to the
    reference's Random.h) and uploaded to both engines, as PassGenerator does with its uniform blocks."""
    key = (sc.options.sample_mode, sc.options.bokeh_shape, sc.options.max_render_passes, sc.width, sc.height)
    if key not in _TABLE_CACHE:
        o = oracle_lib.engine()
        P = sc.options.max_render_passes
        seq = np.stack([o.qmc_generate(sc.options.sample_mode, s, P) for s in range(16)])
        ap = np.stack([o.qmc_generate(ffi.HR_SAMPLE_SOBOL, s, P, radial=True) for s in range(16)])
        off = o.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, sc.width * sc.height)
        _TABLE_CACHE[key] = (seq, ap, off)
    return _TABLE_CACHE[key]


def render_both(sc, passes, lut=None, gpu_kw=None, ora_kw=None, device_tables=False):
    g = core.create_engine(**(gpu_kw or {}))
    o = oracle_lib.engine(**(ora_kw or {}))
    out = []
    tables = None if device_tables else host_tables(sc)
    for eng in (g, o):
        sc.apply(eng, lut=lut, tables=tables)
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
        out.append(eng.readback())
    return out[0], out[1], g, o


def assert_parity(g, o, what):
    r = rel_l2(g, o)
    nbad = int((g != o).any(axis=-1).sum())
    assert r <= TOL, f"{what}: rel-L2 {r:.3e} > {TOL} ({nbad} differing pixels)"
    assert g.tobytes() == o.tobytes(), f"{what}: not bit-exact: {nbad} differing pixels, rel-L2 {r:.3e}"


# ---------------------------------------------------------------------------------- tables
@pytest.mark.parametrize("name,mode", [("sobol", ffi.HR_SAMPLE_SOBOL), ("halton", ffi.HR_SAMPLE_HALTON),
                                       ("hammersley", ffi.HR_SAMPLE_HAMMERSLEY)])
def test_device_qmc_bit_exact_vs_reference_golden(golden, gpu, name, mode):
    for P in (32, 1024):
        for seq in range(16):
            got = gpu.qmc_generate(mode, seq, P)
            assert got.tobytes() == golden[f"{name}_p{P}_s{seq}"].tobytes(), (name, P, seq)


def test_device_sequence_offsets_bit_exact(golden, gpu):
    assert gpu.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, 64 * 64).tobytes() == golden["seqoffsets_64x64"].tobytes()


def test_device_radial_sobol(golden, gpu):
    # the reference uses libm sqrtf / cosf / sinf here (Random.h:278-281): libhrcore generates the Sobol points on the device
    # and applies the disk mapping on the host with the C library, so the aperture tables are the reference's bit for bit
    for seq in range(16):
        got = gpu.qmc_generate(ffi.HR_SAMPLE_SOBOL, seq, 1024, radial=True)
        assert got.tobytes() == golden[f"radialsobol_p1024_s{seq}"].tobytes()


def test_device_multiscatter_lut_vs_shipped_tiff(golden, gpu):
    lut, tid = gpu.generate_multiscatter_lut()
    want = golden["multiscatter_lut"]
    assert np.abs(lut - want).max() < 2e-6
    assert rel_l2(lut, want) < 1e-6
    # ... and the same bits as the oracle's generator (both evaluate cos / sin with the contract's Cephes restatement)
    ora_lut, _ = oracle_lib.engine().generate_multiscatter_lut()
    assert lut.tobytes() == ora_lut.tobytes(), f"{int((lut != ora_lut).sum())} texels differ from the oracle's table"


def test_device_tables_of_the_serial_generators_vs_reference_golden(golden, gpu):
    # util::uniformRandomFloats / blueNoise / randomPolygonal (Random.h:113-130, 158-165, 293-355): std::mt19937 with libstdc++'s
    # distributions and the best-candidate blue noise as HIP kernels (hr_tables.h, k_mt_tables, k_blue_noise), against the tables the
    # reference's own header produced (tests/golden/make_golden.py)
    for seq in range(16):
        assert gpu.qmc_generate(ffi.HR_SAMPLE_RANDOM, seq, 32).tobytes() == golden[f"random_p32_s{seq}"].tobytes(), seq
        assert gpu.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, seq, 32).tobytes() == golden[f"bluenoise_p32_s{seq}"].tobytes(), seq
        for shape, edges in ((ffi.HR_BOKEH_PENTAGON, 5), (ffi.HR_BOKEH_HEXAGON, 6), (ffi.HR_BOKEH_OCTAGON, 8)):
            assert gpu.aperture_generate(shape, seq, 32).tobytes() == golden[f"polygon{edges}_p32_s{seq}"].tobytes(), (edges, seq)
        assert gpu.aperture_generate(ffi.HR_BOKEH_CIRCULAR, seq, 1024).tobytes() == golden[f"radialsobol_p1024_s{seq}"].tobytes()


@pytest.mark.parametrize("count", [1, 2, 311, 312, 313, 5000])
def test_device_mt_tables_across_twists(gpu, count):
    # 624 draws per twist: 312 samples of uniformRandomFloats end exactly on a block; the polygon tables use a data-dependent number of
    # draws per sample (about 5.3), so 5000 samples walk ~40 blocks with the rejection state carried across every boundary
    o = oracle_lib.engine()
    for seq in (0, 7, 15, 123456):
        assert gpu.qmc_generate(ffi.HR_SAMPLE_RANDOM, seq, count).tobytes() == o.qmc_generate(ffi.HR_SAMPLE_RANDOM, seq, count).tobytes()
        for shape in (ffi.HR_BOKEH_PENTAGON, ffi.HR_BOKEH_HEXAGON, ffi.HR_BOKEH_OCTAGON):
            got, want = gpu.aperture_generate(shape, seq, count), o.aperture_generate(shape, seq, count)
            assert got.tobytes() == want.tobytes(), (shape, seq, int((got != want).any(axis=1).argmax()))


@pytest.mark.parametrize("count", [1, 2, 33, 700, 2500])
def test_device_blue_noise_vs_oracle(gpu, count):
    o = oracle_lib.engine()
    for seq in (0, 5, 15):
        got, want = gpu.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, seq, count), o.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, seq, count)
        assert got.tobytes() == want.tobytes(), (seq, int((got != want).any(axis=1).argmax()))
    assert (want >= 0).all() and (want <= 1).all()


def test_device_blue_noise_beyond_the_lds_copy(gpu):
    # more than 8192 points: the points stay in memory instead of LDS (k_blue_noise<false>); one sequence, ~1e9 distance tests for the checker
    o = oracle_lib.engine()
    got, want = gpu.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, 3, 8300), o.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, 3, 8300)
    assert got.tobytes() == want.tobytes(), int((got != want).any(axis=1).argmax())


@pytest.mark.parametrize("mode,shape", [(ffi.HR_SAMPLE_BLUE_NOISE, ffi.HR_BOKEH_HEXAGON), (ffi.HR_SAMPLE_RANDOM, ffi.HR_BOKEH_OCTAGON),
                                        (ffi.HR_SAMPLE_HALTON, ffi.HR_BOKEH_PENTAGON)])
def test_render_with_device_made_tables_of_every_kind(golden, mode, shape):
    # generateRandomSequences(P, mode, shape) on the device, then a depth-of-field render: the frame equals the oracle's, whose tables
    # come from its own <random> / blue-noise loops
    sc = scenes.multi_material(64, 48, bounces=3)
    sc.options.fstop = 2.0
    sc.options.sample_mode, sc.options.bokeh_shape = mode, shape
    g, o, _, _ = render_both(sc, 5, lut=golden["multiscatter_lut"], device_tables=True)
    assert_parity(g, o, f"tables {mode}/{shape}")


def test_unknown_table_kinds_fail_loudly(gpu):
    with pytest.raises(ffi.EngineError):
        gpu.qmc_generate(7, 0, 32)
    with pytest.raises(ffi.EngineError):
        gpu.generate_sequences(ffi.HR_SAMPLE_SOBOL, 9, 32)
    with pytest.raises(ffi.EngineError):
        gpu.aperture_generate(-1, 0, 32)
    with pytest.raises(ffi.EngineError):
        gpu.qmc_generate(ffi.HR_SAMPLE_BLUE_NOISE, 0, 0)


# ------------------------------------------------------------------------------- traversal
@pytest.mark.parametrize("n_tris", [3, 40, 3000, 60000])
def test_traversal_hits_bit_exact(n_tris):
    sc = scenes.triangle_soup(n_tris, width=32, height=32)
    g, o = core.create_engine(), oracle_lib.engine()
    sc.apply(g), sc.apply(o)
    gi, oi = g.scene_info(), o.scene_info()
    assert gi.n_triangles == oi.n_triangles == n_tris
    assert list(gi.aabb_min) == list(oi.aabb_min) and list(gi.aabb_max) == list(oi.aabb_max)
    assert gi.ray_epsilon == oi.ray_epsilon
    # (the product collapses the LBVH into quantised 4-wide nodes; the oracle keeps the binary spec tree, so node counts differ)
    rng = np.random.default_rng(n_tris)
    n = 20000
    org = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # a third of the rays aim at triangle centroids so that small scenes get hits too
    tgt = np.concatenate([m.positions.reshape(-1, 3, 3).mean(axis=1) for m in sc.meshes])
    k = n // 3
    aim = tgt[rng.integers(0, tgt.shape[0], k)] - org[:k]
    d[:k] = (aim / np.linalg.norm(aim, axis=1, keepdims=True)).astype(np.float32)
    hg, ho = g.debug_trace(org, d), o.debug_trace(org, d)
    assert (ho["prim"] >= 0).sum() > 100
    assert hg.tobytes() == ho.tobytes(), f"{(hg != ho).sum()} of {n} closest hits differ"
    tm = rng.uniform(0.05, 2.5, n).astype(np.float32)
    sk = ho["prim"].copy()
    ag = g.debug_trace(org, d, tmax=tm, skip_prim=sk, any_hit=True)
    ao = o.debug_trace(org, d, tmax=tm, skip_prim=sk, any_hit=True)
    assert ag.tobytes() == ao.tobytes()


@pytest.mark.parametrize("tune,builder", [("ploc=0", 0), ("ploc=2", 1), ("ploc=2,plocr=3", 1), ("ploc=1", None)])
def test_both_tree_builders_give_the_oracles_hits(monkeypatch, tune, builder):
    # hr_build.hip builds two binary trees over the Morton-ordered triangles — the radix tree (LBVH) and PLOC — and collapses the one
    # with the cheaper 4-wide tree (HR_TUNE ploc=1, the default; 0 / 2 force one).  Hits never depend on the tree: a mesh (where PLOC
    # wins), the uniform soup (where the radix tree wins) and the hostile scene give the oracle's hits bit for bit either way.
    monkeypatch.setenv("HR_TUNE", tune)
    for sc in (scenes.terrain(120, 60, width=32, height=32), scenes.triangle_soup(30000, width=32, height=32), _hostile_scene()[0]):
        g, o = core.create_engine(), oracle_lib.engine()
        sc.apply(g), sc.apply(o)
        gi = g.scene_info()
        if builder is not None and gi.n_triangles >= 4:
            assert gi.builder == builder, (sc.name, gi.builder)
        if tune != "ploc=0":
            assert gi.cost_radix > 0 and gi.cost_ploc > 0
            if tune == "ploc=1":
                assert gi.builder == (1 if gi.cost_ploc < 0.95 * gi.cost_radix else 0)
        rng = np.random.default_rng(5)
        n = 20000
        # (the hostile scene keeps the ray distribution of its own test above — origins within a few units, every ray aimed at a
        # triangle: its 8000-unit triangle puts the bounds thousands of units from half-unit triangles, and at such distances float32
        # Moeller-Trumbore itself reports phantom hits metres off a triangle, which no box test around the real triangle can — or
        # should — reproduce)
        hostile = sc.name == "hostile"
        lo, hi = (np.full(3, -1.25), np.full(3, 1.25)) if hostile else (np.array(gi.aabb_min), np.array(gi.aabb_max))
        org = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3))
        tgt = np.concatenate([m.positions[m.indices.astype(np.int64)].reshape(-1, 3, 3).mean(axis=1) for m in sc.meshes])
        k = n if hostile else n // 2
        aim = tgt[rng.integers(0, tgt.shape[0], k)] + rng.normal(scale=1e-3, size=(k, 3)) - org[:k]
        d[:k] = aim
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(np.float32)
        hg, ho = g.debug_trace(org, d), o.debug_trace(org, d)
        assert (ho["prim"] >= 0).sum() > 100
        assert hg.tobytes() == ho.tobytes(), f"{sc.name}: {(hg != ho).sum()} of {n} closest hits differ"
        tm = rng.uniform(0.05, 2.5, n).astype(np.float32) * float(np.linalg.norm(hi - lo)) * 0.3
        ag = g.debug_trace(org, d, tmax=tm, skip_prim=ho["prim"], any_hit=True)
        ao = o.debug_trace(org, d, tmax=tm, skip_prim=ho["prim"], any_hit=True)
        assert ag.tobytes() == ao.tobytes()
        # a transform edit refits whichever tree was built
        g.set_transform(0, scenes._translate(0.01, 0.02, -0.01)), o.set_transform(0, scenes._translate(0.01, 0.02, -0.01))
        g.commit(), o.commit()
        assert g.scene_info().refitted == 1
        assert g.debug_trace(org, d).tobytes() == o.debug_trace(org, d).tobytes()
        g.close(), o.close()


def test_traversal_vs_brute_force_oracle():
    sc = scenes.triangle_soup(5000, width=32, height=32)
    g, o = core.create_engine(), oracle_lib.engine()
    sc.apply(g), sc.apply(o)
    oracle_lib.load().ora_set_brute_force(o._ctx, 1)
    rng = np.random.default_rng(5)
    org = rng.uniform(-1.2, 1.2, (6000, 3)).astype(np.float32)
    d = rng.normal(size=(6000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    assert g.debug_trace(org, d).tobytes() == o.debug_trace(org, d).tobytes()


def _hostile_scene():
    """Geometry that stresses the builder and the quantised boxes: exact duplicates (ties broken by triangle id), zero-area
    and collinear triangles, axis-aligned flat sheets (zero-thickness boxes), needle triangles, tiny triangles far from a
    huge one (12 orders of magnitude of extent), many triangles with identical centroids (identical Morton codes)."""
    rng = np.random.default_rng(99)
    tris = []
    base = rng.uniform(-1, 1, (40, 3, 3)).astype(np.float32)
    tris += [base, base.copy(), base[:10].copy()]                                   # duplicates, three deep
    z = rng.uniform(-1, 1, (20, 3)).astype(np.float32)
    tris.append(np.stack([z, z, z], axis=1))                                         # points (zero area)
    tris.append(np.stack([z, z + np.float32(0.3), z + np.float32(0.6)], axis=1))     # collinear
    sheet = rng.uniform(-1, 1, (60, 3, 3)).astype(np.float32)
    sheet[:20, :, 0] = 0.25
    sheet[20:40, :, 1] = -0.5
    sheet[40:, :, 2] = 0.0
    tris.append(sheet)                                                                # flat, axis-aligned
    needle = rng.uniform(-1, 1, (30, 3, 3)).astype(np.float32)
    needle[:, 2] = needle[:, 1] + np.float32(1e-6)
    tris.append(needle)
    tiny = (rng.uniform(-1, 1, (50, 3, 3)) * 1e-6 + np.array([0.3, 0.3, 0.3])).astype(np.float32)
    tris.append(tiny)
    tris.append(np.array([[[-4000, -4000, -3], [4000, -4000, -3], [0, 4000, -3]]], np.float32))   # huge
    c = rng.uniform(-0.5, 0.5, (1, 1, 3)).astype(np.float32)
    fan = (c + rng.uniform(-0.2, 0.2, (64, 3, 3)).astype(np.float32))
    fan -= fan.mean(axis=1, keepdims=True) - c                                        # 64 triangles, one centroid
    tris.append(fan.astype(np.float32))
    pos = np.concatenate(tris).astype(np.float32)
    sc = scenes.Scene("hostile", width=32, height=32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0] * 3, 1))
    sc.materials = scenes._material_palette(scenes.SplitMix64(1), 2)
    sc.meshes.append(scenes.MeshData(pos.reshape(-1, 3), nrm, np.arange(pos.shape[0] * 3, dtype=np.uint32), material_id=0))
    sc.lights.add_directional(color=(1, 1, 1), illuminance=10.0, phi=0.3, theta=0.5)
    scenes._camera_for(sc, np.array([-1, -1, -1], np.float32), np.array([1, 1, 1], np.float32))
    sc.options.max_ray_depth, sc.options.max_render_passes = 3, 8
    sc.options.fstop = host.FSTOP_DISABLED
    return sc, pos


def test_hostile_geometry_hits_bit_exact(golden):
    sc, pos = _hostile_scene()
    g, o, ob = core.create_engine(), oracle_lib.engine(), oracle_lib.engine()
    for e in (g, o, ob):
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    oracle_lib.load().ora_set_brute_force(ob._ctx, 1)
    rng = np.random.default_rng(4)
    n = 30000
    org = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    cent = pos.mean(axis=1)
    aim = cent[rng.integers(0, cent.shape[0], n)] + rng.normal(scale=1e-3, size=(n, 3)).astype(np.float32) - org
    d = (aim / np.maximum(np.linalg.norm(aim, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    d[: n // 10] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 10)] * rng.choice([-1.0, 1.0], (n // 10, 1)).astype(np.float32)  # axis-parallel rays
    hg, ho, hb = g.debug_trace(org, d), o.debug_trace(org, d), ob.debug_trace(org, d)
    assert (hb["prim"] >= 0).sum() > n // 4
    assert ho.tobytes() == hb.tobytes()                       # the oracle's tree against its brute force
    assert hg.tobytes() == hb.tobytes(), f"{(hg != hb).sum()} of {n} closest hits differ"
    # duplicates: the winner is the smallest triangle id among equal distances
    dup = hb["prim"][(hb["prim"] >= 0) & (hb["prim"] < 90)]
    assert dup.size and (dup < 40).all()
    tm = rng.uniform(0.01, 3.0, n).astype(np.float32)
    ag = g.debug_trace(org, d, tmax=tm, skip_prim=hb["prim"], any_hit=True)
    ab = ob.debug_trace(org, d, tmax=tm, skip_prim=hb["prim"], any_hit=True)
    assert ag.tobytes() == ab.tobytes()
    # and a render through the whole path
    for s in range(3):
        g.render_pass(sc.options.pass_params(s)), o.render_pass(sc.options.pass_params(s))
    assert_parity(g.readback(), o.readback(), "hostile geometry render")


@pytest.mark.parametrize("tune", ["", "ploc=2"])
def test_deep_tree_stresses_the_traversal_stack(monkeypatch, tune):
    # A deliberately deep tree: 8192 coincident triangles inside one Morton cell of a scene 2000 units wide (the radix tree splits
    # them on index bits, and every ray through them hits all four children at every level), a chain of 28 nested, mutually
    # overlapping sheets whose centroids sit in ever smaller Morton cells, and some filler.  The traversal stack (16 LDS entries
    # + private overflow, sized for 3 entries per level of the deepest tree the builder can make) must neither overflow nor drop
    # a hit; equal distances are resolved towards the smallest triangle id.
    rng = np.random.default_rng(123)
    one = np.array([[-0.01, -0.01, 0.5], [0.02, -0.01, 0.5], [-0.01, 0.02, 0.5]], np.float32)
    coincident = np.tile(one[None], (8192, 1, 1))
    sheets = []
    for j in range(28):
        c, s = 700.0 * 2.0 ** -j, 2.5 * 700.0 * 2.0 ** -j
        sheets.append([[c - s, c - s, 1.0 + 0.01 * j], [c + 2 * s, c - s, 1.0 + 0.01 * j], [c - s, c + 2 * s, 1.0 + 0.01 * j]])
    sheets = np.array(sheets, np.float32)
    filler = rng.uniform(-1000, 1000, (300, 1, 3)).astype(np.float32) + rng.uniform(-30, 30, (300, 3, 3)).astype(np.float32)
    pos = np.concatenate([coincident, sheets, filler]).astype(np.float32)
    sc = scenes.Scene("deep", width=16, height=16)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0] * 3, 1))
    sc.materials = scenes._material_palette(scenes.SplitMix64(2), 1)
    sc.meshes.append(scenes.MeshData(pos.reshape(-1, 3), nrm, np.arange(pos.shape[0] * 3, dtype=np.uint32), material_id=0))
    monkeypatch.setenv("HR_TUNE", tune)
    g, ob = core.create_engine(), oracle_lib.engine()
    sc.apply(g), sc.apply(ob)
    oracle_lib.load().ora_set_brute_force(ob._ctx, 1)
    info = g.scene_info()
    assert info.bvh_levels >= (8 if not tune else 5), info.bvh_levels   # a shallow tree would not test anything
    assert 3 * info.bvh_levels <= 16 + 160                          # kStackLDS + kStackOvf (hr_trace.h)
    n = 4000
    org = np.zeros((n, 3), np.float32)
    org[:, :2] = rng.uniform(-0.02, 0.03, (n, 2))
    org[:, 2] = rng.choice([-5.0, 8.0], n)
    d = np.zeros((n, 3), np.float32)
    d[:, 2] = -np.sign(org[:, 2])
    d[:, :2] = rng.normal(scale=2e-3, size=(n, 2))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hg, hb = g.debug_trace(org, d), ob.debug_trace(org, d)
    assert (hb["prim"] >= 0).mean() > 0.5
    assert hg.tobytes() == hb.tobytes(), f"{(hg != hb).sum()} of {n} closest hits differ"
    first = hb["prim"][(hb["prim"] >= 0) & (hb["prim"] < 8192)]
    assert first.size > 100 and (first == 0).all()                  # 8192 equal distances: triangle 0 wins
    tm = rng.uniform(0.5, 12.0, n).astype(np.float32)
    ag = g.debug_trace(org, d, tmax=tm, skip_prim=hb["prim"], any_hit=True)
    ab = ob.debug_trace(org, d, tmax=tm, skip_prim=hb["prim"], any_hit=True)
    assert ag.tobytes() == ab.tobytes()


def test_tiny_scenes_and_frames(golden):
    # 1, 2 and 5 triangles (root leaf / a root with leaf children only), frames of 1x1, 33x17 and 1x64 pixels
    for n_tris, (w, h) in [(1, (1, 1)), (2, (33, 17)), (5, (1, 64)), (1, (40, 24))]:
        sc = scenes.triangle_soup(n_tris, width=w, height=h, bounces=2, passes=4, env=True)
        g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"], device_tables=True)
        assert_parity(g, o, f"{n_tris} triangles at {w}x{h}")
        assert (g[..., 3] == 3).all()


# --------------------------------------------------------------------------------- renders
STAT_KEYS = ("paths", "rays_closest", "rays_any", "shaded_hits", "accumulates")


def test_cornell_config1(golden):
    # BASELINE config 1: Cornell box (32 tris), 256x256, 4 bounces
    sc = scenes.cornell_box(256, 256, bounces=4, passes=32)
    g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "cornell 256x256")
    gs, os_ = ge.stats().as_dict(), oe.stats().as_dict()
    for k in STAT_KEYS:
        assert gs[k] == os_[k], (k, gs[k], os_[k])


@pytest.mark.parametrize("textured", [False, True])
def test_multi_material_all_shaders(golden, textured):
    # PBR (diffuse / GGX / clearcoat / multiscatter LUT), glass, directional + point + spot lights,
    # solid environment, triangle strip, textures
    sc = scenes.multi_material(160, 90, bounces=8, passes=16, textured=textured)
    g, o, ge, oe = render_both(sc, 4, lut=golden["multiscatter_lut"])
    assert_parity(g, o, f"multi_material textured={textured}")
    gs, os_ = ge.stats().as_dict(), oe.stats().as_dict()
    for k in STAT_KEYS:
        assert gs[k] == os_[k], (k, gs[k], os_[k])


def test_device_generated_tables_and_lut_render():
    # THE DEFAULT PRODUCT PATH (what bench.py and the C++ layer use): everything generated on the device (QMC tables, offsets,
    # aperture table, LUT) against everything generated by the oracle — bit for bit, like every other parity test
    sc = scenes.multi_material(96, 54, bounces=6, passes=16)
    g, o, _, _ = render_both(sc, 4, device_tables=True)
    assert_parity(g, o, "device-generated tables and LUT")
    sc = scenes.triangle_soup(5000, width=96, height=54, bounces=6, passes=16, env=True, glass_fraction=0.25, clearcoat_fraction=0.25)
    sc.options.fstop = 2.8  # the aperture table matters
    g, o, _, _ = render_both(sc, 3, device_tables=True)
    assert_parity(g, o, "device-generated tables and LUT, depth of field")


def test_soup_with_environment(golden):
    sc = scenes.triangle_soup(20000, width=160, height=90, bounces=8, passes=16, env=True)
    g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "soup20k+env")
    assert ge.stats().rays_any == oe.stats().rays_any


def test_closed_room_paths_bounce(golden):
    # workload c3d at a small size: the soup inside a closed room with a skylight, camera inside — nearly every ray hits, paths are
    # five closest-hit segments long on average (the open soup: 1.4), Russian roulette (physicallyBased.rlsl:279-288) ends them
    sc = scenes.triangle_soup(20000, width=160, height=96, bounces=8, passes=32, env=True, room=True)
    g, o, ge, oe = render_both(sc, 6, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "closed room")
    st, ost = ge.stats(), oe.stats()
    assert (st.paths, st.rays_closest, st.rays_any, st.shaded_hits) == (ost.paths, ost.rays_closest, ost.rays_any, ost.shaded_hits)
    assert st.rays_closest >= 4 * st.paths and st.shaded_hits >= 0.8 * st.rays_closest
    assert g[..., :3].max() > 0


def test_known_answer_scenes_of_the_surface_inputs(golden):
    # the scenes whose oracle images tests/test_oracle_kat_surface.py checks against closed forms (textures in every slot, vertex colours,
    # a tangent-space normal map, the clamp, an alpha-masked non-occluder crossed by camera and shadow rays, a single-sided back face):
    # the HIP path gives the same bits
    import test_oracle_kat_surface as kat
    for what, sc in kat.gpu_parity_scenes():
        g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])
        assert_parity(g, o, what)
        assert ge.stats().rays_closest == oe.stats().rays_closest and ge.stats().rays_any == oe.stats().rays_any
        ge.close(), oe.close()


def test_glass_clearcoat_dof_config5_small(golden):
    # BASELINE config 5 at test size: 25 % glass, 25 % clearcoat, f/2.8 depth of field, 16 bounces
    sc = scenes.triangle_soup(4000, width=96, height=54, bounces=16, passes=8, env=True, glass_fraction=0.25,
                              clearcoat_fraction=0.25)
    sc.options.fstop = 2.8
    g, o, _, _ = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "config5-small")


def test_terrain_indexed_mesh(golden):
    sc = scenes.terrain(60, 30, width=128, height=72, bounces=6, passes=8, env=True)
    g, o, _, _ = render_both(sc, 2, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "terrain")


def test_single_sided_and_alpha_mask_passthrough(golden):
    sc = scenes.multi_material(96, 54, bounces=4, passes=8, textured=True)
    # ground: single-sided + alpha mask with a texture whose alpha has holes; clearcoat sphere: single sided
    chk = ((np.add.outer(np.arange(16), np.arange(16)) // 2) % 2).astype(np.float32)
    rgba = np.stack([np.full_like(chk, 0.8), np.full_like(chk, 0.7), np.full_like(chk, 0.6), chk], axis=-1)
    sc.textures[0] = (rgba, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST)
    sc.materials[0] = host.bake_pbr(base_color=(0.9, 0.9, 0.9), roughness=1.0, specular_f0=0.0, alpha_mask=True,
                                    double_sided=False, base_color_texture=0)
    sc.meshes[0].is_occluder = False  # Mesh.cpp:95-100
    sc.materials[3] = host.bake_pbr(base_color=(0.2, 0.5, 0.9), roughness=0.4, double_sided=False)
    g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "passthrough")


def test_pass_through_chain_longer_than_the_stage_ring(golden):
    # 150 stacked single-sided / alpha-masked sheets seen from behind.  Every path is > 150 ray segments long
    # (pass-through re-emission has no depth bound), twice the number of per-stage counters a pass slot holds (kMaxBounceSlots = 72),
    # which are therefore a ring.  Bit-exact against the oracle, whose loop is unbounded, also with several passes in flight.
    sc = scenes.stacked_sheets(150, width=64, height=36, bounces=3, passes=8)
    g, o, ge, oe = render_both(sc, 5, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "150 stacked sheets")
    gs, os_ = ge.stats().as_dict(), oe.stats().as_dict()
    for k in STAT_KEYS:
        assert gs[k] == os_[k], (k, gs[k], os_[k])
    assert gs["rays_closest"] >= 150 * gs["paths"] * 0.4  # the chains really are that long
    # an orbit view from the side: most rays cross only some of the sheets, passes retire at different stages
    sc.options.view_matrix = host.orbit_view_matrix(4.0, 1.2, 0.4, target=(0, 0, -1.0))
    g, o, _, _ = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "150 stacked sheets, oblique")


def test_passthrough_scene_full_size_pipelined(golden):
    # A 1080p glTF-like scene: 30 % of the materials single-sided or alpha-masked, whose rays pass through back faces and holes
    # and are not bounded by maxRayDepth.  Such passes are pipelined like any other (the host retires a pass when a snapshot of
    # its queue lengths shows the queue ran empty); the result must not depend on that scheduling.
    sc = scenes.triangle_soup(50000, width=1920, height=1080, bounces=8, passes=32, passthrough_fraction=0.3)
    assert sum(1 for m in sc.materials.values() if not (m.flags & ffi.HR_MF_DOUBLE_SIDED)) == 5
    e = core.create_engine()
    sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    passes = 12                                               # more than a batch: passes overlap in the pipeline
    for s in range(passes):
        e.render_pass(sc.options.pass_params(s))
    a = e.readback()
    st = e.stats()
    assert (a[..., 3] == passes).all() and np.isfinite(a).all()
    assert st.paths == 1920 * 1080 * passes
    e.clear()                                                 # one pass at a time (readback completes each): same bits
    acc = None
    for s in range(3):
        e.render_pass(sc.options.pass_params(s))
        acc = e.readback()
    e.clear()
    for s in range(3):
        e.render_pass(sc.options.pass_params(s))
    assert e.readback().tobytes() == acc.tobytes()
    for tile_id in (1007, 523):                               # tiles of the 12-pass frame against the oracle, bit for bit
        o = oracle_lib.engine(rank=tile_id, world=2040, tile_size=32)
        sc.apply(o, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(passes):
            o.render_pass(sc.options.pass_params(s))
        ob = o.readback()
        own = ob[..., 3] > 0
        assert own.sum() == 32 * 32
        assert a[own].tobytes() == ob[own].tobytes()
    # rays did pass through something: a path may have more closest-hit segments than maxRayDepth + 1 allows otherwise
    e.close()


@pytest.mark.parametrize("u8_env", [False, True])
def test_env_mis_estimator_bit_exact(golden, u8_env):
    # HR_ESTIMATOR_ENV_MIS (include/hrcore.h): the importance table of the environment map is built on the device (integer sums:
    # order-independent) and the one-sample MIS estimator reproduces the oracle's arithmetic — bit-exact HDR buffers again
    sc = scenes.multi_material(96, 54, bounces=4, passes=16, textured=True)
    env = scenes.synthetic_hdri(256, 128)
    sc.env_pixels = (np.clip(env / 50.0, 0, 1) * 255).astype(np.uint8) if u8_env else env
    sc.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS
    sc.lights.env_theta_rotation = 0.7
    g, o, ge, oe = render_both(sc, 6, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "env MIS estimator, multi-material")
    assert ge.stats().rays_any == oe.stats().rays_any
    sc.options.estimator = ffi.HR_ESTIMATOR_REFERENCE               # switching back and forth on live engines
    for eng in (ge, oe):
        eng.clear()
        for s_ in range(3):
            eng.render_pass(sc.options.pass_params(s_))
    ref_g, ref_o = ge.readback(), oe.readback()
    assert_parity(ref_g, ref_o, "reference estimator after the MIS one")
    assert ref_g.tobytes() != g.tobytes()
    soup = scenes.triangle_soup(20000, width=64, height=36, bounces=6, passes=16, env=True)
    soup.env_pixels = env
    soup.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS
    g2, o2, _, _ = render_both(soup, 5, lut=golden["multiscatter_lut"])
    assert_parity(g2, o2, "env MIS estimator, soup with HDRI")


def test_all_lights_estimator_bit_exact(golden):
    # HR_ESTIMATOR_ALL_LIGHTS (include/hrcore.h): two occlusion rays per vertex, the pass sample kept as two partial sums that meet in
    # the resolve — bit-exact against the oracle's restatement; pass slots are re-allocated when the mode is first used on a live engine
    sc = scenes.multi_material(96, 54, bounces=5, passes=16, textured=True)
    sc.env_pixels = scenes.synthetic_hdri(256, 128)
    sc.lights.add_point((0.5, 2.0, 1.0), luminous_intensity=683.0 * 2.0)
    sc.lights.add_spot((-1.0, 3.0, 0.5), luminous_intensity=683.0 * 40.0)
    g0, o0, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])          # reference estimator first: slots exist at the old size
    assert_parity(g0, o0, "reference estimator before the switch")
    sc.options.estimator = ffi.HR_ESTIMATOR_ALL_LIGHTS
    for eng in (ge, oe):
        eng.clear()
        for s_ in range(6):
            eng.render_pass(sc.options.pass_params(s_))
    g, o = ge.readback(), oe.readback()
    assert_parity(g, o, "all-lights estimator, multi-material")
    assert ge.stats().rays_any == oe.stats().rays_any
    assert g.tobytes() != g0.tobytes()
    sc.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS                              # and back, passes of both kinds in one pipeline
    for eng in (ge, oe):
        for s_ in range(6, 9):
            eng.render_pass(sc.options.pass_params(s_))
    assert_parity(ge.readback(), oe.readback(), "MIS passes on top of all-lights passes")
    soup = scenes.triangle_soup(20000, width=64, height=36, bounces=6, passes=16, env=True)
    soup.options.estimator = ffi.HR_ESTIMATOR_ALL_LIGHTS
    soup.options.texture_lod = ffi.HR_TEXTURE_LOD_CONE
    g2, o2, e2, oe2 = render_both(soup, 5, lut=golden["multiscatter_lut"])
    assert_parity(g2, o2, "all-lights estimator + cone LOD, soup with HDRI")
    assert e2.stats().rays_any == oe2.stats().rays_any


def test_all_lights_estimator_on_glass_and_clearcoat(golden):
    # glass vertices send the analytic-light ray and the environment ray as well (glass.rlsl:83-129 picks one light)
    sc = scenes.triangle_soup(6000, width=96, height=54, bounces=10, passes=16, env=True, glass_fraction=0.4, clearcoat_fraction=0.3)
    sc.lights.add_point((0.3, 0.8, 0.2), luminous_intensity=683.0 * 3.0)
    sc.options.estimator = ffi.HR_ESTIMATOR_ALL_LIGHTS
    g, o, ge, oe = render_both(sc, 5, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "all-lights estimator, glass + clearcoat soup")
    assert ge.stats().rays_any == oe.stats().rays_any
    sc.options.estimator = ffi.HR_ESTIMATOR_REFERENCE
    r, _, re, _ = render_both(sc, 5, lut=golden["multiscatter_lut"])
    assert ge.stats().rays_any > re.stats().rays_any                    # more occlusion rays per vertex ...
    assert abs(float(g[..., :3].mean()) - float(r[..., :3].mean())) < 0.25 * float(r[..., :3].mean())   # ... towards the same image


def test_texture_lod_cone_bit_exact(golden):
    # HR_TEXTURE_LOD_CONE (include/hrcore.h): mip chains and per-triangle texel densities are built on the device, the ray cone rides
    # in the ray record, the trilinear lookup follows the oracle's arithmetic — bit-exact HDR buffers, f32 and u8 textures
    sc = scenes.multi_material(96, 54, bounces=5, passes=16, textured=True)
    rng = np.random.default_rng(11)
    sc.textures[0] = (rng.random((37, 50, 3)).astype(np.float32), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR)            # odd, non-square
    sc.textures[1] = ((rng.random((64, 16, 3)) * 255).astype(np.uint8), ffi.HR_WRAP_CLAMP_TO_EDGE, ffi.HR_FILTER_LINEAR)
    sc.options.texture_lod = ffi.HR_TEXTURE_LOD_CONE
    g, o, ge, oe = render_both(sc, 5, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "cone LOD, multi-material")
    sc.options.texture_lod = ffi.HR_TEXTURE_LOD_BASE                   # switching back and forth on live engines
    for eng in (ge, oe):
        eng.clear()
        for s_ in range(3):
            eng.render_pass(sc.options.pass_params(s_))
    base_g, base_o = ge.readback(), oe.readback()
    assert_parity(base_g, base_o, "level 0 after the cone mode")
    assert base_g.tobytes() != g.tobytes()
    # a transform edit changes world areas, hence the per-triangle level offsets: they are rebuilt with the commit
    sc.options.texture_lod = ffi.HR_TEXTURE_LOD_CONE
    m = np.diag([2.5, 0.4, 1.7, 1.0]).astype(np.float32)
    for eng in (ge, oe):
        eng.set_transform(0, m)
        eng.commit()
        eng.clear()
        for s_ in range(3):
            eng.render_pass(sc.options.pass_params(s_))
    assert_parity(ge.readback(), oe.readback(), "cone LOD after a transform edit")
    # the lookup itself (base-colour visualiser) on a fine checker at three distances: level 0, a fractional level, the uniform top
    import test_oracle_kat as kat
    for dist in (0.05, 6.0, 40.0):
        ck = kat.checker_plane(dist, ffi.HR_TEXTURE_LOD_CONE, size=32)
        g2, o2, _, _ = render_both(ck, 1, lut=golden["multiscatter_lut"])
        assert_parity(g2, o2, f"checker at {dist}")


def test_debug_visualizers(golden):
    sc = scenes.multi_material(64, 36, bounces=2, textured=True)
    for mode in (ffi.HR_VIS_GEOMETRIC_NORMALS, ffi.HR_VIS_UVS, ffi.HR_VIS_FINAL_NORMALS, ffi.HR_VIS_BASE_COLOR,
                 ffi.HR_VIS_ROUGHNESS, ffi.HR_VIS_METALLIC, ffi.HR_VIS_SHADER):
        sc.options.visualizer_mode = mode
        g, o, _, _ = render_both(sc, 1, lut=golden["multiscatter_lut"])
        assert_parity(g, o, f"visualizer {mode}")
    sc.options.visualizer_mode = ffi.HR_VIS_NONE
    sc.options.show_nans = True
    g, o, _, _ = render_both(sc, 1, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "nan visualizer")
    sc.options.show_nans, sc.options.show_inf = False, True          # accumulator.rlsl:16-21, the Inf detector
    g, o, _, _ = render_both(sc, 1, lut=golden["multiscatter_lut"])
    assert_parity(g, o, "inf visualizer")
    sc.options.show_inf = False
    # the remaining modes need tangent space / a normal map / clearcoat maps: a mesh with tangents and every texture slot filled
    rng = np.random.default_rng(17)
    tex = lambda c: ((rng.uniform(0.2, 1.0, (16, 16, c)) * 255).astype(np.uint8), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR)
    sc2 = scenes.multi_material(64, 36, bounces=2)
    sc2.textures = [tex(4), tex(3), tex(3), tex(3), tex(1), tex(1), tex(3)]
    sc2.materials[3] = host.bake_pbr(base_color=(0.7, 0.6, 0.5), roughness=0.5, metallic=0.2, clear_coat=1.0, clear_coat_roughness=0.3,
                                     base_color_texture=0, metallic_roughness_texture=1, emissive_texture=2, normalmap=3,
                                     clear_coat_texture=4, clear_coat_roughness_texture=5, clear_coat_normalmap=6)
    for me in sc2.meshes:
        n = me.positions.shape[0]
        me.tangents = np.tile(np.array([1, 0, 0], np.float32), (n, 1))
        me.bitangents = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
        me.material_id = 3
    for mode in (ffi.HR_VIS_TANGENTS, ffi.HR_VIS_BITANGENTS, ffi.HR_VIS_NORMALMAP, ffi.HR_VIS_EMISSIVE, ffi.HR_VIS_CLEARCOAT,
                 ffi.HR_VIS_CLEARCOAT_ROUGHNESS, ffi.HR_VIS_CLEARCOAT_NORMALMAP):
        sc2.options.visualizer_mode = mode
        g, o, _, _ = render_both(sc2, 1, lut=golden["multiscatter_lut"])
        assert_parity(g, o, f"visualizer {mode}")
        assert g[..., :3].max() > 0.0, mode                            # the mode really drew something


def test_interactive_block_mode(golden):
    sc = scenes.multi_material(66, 39, bounces=3)
    sc.options.enable_interactive_mode = True
    imgs = []
    for eng in (core.create_engine(), oracle_lib.engine()):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for by in range(3):
            for bx in range(3):
                eng.render_pass(sc.options.pass_params(0, current_block_pixel=(bx, by)))
        imgs.append(eng.readback())
    assert_parity(imgs[0], imgs[1], "interactive")
    assert (imgs[0][..., 3] == 1.0).all()  # nine sub-passes sample every pixel of each 3x3 block once
    # the reference shuffles the block table (PassGenerator.cpp:276-278): a host-supplied shuffled list on both engines
    perm = np.random.default_rng(5).permutation(9)
    coords = np.array([(i // 3, i % 3) for i in perm], dtype=np.int32).reshape(3, 3, 2)   # (row, col) pairs in texture order
    shuffled = []
    for eng in (core.create_engine(), oracle_lib.engine()):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        eng.set_interactive_blocks(coords)
        eng.render_pass(sc.options.pass_params(0, current_block_pixel=(1, 2)))
        first = eng.readback()
        for by in range(3):
            for bx in range(3):
                if (bx, by) != (1, 2):
                    eng.render_pass(sc.options.pass_params(0, current_block_pixel=(bx, by)))
        shuffled.append((first, eng.readback()))
    assert_parity(shuffled[0][0], shuffled[1][0], "interactive, shuffled table, one sub-pass")
    assert_parity(shuffled[0][1], shuffled[1][1], "interactive, shuffled table")
    assert (shuffled[0][1][..., 3] == 1.0).all()          # a permutation still covers every pixel exactly once
    assert shuffled[0][0].tobytes() != imgs[0].tobytes()


def test_tile_shards_equal_full_frame(golden):
    # SURVEY §8e acceptance: N-shard HDR buffer == 1-shard HDR buffer bit for bit
    sc = scenes.multi_material(160, 96, bounces=5)
    full, _, _, _ = render_both(sc, 2, lut=golden["multiscatter_lut"])
    parts = []
    for r in range(3):
        e = core.create_engine(rank=r, world=3, tile_size=32)
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(2):
            e.render_pass(sc.options.pass_params(s))
        parts.append(e.readback())
    assert (sum((p[..., 3] > 0).astype(int) for p in parts) == 1).all()
    assert (parts[0] + parts[1] + parts[2]).tobytes() == full.tobytes()


def test_tile_shards_and_interactive_blocks_with_the_opt_in_modes(golden):
    # the all-lights estimator keeps four partial sums per pass sample behind each other in full-frame indexing, the cone LOD carries
    # per-path state: both under tile sharding (shards must add up to the full frame) and under the 3x3 interactive block mode
    sc = scenes.multi_material(160, 96, bounces=4, textured=True)
    sc.env_pixels = scenes.synthetic_hdri(128, 64)
    sc.options.estimator = ffi.HR_ESTIMATOR_ALL_LIGHTS
    sc.options.texture_lod = ffi.HR_TEXTURE_LOD_CONE
    full, ofull, _, _ = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert_parity(full, ofull, "all-lights + cone LOD, full frame")
    parts = []
    for r in range(3):
        e = core.create_engine(rank=r, world=3, tile_size=32)
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(3):
            e.render_pass(sc.options.pass_params(s))
        parts.append(e.readback())
    assert (sum((p[..., 3] > 0).astype(int) for p in parts) == 1).all()
    assert (parts[0] + parts[1] + parts[2]).tobytes() == full.tobytes()
    sc.options.enable_interactive_mode = True
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for k in range(9):
            eng.render_pass(sc.options.pass_params(0, current_block_pixel=(k % 3, k // 3)))
    assert_parity(g.readback(), o.readback(), "all-lights + cone LOD, interactive blocks")


def test_reset_resize_and_transform(golden):
    sc = scenes.cornell_box(48, 48, bounces=3)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        eng.render_pass(sc.options.pass_params(0))
        eng.clear()                                               # rlClear
        eng.set_transform(0, scenes._translate(0.1, 0.0, -0.05))  # Scene::applyTransform
        eng.commit()
        eng.render_pass(sc.options.pass_params(0))
        eng.render_pass(sc.options.pass_params(1))
    assert_parity(g.readback(), o.readback(), "after clear + transform")
    off = g.qmc_generate(ffi.HR_SAMPLE_SOBOL, 0, 40 * 24)
    sc.options.aspect_ratio = 40 / 24
    for eng in (g, o):
        eng.resize(40, 24)
        eng.set_seq_offsets(off)
        eng.render_pass(sc.options.pass_params(0))
    assert_parity(g.readback(), o.readback(), "after resize")


def test_error_paths(gpu):
    e = core.create_engine()
    with pytest.raises(ffi.EngineError):
        e.render_pass(host.RenderOptions().pass_params(0))  # no frame
    e.resize(16, 16)
    with pytest.raises(ffi.EngineError):
        e.render_pass(host.RenderOptions().pass_params(0))  # scene not committed
    with pytest.raises(ffi.EngineError):
        e.add_mesh(np.zeros((3, 3)), np.zeros((3, 3)), [0, 1, 5])  # index out of range
    with pytest.raises(ffi.EngineError):
        core.create_engine(rank=2, world=2)
    # shard-exchange entry points: argument checks before anything is launched
    f = core.create_engine()
    with pytest.raises(ffi.EngineError):
        f.packed_slots(0, 1)                              # no frame yet
    f.resize(40, 24)
    assert f.packed_slots(0, 1) == 2 * 1 * 32 * 32        # 2 x 1 tiles of 32 x 32 slots (edge tiles padded)
    assert f.packed_slots(1, 2) == 32 * 32 and f.packed_slots(2, 3) == 0
    with pytest.raises(ffi.EngineError):
        f.packed_slots(2, 2)
    with pytest.raises(ffi.EngineError):
        f.pack_owned(0)                                   # null output
    with pytest.raises(ffi.EngineError):
        f.unpack(0, 0, 1, 1)                              # world 0


# ------------------------------------------------------------- full BASELINE sizes: properties
def test_full_size_config2_properties(golden):
    # BASELINE config 2 size (1080p, ~50k tris, 8 bounces): size-independent properties
    sc = scenes.triangle_soup(50000, width=1920, height=1080, bounces=8, passes=32)
    e = core.create_engine()
    sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    passes = 3
    for s in range(passes):
        e.render_pass(sc.options.pass_params(s))
    a = e.readback()
    st = e.stats()
    assert (a[..., 3] == passes).all()                       # every pixel sampled once per pass
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    assert st.paths == 1920 * 1080 * passes
    assert st.rays_closest + st.rays_any <= 1920 * 1080 * passes * 2 * (8 + 1)   # SURVEY §8a ray budget
    # each accumulate() is clamped to maxChannelValue: bounded energy per pixel
    assert a[..., :3].max() <= sc.options.max_channel_value * passes * (2 * 9 + 1)
    # idempotence: clearing and re-rendering reproduces the buffer bit for bit (no atomics in the data path)
    e.clear()
    for s in range(passes):
        e.render_pass(sc.options.pass_params(s))
    assert e.readback().tobytes() == a.tobytes()
    # linearity of the accumulator: passes 0..2 == pass 0 + pass 1 + pass 2 rendered separately
    parts = []
    for s in range(passes):
        e.clear()
        e.render_pass(sc.options.pass_params(s))
        parts.append(e.readback().astype(np.float64))
    assert np.allclose(parts[0] + parts[1] + parts[2], a, rtol=1e-5, atol=1e-6)
    # the WHOLE full-size frame against the oracle, every one of the 2,073,600 pixels bit for bit (the oracle renders a 1080p pass of
    # this scene in well under a second on the box's 16 host threads)
    o = oracle_lib.engine()
    sc.apply(o, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    for s in range(passes):
        o.render_pass(sc.options.pass_params(s))
    ob = o.readback()
    assert ob.shape == a.shape == (1080, 1920, 4)
    assert_parity(a, ob, "config 2, whole 1080p frame")
    ost = o.stats()
    assert (st.paths, st.rays_closest, st.rays_any, st.shaded_hits) == (ost.paths, ost.rays_closest, ost.rays_any, ost.shaded_hits)


# ------------------------------------------------------------- the pass pipeline's scheduling never changes the result
@pytest.mark.parametrize("tune", ["groups=1,batch=1", "groups=2", "groups=3,batch=2", "groups=2,batch=5,depth=24",
                                  "fmin=16,fmax=256,refill=16,tri=8,blocks=3", "groups=4,batch=16",
                                  # round 3's knobs: every launch through the work cursor, every launch dealt out statically
                                  "sdeal=0,batch=3", "sdeal=1000000,groups=1",
                                  # the work fetch: one cursor / 64 range cursors, camera rays in long / short chunks
                                  "sdeal=0,heads=0,fprim=256,fgate=0", "sdeal=0,heads=6,fprim=32,fgate=0,fmin=16,batch=7"])
def test_pipeline_scheduling_is_result_invariant(golden, monkeypatch, tune):
    # passes in flight on several streams, several passes per launch, chunked work fetch: the HDR buffer must not
    # depend on any of it (resolves happen in pass order; no float atomics)
    monkeypatch.setenv("HR_TUNE", tune)
    sc = scenes.multi_material(96, 64, bounces=5, textured=True)
    g, o, ge, oe = render_both(sc, 11, lut=golden["multiscatter_lut"])
    assert_parity(g, o, f"multi_material with HR_TUNE={tune}")
    # passes of different depths must still be added in pass order
    ge.clear(), oe.clear()
    for eng in (ge, oe):
        for s, depth in enumerate([5, 5, 1, 1, 7, 0, 3, 3, 3]):
            pp = sc.options.pass_params(s)
            pp.max_ray_depth = depth
            eng.render_pass(pp)
    assert_parity(ge.readback(), oe.readback(), f"mixed depths with HR_TUNE={tune}")
    assert ge.stats().paths == oe.stats().paths


_C3_FULL = {}


def _config3_full_size(golden):
    """BASELINE config 3 at full size, once per session: the scene, the one-context GPU frame of two passes, its counters, and the
    oracle's WHOLE frame of the same two passes (2,073,600 pixels; ~1 s per pass on 16 host threads)."""
    if not _C3_FULL:
        sc = scenes.triangle_soup(1_000_000, width=1920, height=1080, bounces=8, passes=32, env=True)
        e = core.create_engine()
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        assert e.scene_info().n_triangles == 1_000_000
        passes = 2
        for s in range(passes):
            e.render_pass(sc.options.pass_params(s))
        a = e.readback()
        st = e.stats()
        e.clear()                                                  # idempotence: no atomics in the data path
        for s in range(passes):
            e.render_pass(sc.options.pass_params(s))
        assert e.readback().tobytes() == a.tobytes()
        e.close()
        o = oracle_lib.engine()
        sc.apply(o, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(passes):
            o.render_pass(sc.options.pass_params(s))
        _C3_FULL.update(sc=sc, passes=passes, gpu=a, gpu_stats=st, oracle=o.readback(), oracle_stats=o.stats())
    return _C3_FULL


def test_full_size_config3_whole_frame(golden):
    # BASELINE config 3 (1080p, 1M tris, HDRI + NEE, 8 bounces) — the north-star workload: every pixel of the frame against the oracle
    c3 = _config3_full_size(golden)
    a, ob, st, ost, passes = c3["gpu"], c3["oracle"], c3["gpu_stats"], c3["oracle_stats"], c3["passes"]
    assert a.shape == ob.shape == (1080, 1920, 4)
    assert (a[..., 3] == passes).all()
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    assert st.paths == 1920 * 1080 * passes
    assert st.rays_closest + st.rays_any <= 1920 * 1080 * passes * 2 * (8 + 1)
    assert_parity(a, ob, "config 3, whole 1080p frame (2,073,600 pixels)")
    assert (st.paths, st.rays_closest, st.rays_any, st.shaded_hits) == (ost.paths, ost.rays_closest, ost.rays_any, ost.shaded_hits)


def test_full_size_config4_eight_shards_of_config3(golden):
    # BASELINE config 4: config 3's frame as an 8-way tile shard at FULL size.  One device here, so the eight ranks' contexts render
    # one after the other (each is exactly what rank r of an 8-GPU job runs: same library, same rank / world description); no
    # collective is involved in producing the pixels (pixels are copied, never summed).  Acceptance of 8e: the shards are
    # disjoint, together they are the one-context frame bit for bit, and that frame is the oracle's.
    from heatray_amd import tiles
    c3 = _config3_full_size(golden)
    sc, passes, one = c3["sc"], c3["passes"], c3["gpu"]
    world = 8
    owner = tiles.owner_map(1920, 1080, world)
    total = np.zeros_like(one)
    covered = np.zeros(one.shape[:2], dtype=np.int32)
    rays = paths = 0
    for r in range(world):
        e = core.create_engine(rank=r, world=world, tile_size=32)
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(passes):
            e.render_pass(sc.options.pass_params(s))
        f = e.readback()
        st = e.stats()
        e.close()
        mine = owner == r
        assert (f[~mine] == 0).all(), f"rank {r} wrote pixels it does not own"
        assert (f[mine][:, 3] == passes).all()
        assert st.paths == int(mine.sum()) * passes
        total += f                                   # (exact: every pixel has one non-zero contributor)
        covered += (f[..., 3] > 0).astype(np.int32)
        rays += st.rays_closest + st.rays_any
        paths += st.paths
    assert (covered == 1).all()                      # disjoint and complete
    assert total.tobytes() == one.tobytes()          # 8 shards == the one-context frame, bit for bit
    assert_parity(total, c3["oracle"], "config 4: eight shards of config 3 vs the oracle's whole frame")
    gst = c3["gpu_stats"]
    assert paths == gst.paths and rays == gst.rays_closest + gst.rays_any


def test_full_size_config5_properties(golden):
    # BASELINE config 5 at full size on one GPU: 3840x2160, 16 bounces, 1M triangles, 25 % glass + 25 % clearcoat materials,
    # f/2.8 depth of field with the PENTAGON aperture (util::randomPolygonal tables from the device generator)
    W, H, depth = 3840, 2160, 16
    sc = scenes.triangle_soup(1_000_000, width=W, height=H, bounces=depth, passes=32, env=True, glass_fraction=0.25, clearcoat_fraction=0.25)
    sc.options.fstop = 2.8
    sc.options.bokeh_shape = ffi.HR_BOKEH_PENTAGON
    assert sum(1 for m in sc.materials.values() if m.type == ffi.HR_MAT_GLASS) == 4
    assert sum(1 for m in sc.materials.values() if m.type == ffi.HR_MAT_PBR and m.clear_coat > 0) == 4
    e = core.create_engine()
    sc.apply(e, lut=golden["multiscatter_lut"])
    assert e.scene_info().n_triangles == 1_000_000
    passes = 2
    for s in range(passes):
        e.render_pass(sc.options.pass_params(s))
    a = e.readback()
    st = e.stats()
    assert a.shape == (H, W, 4)
    assert (a[..., 3] == passes).all()
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all()
    assert st.paths == W * H * passes
    assert st.rays_closest + st.rays_any <= W * H * passes * 2 * (depth + 1)       # SURVEY §8a ray budget
    assert st.rays_closest > st.paths and st.rays_any > 0                          # paths bounce and sample lights
    assert a[..., :3].max() <= sc.options.max_channel_value * passes * (2 * (depth + 1) + 1)
    e.clear()                                                                      # idempotence
    for s in range(passes):
        e.render_pass(sc.options.pass_params(s))
    assert e.readback().tobytes() == a.tobytes()
    # one sixteenth of the full-size frame's tiles against the oracle, bit for bit: the interleaved shard 5 of 16 — 510 tiles spread
    # over the whole frame, the ragged top row (2160 = 67 * 32 + 16) included; ~518,000 pixels
    from heatray_amd import tiles
    o = oracle_lib.engine(rank=5, world=16, tile_size=32)
    sc.apply(o, lut=golden["multiscatter_lut"])
    for s in range(passes):
        o.render_pass(sc.options.pass_params(s))
    ob = o.readback()
    own = tiles.owner_map(W, H, 16) == 5
    assert abs(int(own.sum()) - W * H // 16) <= 1024 and (ob[own][:, 3] == passes).all() and (ob[~own] == 0).all()
    assert any(t // 120 == 67 for t in tiles.owned_tiles(W, H, 5, 16))            # cropped top-row tiles are part of the sample
    r = rel_l2(a[own], ob[own])
    assert r <= TOL
    assert a[own].tobytes() == ob[own].tobytes(), f"config 5: {int((a[own] != ob[own]).any(axis=-1).sum())} of {int(own.sum())} pixels differ (rel-L2 {r:.3e})"
    ost = o.stats()
    assert ost.shaded_hits > 0 and ost.rays_any > 0
    e.close()


# ------------------------------------------------------------- display resolve
DISPLAY_SETTINGS = [
    dict(),                                                                     # the reference's defaults
    dict(tonemapping_enabled=True),
    dict(tonemapping_enabled=True, exposure=1.5, brightness=0.1, contrast=1.1, hue=0.9, saturation=1.7, vibrance=0.6,
         red=1.2, green=0.8, blue=1.4, vignette_intensity=0.7, vignette_falloff=0.3),
    dict(exposure=-3.0, brightness=-0.4, saturation=0.0, vignette_intensity=1.0, vignette_falloff=0.0),   # negative colours -> NaN paths
]


@pytest.mark.parametrize("k", range(len(DISPLAY_SETTINGS)))
def test_display_resolve_bit_exact(golden, k):
    sc = scenes.multi_material(200, 120, bounces=4, textured=True)
    g, o, ge, oe = render_both(sc, 3, lut=golden["multiscatter_lut"])
    assert g.tobytes() == o.tobytes()
    P = ffi.display_params(**DISPLAY_SETTINGS[k])
    for fmt in (ffi.HR_DISPLAY_RGBA8, ffi.HR_DISPLAY_RGBA32F, ffi.HR_DISPLAY_HDR_RGBA32F):
        a, b = ge.display(P, fmt), oe.display(P, fmt)
        assert a.shape == b.shape and a.dtype == b.dtype
        assert a.tobytes() == b.tobytes(), f"display format {fmt}, settings {k}: {int((a != b).any(axis=-1).sum())} pixels differ"
    rgba8 = ge.display(P, ffi.HR_DISPLAY_RGBA8)
    assert (rgba8[..., 3] == 255).all()
    if k == 0:
        assert rgba8[..., :3].std() > 10          # an actual image, not a constant


def test_display_resolve_device_output_shards_and_errors(golden):
    import torch
    sc = scenes.multi_material(100, 70, bounces=3)
    P = ffi.display_params(tonemapping_enabled=True, exposure=0.5)
    full = None
    parts = []
    for world, rank in [(1, 0), (3, 0), (3, 1), (3, 2)]:
        e = core.create_engine(rank=rank, world=world, tile_size=32)
        sc.apply(e, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(2):
            e.render_pass(sc.options.pass_params(s))    # display completes the pipeline by itself
        out = torch.zeros((sc.height, sc.width), dtype=torch.int32, device="cuda")
        e.display_device(out.data_ptr(), P, ffi.HR_DISPLAY_RGBA8)
        e.synchronize()
        img = out.cpu().numpy().view(np.uint8).reshape(sc.height, sc.width, 4)
        assert img.tobytes() == e.display(P, ffi.HR_DISPLAY_RGBA8).tobytes()
        if world == 1:
            full = img
        else:
            parts.append(img)
    # shards write only their own pixels: their images add up to the full one
    assert (parts[0].astype(np.int32) + parts[1] + parts[2] == full).all()
    e = core.create_engine()
    with pytest.raises(ffi.EngineError):
        e.display(P)                                      # no frame
    e.resize(8, 8)
    with pytest.raises(ffi.EngineError):
        e.display(P, 7)                                   # unknown format


# ------------------------------------------------------------- BASELINE metric 2: passes-to-converge
def test_passes_to_converge_agrees_with_the_oracle(golden):
    # a property of the estimator, not of speed: the HIP core and the oracle must report the same pass counts
    from heatray_amd import convergence as cv
    sc = scenes.cornell_box(32, 32, bounces=3, passes=96)
    g, o = core.create_engine(), oracle_lib.engine()
    tables = host_tables(sc)
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=tables)
    oracle_lib.load().ora_set_threads(o._ctx, 8)
    ref_g = cv.reference_image(g, sc.options, 96, sc.width, sc.height)
    ref_o = cv.reference_image(o, sc.options, 96, sc.width, sc.height)
    assert ref_g.tobytes() == ref_o.tobytes()
    got, want = [], []
    for run in range(3):
        ng, eg = cv.run_with_readback(g, sc.options, run, 96, ref_g, sc.width, sc.height, threshold=0.15)
        no, eo = cv.run_with_readback(o, sc.options, run, 96, ref_o, sc.width, sc.height, threshold=0.15)
        assert eg == eo                                   # err(n) identical for every n (bit-exact buffers)
        got.append(ng), want.append(no)
    assert got == want and all(n is not None and n > 1 for n in got)
    assert cv.p50(got, 96) == cv.p50(want, 96)
    assert eg[0] > eg[-1]                                  # the error falls as passes accumulate


def test_pentagon_bokeh_dof_config5_aperture(golden):
    # BASELINE config 5's pentagon bokeh: util::randomPolygonal(5) tables, made by k_mt_tables on the one side and by the oracle's <random> on the other
    sc = scenes.multi_material(96, 64, bounces=4)
    sc.options.fstop = 2.8
    sc.options.bokeh_shape = ffi.HR_BOKEH_PENTAGON
    g, o, _, _ = render_both(sc, 4, lut=golden["multiscatter_lut"], device_tables=True)
    assert_parity(g, o, "pentagon bokeh DoF")
    sc2 = scenes.multi_material(96, 64, bounces=4)
    sc2.options.fstop = 2.8
    g2, _, _, _ = render_both(sc2, 4, lut=golden["multiscatter_lut"], device_tables=True)
    assert g2.tobytes() != g.tobytes()                      # the aperture shape does change the image


def test_scene_edit_sequences_keep_parity(golden):
    # SURVEY §8f row 3 (dynamic updates): transform-only edits reuse the resident geometry (fast re-commit: assemble + LBVH
    # only), adding / removing geometry re-stages it; every state must still render like the oracle's
    sc = scenes.multi_material(64, 48, bounces=3)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    tri_p = np.array([[-0.5, 0.2, 0.1], [0.5, 0.2, 0.1], [0.0, 0.9, 0.3]], np.float32)
    tri_n = np.tile(np.array([0, 0, 1], np.float32), (3, 1))

    def both(fn):
        return fn(g), fn(o)

    def check(what, passes=2):
        for eng in (g, o):
            eng.clear()
            for s in range(passes):
                eng.render_pass(sc.options.pass_params(s))
        assert_parity(g.readback(), o.readback(), what)

    check("initial")
    both(lambda e: (e.set_transform(0, scenes._translate(0.2, 0.0, 0.1)), e.commit()))           # transform only -> reuse
    check("after a transform")
    both(lambda e: (e.set_transform(1, scenes._translate(-0.1, 0.05, 0.0)), e.set_transform(0, scenes._translate(0, 0, 0)), e.commit()))
    check("after two more transforms")
    ids = both(lambda e: e.add_mesh(tri_p, tri_n, [0, 1, 2], material_id=1))                     # new geometry -> re-staged
    assert ids[0] == ids[1]
    both(lambda e: e.commit())
    check("after adding a triangle")
    both(lambda e: (e.set_transform(ids[0], scenes._translate(0.0, -0.3, 0.2)), e.commit()))     # reuse again, new geom included
    check("after moving the new triangle")
    both(lambda e: (e.remove_mesh(ids[0]), e.commit()))
    check("after removing it")
    assert g.scene_info().n_triangles == o.scene_info().n_triangles


def _random_transform(rng, scale_range=(0.7, 1.4), shift=0.6, angle=0.8):
    ang = rng.uniform(-angle, angle, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = (rx @ ry @ rz) * rng.uniform(*scale_range)
    m[:3, 3] = rng.uniform(-shift, shift, 3)
    return m.astype(np.float32)


def test_hundred_random_edit_steps_refit_and_rebuild(golden):
    # a commit after transform-only edits REFITS the tree (same topology, boxes recomputed bottom-up on the
    # device); adding / removing geometry, or a refit whose boxes have degenerated, rebuilds it.  The hit is defined by the
    # triangle test alone, so every state must trace and render exactly like the oracle's freshly built scene.
    rng = np.random.default_rng(2026)
    sc = scenes.triangle_soup(6000, width=48, height=32, bounces=3, passes=8, env=True, n_materials=6)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    n_geoms = len(sc.meshes)
    live = list(range(n_geoms))
    cur = {gid: np.eye(4, dtype=np.float32) for gid in live}   # worldFromEntity of every submesh, as Scene keeps it
    def place(gid, m):
        cur[gid] = m
        g.set_transform(gid, m), o.set_transform(gid, m)
    tri_n = np.tile(np.array([0, 0, 1], np.float32), (3, 1))
    n_rays = 3000
    refits = rebuilds = 0
    for step in range(100):
        kind = rng.choice(["one", "all", "far", "add", "remove"], p=[0.45, 0.3, 0.05, 0.1, 0.1])
        expect_refit = True
        if kind == "one":                                    # one submesh moves a little (a dragged object): its triangles share
            gid = int(rng.choice(live))                      # leaves with the others', whose boxes grow; small motions stay refits
            place(gid, (_random_transform(rng, scale_range=(0.99, 1.01), shift=0.01, angle=0.01) @ cur[gid]).astype(np.float32))
            expect_refit = None
        elif kind == "all":                                  # Scene::applyTransform: the whole scene, rigidly (Scene.cpp:38-49)
            big = step % 3 == 0                              # a large rotation inflates every axis-aligned box: refit or rebuild
            m = _random_transform(rng, scale_range=(0.5, 2.0), shift=2.0, angle=0.8 if big else 0.15)
            for gid in live:                                 # transform * submesh.transform for every submesh
                place(gid, (m @ cur[gid]).astype(np.float32))
            expect_refit = None                              # (a rotation inflates axis-aligned boxes: refit or rebuild, by the guard)
        elif kind == "far":                                  # one submesh flung far away: the refitted boxes degenerate -> rebuild
            gid = int(rng.choice(live))
            place(gid, _random_transform(rng, shift=40.0))
            expect_refit = None
        elif kind == "add":
            p = rng.uniform(-0.8, 0.8, (3, 3)).astype(np.float32)
            a, b = g.add_mesh(p, tri_n, [0, 1, 2], material_id=1), o.add_mesh(p, tri_n, [0, 1, 2], material_id=1)
            assert a == b
            live.append(a)
            cur[a] = np.eye(4, dtype=np.float32)
            expect_refit = False
        else:
            if len(live) > n_geoms:
                gid = live.pop()
                g.remove_mesh(gid), o.remove_mesh(gid)
                expect_refit = False
        g.commit(), o.commit()
        gi, oi = g.scene_info(), o.scene_info()
        assert gi.n_triangles == oi.n_triangles
        assert list(gi.aabb_min) == list(oi.aabb_min) and list(gi.aabb_max) == list(oi.aabb_max) and gi.ray_epsilon == oi.ray_epsilon
        if expect_refit is not None:
            assert bool(gi.refitted) == expect_refit, (step, kind)
        refits += int(gi.refitted)
        rebuilds += int(not gi.refitted)
        # the tree-quality figure the guard acts on: 1 right after a build, at most the guard's 1.25 for a tree that was kept
        if gi.n_triangles > 4:
            assert (abs(gi.box_area_ratio - 1.0) < 1e-3) if not gi.refitted else (0.0 < gi.box_area_ratio <= 1.25 * 1.001), (step, kind, gi.refitted, gi.box_area_ratio)
        lo, hi = np.array(gi.aabb_min), np.array(gi.aabb_max)
        org = rng.uniform(lo - 0.2, hi + 0.2, (n_rays, 3)).astype(np.float32)
        tgt = rng.uniform(lo, hi, (n_rays, 3))
        d = tgt - org
        d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20)).astype(np.float32)
        hg, ho = g.debug_trace(org, d), o.debug_trace(org, d)
        assert hg.tobytes() == ho.tobytes(), f"step {step} ({kind}): {(hg != ho).sum()} of {n_rays} hits differ"
        if step % 10 == 9:                                   # and through the whole path
            for eng in (g, o):
                eng.clear()
                for s_ in range(2):
                    eng.render_pass(sc.options.pass_params(s_))
            assert_parity(g.readback(), o.readback(), f"render after edit step {step} ({kind})")
    assert refits >= 40 and rebuilds >= 5, (refits, rebuilds)


def test_tree_cache_round_trip(tmp_path, monkeypatch):
    # hr_scene_cache: the tree of a scene is written on the first commit and read back — after a device-side content hash of the
    # meshes matched — by the next context; a changed scene misses; hits never depend on where the tree came from
    sc = scenes.triangle_soup(30000, width=32, height=32, n_materials=4)
    path = tmp_path / "scene.hrbvh"
    rng = np.random.default_rng(8)
    org = rng.uniform(-1.2, 1.2, (8000, 3)).astype(np.float32)
    d = rng.normal(size=(8000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = oracle_lib.engine()
    sc.apply(o)
    want = o.debug_trace(org, d)
    a = core.create_engine()
    a.set_scene_cache(path)
    sc.apply(a)
    assert a.scene_info().refitted == 0 and path.exists() and path.stat().st_size > 30000 * 4
    assert a.debug_trace(org, d).tobytes() == want.tobytes()
    b = core.create_engine()
    b.set_scene_cache(path)
    sc.apply(b)
    assert b.scene_info().refitted == 2                                   # read, not built
    assert b.scene_info().n_nodes == a.scene_info().n_nodes
    ia, ib = a.scene_info(), b.scene_info()                               # the file says which of the two candidate trees it holds, and their costs
    assert (ib.builder, ib.cost_radix, ib.cost_ploc) == (ia.builder, ia.cost_radix, ia.cost_ploc) and ia.cost_radix > 0
    assert b.debug_trace(org, d).tobytes() == want.tobytes()
    b.set_transform(0, scenes._translate(0.01, 0.0, 0.0))                 # a cached tree refits like a built one
    b.commit()
    o.set_transform(0, scenes._translate(0.01, 0.0, 0.0))
    o.commit()
    assert b.scene_info().refitted == 1
    assert b.debug_trace(org, d).tobytes() == o.debug_trace(org, d).tobytes()
    sc2 = scenes.triangle_soup(30000, width=32, height=32, n_materials=4, seed=scenes.SEED + 1)   # other geometry, same counts
    c2 = core.create_engine()
    c2.set_scene_cache(path)
    sc2.apply(c2)
    assert c2.scene_info().refitted == 0                                  # key mismatch: built (and the file replaced)
    o2 = oracle_lib.engine()
    sc2.apply(o2)
    assert c2.debug_trace(org, d).tobytes() == o2.debug_trace(org, d).tobytes()
    # the builder options are part of the key: a file made by one builder is not served to a context that asks for the other
    for tune, builder in (("ploc=2", 1), ("ploc=0", 0)):
        monkeypatch.setenv("HR_TUNE", tune)
        e1 = core.create_engine()
        e1.set_scene_cache(path)
        sc2.apply(e1)
        assert e1.scene_info().refitted == 0 and e1.scene_info().builder == builder     # (built: the file was the other builder's)
        e2 = core.create_engine()
        e2.set_scene_cache(path)
        sc2.apply(e2)
        assert e2.scene_info().refitted == 2 and e2.scene_info().builder == builder     # ... and now it is this one's
        assert e2.debug_trace(org, d).tobytes() == o2.debug_trace(org, d).tobytes()
        e1.close(), e2.close()
    monkeypatch.delenv("HR_TUNE")
    path.write_bytes(path.read_bytes()[:1000])                            # a truncated file is ignored
    c3 = core.create_engine()
    c3.set_scene_cache(path)
    sc2.apply(c3)
    assert c3.scene_info().refitted == 0
    assert c3.debug_trace(org, d).tobytes() == o2.debug_trace(org, d).tobytes()


def test_tree_cache_rejects_damaged_files_and_keys_are_stable(tmp_path):
    # (1) the key covers the scene, not the file — a full-length file with flipped bits, or one written by something else
    # with out-of-range child / slot indices and a *matching* checksum, must be rejected (a rebuild), never uploaded; (2) the key must
    # not depend on the unwritten alignment padding of the mesh blocks (vertex counts that are not multiples of four, odd index
    # counts, after allocator churn).
    import struct
    sc = scenes.triangle_soup(6001, width=32, height=32, n_materials=3)     # 6003 vertices in the first mesh: 12 * nVerts is not a multiple of 16
    assert sc.meshes[0].positions.shape[0] % 4 != 0
    rng = np.random.default_rng(9)
    org = rng.uniform(-1.2, 1.2, (4000, 3)).astype(np.float32)
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = oracle_lib.engine()
    sc.apply(o)
    want = o.debug_trace(org, d).tobytes()
    path = tmp_path / "scene.hrbvh"

    def fresh(expect_cached):
        e = core.create_engine()
        e.set_scene_cache(path)
        sc.apply(e)
        assert (e.scene_info().refitted == 2) == expect_cached, e.scene_info().refitted
        assert e.debug_trace(org, d).tobytes() == want
        e.close()

    fresh(False)                      # builds, writes the file
    good = path.read_bytes()
    # churn the allocator: contexts with other scenes come and go, so the next blocks land on recycled memory
    for k in range(3):
        t = core.create_engine()
        scenes.triangle_soup(5000 + 777 * k, width=16, height=16, seed=scenes.SEED + 5 + k).apply(t)
        t.close()
    fresh(True)                       # same scene, recycled memory: the key still matches
    assert path.read_bytes() == good  # (a hit does not rewrite the file)
    header = 8 + 4 + 4 + 8 + 6 * 4 + 65 * 4 + 2 * 4   # (CacheHeader, version 4: ... levelStart[65], the two candidates' costs)
    header += (-header) % 8           # the 64-bit checksum is 8-byte aligned
    sum_off = header
    payload_off = header + 8
    (stored,) = struct.unpack_from("<Q", good, sum_off)
    assert stored != 0
    # (a) one flipped bit in the node array, same length: checksum mismatch
    bad = bytearray(good)
    bad[payload_off + 64 * 5 + 40] ^= 0x10
    path.write_bytes(bytes(bad))
    fresh(False)
    # the rebuild wrote the file again: same scene key, same size (node order may differ from build to build, hits never do)
    again = path.read_bytes()
    assert again[:24] == good[:24] and len(again) == len(good) and again != bytes(bad)
    good = again
    (stored,) = struct.unpack_from("<Q", good, sum_off)
    # (b) bytes appended
    path.write_bytes(good + b"\0" * 64)
    fresh(False)
    # (c) a hostile file: child base of node 0 far out of range AND the checksum recomputed to match
    lib = core.load_library()
    bad = bytearray(good)
    struct.pack_into("<I", bad, payload_off + 32 + 8, 0x0FFFFFF0)        # Node4::c.z (innerBase) of node 0
    import ctypes as C
    def checksum(b):
        h = 0x9E3779B97F4A7C15
        M = (1 << 64) - 1
        n = len(b)
        i = 0
        while i + 8 <= n:
            (w,) = struct.unpack_from("<Q", b, i)
            h = ((h ^ w) * 0xFF51AFD7ED558CCD) & M
            h ^= h >> 29
            i += 8
        tail = int.from_bytes(b[i:], "little") if i < n else 0
        h = ((h ^ tail ^ n) * 0xC4CEB9FE1A85EC53) & M
        return h ^ (h >> 32)
    struct.pack_into("<Q", bad, sum_off, checksum(bytes(bad[payload_off:])))
    path.write_bytes(bytes(bad))
    fresh(False)                      # range check, not the checksum, stops this one
    # (d) the same with a slot of the prim -> slot map out of range
    bad = bytearray(good)
    struct.pack_into("<I", bad, len(bad) - 4, 0x7FFFFFFF)
    struct.pack_into("<Q", bad, sum_off, checksum(bytes(bad[payload_off:])))
    path.write_bytes(bytes(bad))
    fresh(False)


def test_strided_and_interleaved_vertex_buffers(golden):
    # rlVertexAttribBuffer takes a byte stride (Mesh.cpp:104-132): attributes interleaved in one buffer, and padded planar
    # buffers, reach the device as they are and are read with their stride
    sc = scenes.multi_material(64, 36, bounces=3, textured=True)
    g, o = core.create_engine(), oracle_lib.engine()
    sc.apply(o, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    plain = sc.meshes
    inter = []
    for me in plain:
        n = me.positions.shape[0]
        buf = np.zeros((n, 11), np.float32)                  # pos(3) pad(1) nrm(3) uv(2) pad(2)
        buf[:, 0:3], buf[:, 4:7] = me.positions, me.normals
        if me.uvs is not None:
            buf[:, 7:9] = me.uvs
        inter.append(buf)
    sc.meshes = []
    sc.apply(g, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    sc.meshes = plain
    for me, buf in zip(plain, inter):
        g.add_mesh_strided(buf, pos_off=0, nrm_off=4, uv_off=7 if me.uvs is not None else None, stride_floats=11, indices=me.indices,
                           mode=me.mode, world=me.world, is_occluder=me.is_occluder, material_id=me.material_id)
    g.commit()
    for eng in (g, o):
        eng.clear()
        for s_ in range(2):
            eng.render_pass(sc.options.pass_params(s_))
    assert_parity(g.readback(), o.readback(), "interleaved vertex buffers")


@pytest.mark.parametrize("tune,passes", [("batch=1,slow=0", 24), ("slow=0", 288), ("", 96)])
def test_progressive_readback_never_drains_and_holds_complete_passes(golden, monkeypatch, tune, passes):
    # hr_readback_progressive: whatever is in the buffer is a prefix of the passes, complete, bit-identical to the oracle's
    # image of that many passes; the pipeline is not completed by it
    # (a frame this small injects 16 passes at a time into two groups unless HR_TUNE says otherwise; slow=0 switches off the completion
    # for callers slower than 4 ms per pass, which is timing-dependent: with it the image must LAG (asserted below); the third case
    # runs with the default, where a slow test host may trigger completions — every image is still a complete prefix, and the
    # display call itself says how many passes it shows)
    monkeypatch.setenv("HR_TUNE", tune)
    sc = scenes.multi_material(64, 48, bounces=4, passes=320)
    g, o = core.create_engine(), oracle_lib.engine()
    oracle_lib.load().ora_set_threads(o._ctx, 16)
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    oracle_frames = {}
    for s in range(passes):
        o.render_pass(sc.options.pass_params(s))
        oracle_frames[s + 1] = o.readback()
    seen = []
    for s in range(passes):
        g.render_pass(sc.options.pass_params(s))
        buf, n = g.readback_progressive()
        assert 0 <= n <= s + 1
        if n:
            assert (buf[..., 3] == n).all()
            assert buf.tobytes() == oracle_frames[n].tobytes()
        else:
            assert (buf == 0).all()
        seen.append(n)
    # the progressive display snapshot shows exactly those passes
    shown, n_now = g.display(ffi.display_params(tonemapping_enabled=True), ffi.HR_DISPLAY_RGBA32F | ffi.HR_DISPLAY_PROGRESSIVE, with_passes=True)
    assert 0 <= n_now <= passes
    if n_now:
        o2 = oracle_lib.engine()
        sc.apply(o2, lut=golden["multiscatter_lut"], tables=host_tables(sc))
        for s in range(n_now):
            o2.render_pass(sc.options.pass_params(s))
        assert shown.tobytes() == o2.display(ffi.display_params(tonemapping_enabled=True), ffi.HR_DISPLAY_RGBA32F).tobytes()
    assert seen == sorted(seen)
    if "slow=0" in tune:
        assert seen[-1] < passes                           # lags behind: the pipeline was never completed
        assert max(seen) > 0
    full = g.readback()                                        # this one completes everything
    assert full.tobytes() == oracle_frames[passes].tobytes()
    _, n = g.readback_progressive()
    assert n == passes
    g.clear()
    _, n = g.readback_progressive()
    assert n == 0


def test_a_slow_caller_does_not_wait_for_a_batch_to_fill(golden):
    # Passes are injected a batch at a time (12 x 1080p paths per macro step; 32 passes on a frame this small).  A caller that issues
    # ONE pass and then only asks for pixels — the viewer at its refresh rate — must still get that pass: with an idle device,
    # hr_readback_progressive launches what is waiting instead of leaving it queued until a batch is full.
    import time
    sc = scenes.multi_material(64, 48, bounces=4, passes=64)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    assert g.pass_batch(sc.options.max_ray_depth) > 1
    for k in range(3):                                       # one pass per "frame", three frames
        g.render_pass(sc.options.pass_params(k))
        o.render_pass(sc.options.pass_params(k))
        t0, n, buf = time.perf_counter(), 0, None
        while n < k + 1 and time.perf_counter() - t0 < 5.0:
            buf, n = g.readback_progressive()
            time.sleep(0.002)
        assert n == k + 1, f"pass {k} was still waiting for a batch after 5 s"
        assert buf.tobytes() == o.readback().tobytes()


@pytest.mark.parametrize("tune", ["", "groups=2", "groups=2,batch=1"])
def test_step_log_resolved_counter_and_the_kernels_own_clock(golden, monkeypatch, tune):
    # hr_get_step_log / hr_frame_passes_resolved / hr_kernel_times.trace_clock_*: the pipeline's own account of itself
    monkeypatch.setenv("HR_TUNE", tune)
    sc = scenes.triangle_soup(20000, width=320, height=192, bounces=4, passes=64, env=True)
    g = core.create_engine(time_kernels=True)
    sc.apply(g, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    batch = g.pass_batch(sc.options.max_ray_depth)
    assert g.passes_resolved() == 0
    seen = []
    for s in range(3 * batch + 1):
        g.render_pass(sc.options.pass_params(s))
        seen.append(g.passes_resolved())
    assert seen == sorted(seen) and seen[-1] <= 3 * batch + 1          # a host-side counter of enqueued resolves: never ahead of the requests
    g.flush()
    assert g.passes_resolved() == 3 * batch + 1
    log = g.step_log()
    kt = g.kernel_times()
    assert len(log) == kt["trace"][1] == kt["trace_clock"][1] >= sc.options.max_ray_depth + 2
    starts = [r[0] for r in log]
    assert starts[0] == 0.0 and starts == sorted(starts)      # (sorted by the library: with two pipeline groups the records arrive in the order the steps END)
    assert {r[4] for r in log} <= set(range(3))
    assert sum(r[3] for r in log) == 3 * batch + 1                         # every pass was injected by exactly one step
    assert max(r[2] for r in log) <= 3 * batch + 1 and all(r[1] > 0 for r in log)
    # the launch durations by the device clock agree with the HIP events around the same launches (events include the launch's edges)
    assert abs(sum(r[1] for r in log) - kt["trace_clock"][0]) < 1e-3 * max(kt["trace_clock"][0], 1e-3) + 1e-3
    assert 0.5 * kt["trace"][0] < kt["trace_clock"][0] <= 1.05 * kt["trace"][0] + 0.05
    g.clear()
    assert g.passes_resolved() == 0 and g.step_log() == []
    g.close()


def test_memory_budget_bounds_the_batch_and_the_frame_stays_the_same(golden):
    # hr_ctx_desc.memory_budget: fewer passes per pipeline step — first as many as fit with every queue at its longest, more once a full
    # pipeline has shown the real lengths —, same frame; a budget below one pass per step is refused.
    import torch
    sc = scenes.triangle_soup(20000, width=640, height=360, bounces=4, passes=96, env=True)
    free = lambda: torch.cuda.mem_get_info()[0]
    torch.cuda.init()
    ref = core.create_engine()
    sc.apply(ref, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    unlimited = ref.pass_batch(sc.options.max_ray_depth)
    for s in range(96):
        ref.render_pass(sc.options.pass_params(s))
    want = ref.readback().copy()
    ref.close()
    budget = 600 << 20
    f0 = free()
    g = core.create_engine(memory_budget=budget)
    sc.apply(g, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    f1 = free()
    first = g.pass_batch(sc.options.max_ray_depth)
    assert 1 <= first < unlimited
    for s in range(96):
        g.render_pass(sc.options.pass_params(s))
    got = g.readback().copy()
    later = g.pass_batch(sc.options.max_ray_depth)
    held = f1 - free()
    assert got.tobytes() == want.tobytes()          # (the frame never depends on the batch)
    assert first <= later <= unlimited              # the real queue lengths allow at least what the guarantee did
    assert held <= budget * 1.05, (held >> 20, budget >> 20)
    g.close()
    tiny = core.create_engine(memory_budget=8 << 20)
    sc.apply(tiny, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    with pytest.raises(core.EngineError, match="memory_budget"):
        tiny.render_pass(sc.options.pass_params(0))
    tiny.close()


@pytest.mark.parametrize("ovf,kind", [(1, "camera rays"), (2, "closest-hit queue (input)"), (3, "occlusion queue")])
@pytest.mark.parametrize("packets", [0, 1])
def test_a_wrong_queue_bound_is_an_error_not_a_fault(golden, monkeypatch, ovf, kind, packets):
    # Every append to a ray queue compares its slot with the queue's capacity and every reader clamps the counter it reads
    # (hr_render.hip: queueOverflow).  HR_TUNE ovf= hands the kernels HALF (an eighth) of what a bound should be — the mistake that
    # was a memory fault in round 4 —: rays are dropped, the device reports the queue, and the calls that hand work back fail with it.
    monkeypatch.setenv("HR_TUNE", f"ovf={ovf},packets={packets}")
    sc = scenes.triangle_soup(20000, width=320, height=192, bounces=4, passes=64, env=True, room=True)
    g = core.create_engine()
    sc.apply(g, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    for s in range(24):
        g.render_pass(sc.options.pass_params(s))
    with pytest.raises(core.EngineError, match=re.escape("ray queue overflow: " + kind)):
        g.readback()
    with pytest.raises(core.EngineError, match="ray queue overflow"):
        g.synchronize()
    g.clear()                                   # the frame starts afresh, and so does the report
    g.synchronize()
    assert float(np.abs(g.readback()).max()) == 0.0
    g.render_pass(sc.options.pass_params(0))    # ... and the next pass overflows again (the knob is still set)
    with pytest.raises(core.EngineError, match="ray queue overflow"):
        g.readback()
    g.close()
    # without the knob the same render reports nothing
    monkeypatch.setenv("HR_TUNE", f"packets={packets}")
    g = core.create_engine()
    sc.apply(g, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    for s in range(24):
        g.render_pass(sc.options.pass_params(s))
    g.readback(), g.synchronize()
    g.close()


@pytest.mark.parametrize("tune", ["packets=2", "packets=2,punion=101", "packets=1", "packets=0", "packets=1,batch=5", "packets=1,batch=2", "packets=1,corun=2",
                                  "packets=1,corun=2,cblocks=1", "packets=1,corun=0"])
def test_camera_rays_as_packets_where_the_probe_says_so_same_bits_either_way(monkeypatch, golden, tune):
    # hr_core.hip's packet selector: a pass's camera rays walk the tree 64 at a time — the rays of a few neighbouring pixels in the passes
    # injected together (k_raygen_packets) — where a probe finds such a wave's rays visiting nearly the same nodes (union factor below the
    # threshold), and one ray per lane (k_trace) elsewhere.  Forced on (in groups of 16, 4 + 1, 2 passes; the packet kernel beside k_trace
    # on a second stream or in front of it), forced off or chosen, with the default threshold or one nothing can meet: the frame is the
    # oracle's, bit for bit.
    monkeypatch.setenv("HR_TUNE", tune)
    box = scenes.cornell_box(width=256, height=256, bounces=3, passes=40)
    fog = scenes.triangle_soup(150000, width=384, height=256, bounces=3, passes=40, env=True)
    for sc in (box, fog):
        g, o = core.create_engine(), oracle_lib.engine()
        lut = golden["multiscatter_lut"]
        sc.apply(g, lut=lut, tables=host_tables(sc)), sc.apply(o, lut=lut, tables=host_tables(sc))
        n = 3 * g.pass_batch(sc.options.max_ray_depth) + 1  # (the probe of the first batch has reported by the third)
        for s in range(n):
            g.render_pass(sc.options.pass_params(s))
        on, union, _ = g.kernel_times()["camera_packets"]
        if tune.startswith("packets=2"):
            limit = 1.01 if "punion" in tune else 2.2
            assert union > 1.0 and on == (union < limit), (sc.name, on, union)
            if sc is box:
                assert on == ("punion" not in tune), (on, union)  # (a box of 32 large triangles: the union factor is next to 1)
        else:
            assert on == tune.startswith("packets=1")
        n = min(n, 7)
        for s in range(n):
            o.render_pass(sc.options.pass_params(s))
        g.clear()
        for s in range(n):
            g.render_pass(sc.options.pass_params(s))
        assert g.readback().tobytes() == o.readback().tobytes(), (sc.name, tune)
        g.close(), o.close()


def test_packet_selector_follows_the_scene_on_one_context(monkeypatch, golden):
    # a commit makes the selector probe again: with a threshold between the two scenes' union factors the same context goes from packets
    # (a box of large triangles) to one ray per lane (a fog of tiny ones) and back, and renders the oracle's frame each time
    monkeypatch.setenv("HR_TUNE", "punion=115")
    box = scenes.cornell_box(width=256, height=256, bounces=2, passes=40)
    fog = scenes.triangle_soup(150000, width=256, height=256, bounces=2, passes=40, env=True)
    g = core.create_engine()
    lut = golden["multiscatter_lut"]
    for sc, wants_packets in ((box, True), (fog, False), (box, True)):
        g.clear_scene()
        sc.apply(g, lut=lut, tables=host_tables(sc))
        n = 3 * g.pass_batch(sc.options.max_ray_depth) + 1
        for s in range(n):
            g.render_pass(sc.options.pass_params(s))
        on, union, _ = g.kernel_times()["camera_packets"]
        assert on == wants_packets and (union < 1.15) == wants_packets, (sc.name, on, union)
        o = oracle_lib.engine()
        sc.apply(o, lut=lut, tables=host_tables(sc))
        g.clear()
        for s in range(4):
            g.render_pass(sc.options.pass_params(s)), o.render_pass(sc.options.pass_params(s))
        assert g.readback().tobytes() == o.readback().tobytes(), sc.name
        o.close()
    g.close()


@pytest.mark.parametrize("tune", ["packets=1,corun=0", "packets=1,corun=2", "packets=0"])
def test_passes_of_one_batch_may_differ_in_everything(monkeypatch, golden, tune):
    # The passes a packet spans usually differ in their sample index only (then their parameters come through the scalar cache); nothing in the
    # API says they must: here every pass of a batch has its own field of view, focus, aperture and depth-of-field sample — each lane
    # reads its own pass's block — and the frame is the oracle's.
    monkeypatch.setenv("HR_TUNE", tune)
    sc = scenes.multi_material(width=160, height=96, bounces=3, passes=24)
    g, o = core.create_engine(), oracle_lib.engine()
    lut = golden["multiscatter_lut"]
    sc.apply(g, lut=lut, tables=host_tables(sc)), sc.apply(o, lut=lut, tables=host_tables(sc))
    for s in range(19):
        sc.options.focal_length = 35.0 + 3.0 * (s % 5)
        sc.options.fstop = 2.0 + (s % 3)
        sc.options.focus_distance = 2.0 + 0.25 * (s % 4)
        pp = sc.options.pass_params(s)
        g.render_pass(pp), o.render_pass(pp)
    assert g.readback().tobytes() == o.readback().tobytes()
    g.close(), o.close()


@pytest.mark.parametrize("tune,expect", [("packets=0", {(1920, 1080): 12, (3840, 2160): 3, (256, 256): 32, (2560, 1440): 7}),
                                         ("packets=1", {(1920, 1080): 16, (3840, 2160): 4, (256, 256): 32, (2560, 1440): 8})])
def test_batch_is_the_neighbouring_power_of_two_while_packets_are_in_use(monkeypatch, tune, expect):
    # hr_frame_pass_batch: 12 x 1080p pixels' worth of camera rays per macro step; a packet spans 2^k passes, so with packets the batch is the
    # power of two next to that figure (3/4 of the way up rounds up: 12 -> 16, 3 -> 4, 7 -> 8; 5 would go to 4)
    monkeypatch.setenv("HR_TUNE", tune)
    g = core.create_engine()
    for (w, h), want in expect.items():
        g.resize(w, h)
        assert g.pass_batch(3) == want, (tune, w, h, g.pass_batch(3))
    g.close()


def test_frames_do_not_depend_on_how_camera_rays_travel(monkeypatch):
    # One ray per lane, packets in front of k_trace, packets beside it: the same frame, bit for bit, over a run long enough that float32
    # Moeller-Trumbore's phantom hits on sliver triangles occur (a ray that passes a sliver at a distance can be accepted: once in ~10^9
    # rays on the benchmark soup).  A packet lane therefore tests a triangle only when its own ray enters the triangle's box, as k_trace
    # does for that ray; before that rule a 600-pass run at 1080p differed in two pixels (profiles/r4x_soak_digests.txt).
    frames = []
    for tune in ("packets=0", "packets=1,corun=0", "packets=1,corun=2"):
        monkeypatch.setenv("HR_TUNE", tune)
        sc = scenes.triangle_soup(1_000_000, width=1920, height=1080, bounces=2, passes=160, env=True)
        g = core.create_engine()
        sc.apply(g)
        for s in range(160):
            g.render_pass(sc.options.pass_params(s))
        frames.append(g.readback().copy())
        g.close()
    assert frames[0].tobytes() == frames[1].tobytes() == frames[2].tobytes()


def test_large_scene_3m_triangles(golden):
    # maximum-size end of the range (tools/big_scene_check.py goes to 30 M): device LBVH + collapse of 3 M triangles, hits against
    # the oracle's own tree and a render, bit for bit
    sc = scenes.triangle_soup(3_000_000, width=48, height=48, bounces=3, passes=4, env=True)
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"], tables=host_tables(sc))
    assert g.scene_info().n_triangles == 3_000_000
    rng = np.random.default_rng(8)
    org = rng.uniform(-1.1, 1.1, (8000, 3)).astype(np.float32)
    d = rng.normal(size=(8000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    assert g.debug_trace(org, d).tobytes() == o.debug_trace(org, d).tobytes()
    for s in range(2):
        g.render_pass(sc.options.pass_params(s)), o.render_pass(sc.options.pass_params(s))
    assert_parity(g.readback(), o.readback(), "3 M triangles")
