"""Known-answer tests (KATs) that pin the CPU oracle's restatement of the RLSL shaders.

The reference runs its shaders inside the closed OpenRL runtime and holds no numerical fixtures at that
boundary, and the HIP shaders and the oracle are two transcriptions of the same text — so a
common transcription error would be invisible to every GPU-vs-oracle test.  Each test here checks one piece
of the restatement against an answer derived INDEPENDENTLY of the oracle's code: the published formula the
RLSL cites (Fresnel equations, Walter et al. 2007 GGX, Heitz 2018 visible-normal sampling, Beer-Lambert),
evaluated in float64 numpy, or a closed-form radiometric result for a scene built so that only the piece
under test contributes.  RLSL lines are relative to /root/reference/Resources/shaders.

Function-level probes (`ora_kat_*`, oracle/oracle_shade.cpp) evaluate ONE restated function on given inputs;
scene-level tests render through the whole path.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib
from heatray_amd import _ffi as ffi
from heatray_amd import host, scenes

F = np.float32
f3 = C.c_float * 3


def lib():
    L = oracle_lib.load()
    L.ora_kat_scalar.restype = C.c_float
    L.ora_kat_scalar.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    return L


def scalar(which, a, b=0.0, c=0.0):
    return float(lib().ora_kat_scalar(which, float(a), float(b), float(c)))


def render(sc, passes, **kw):
    eng = oracle_lib.engine(**kw)
    sc.apply(eng)
    for s in range(passes):
        eng.render_pass(sc.options.pass_params(s))
    return eng.readback(), eng


# ------------------------------------------------------------------------------------------------ Fresnel
def fresnel_unpolarised(n1, n2, cos_i):
    """Fresnel equations for a dielectric interface, unpolarised light (Born & Wolf §1.5.2), float64."""
    sin_t = n1 / n2 * math.sqrt(max(0.0, 1.0 - cos_i * cos_i))
    if sin_t >= 1.0:
        return 1.0
    cos_t = math.sqrt(1.0 - sin_t * sin_t)
    rs = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t)
    rp = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t)
    return 0.5 * (rs * rs + rp * rp)


@pytest.mark.parametrize("n", [1.33, 1.5, 1.57, 2.4])
def test_fresnel_matches_the_fresnel_equations(n):
    # brdfs.rlsl:59-71 F_Fresnel(eta = nIn / nOut, cosThetaI); glass.rlsl:213,223 calls it with eta = 1/ior outside, ior inside
    for cos_i in np.linspace(0.02, 1.0, 50):
        assert scalar(0, 1.0 / n, cos_i) == pytest.approx(fresnel_unpolarised(1.0, n, cos_i), rel=2e-4, abs=2e-6)
        assert scalar(0, n, cos_i) == pytest.approx(fresnel_unpolarised(n, 1.0, cos_i), rel=2e-4, abs=2e-5)
    # normal incidence: ((n - 1) / (n + 1))^2 from either side
    r0 = ((n - 1.0) / (n + 1.0)) ** 2
    assert scalar(0, 1.0 / n, 1.0) == pytest.approx(r0, rel=1e-5)
    assert scalar(0, n, 1.0) == pytest.approx(r0, rel=1e-5)
    # total internal reflection beyond the critical angle: F = 1, so glass.rlsl:234 never refracts there
    crit = math.sqrt(1.0 - 1.0 / (n * n))
    assert scalar(0, n, crit * 0.98) == 1.0
    assert scalar(0, n, min(1.0, crit * 1.05)) < 1.0


def test_schlick_fresnel_end_points():
    # brdfs.rlsl:46-57: F0 at normal incidence, 1 at grazing, (1 - c)^5 interpolation
    for f0 in (0.0, 0.04, 0.5, 1.0):
        assert scalar(1, f0, 1.0) == pytest.approx(f0, abs=1e-7)
        assert scalar(1, f0, 0.0) == pytest.approx(1.0, abs=1e-7)
        assert scalar(1, f0, 0.5) == pytest.approx(f0 + (1 - f0) * 0.5 ** 5, rel=1e-6)


# ------------------------------------------------------------------------------------------------ GGX terms
def ggx_d(cos_h, a):
    """Walter et al. 2007 eq. 33 (GGX / Trowbridge-Reitz), alpha = roughness^2; the RLSL guards the denominator with
    greaterThanZero = max(1e-5, .) (utility.rlsl:153-156), which caps the peak of very smooth lobes (alpha^4 < 1e-5)."""
    return a * a / (math.pi * max(1e-5, (cos_h * cos_h * (a * a - 1.0) + 1.0) ** 2))


def smith_g1(cos_v, a):
    """Walter et al. 2007 eq. 34: 2 / (1 + sqrt(1 + alpha^2 tan^2))."""
    t2 = (1.0 - cos_v * cos_v) / (cos_v * cos_v)
    return 2.0 / (1.0 + math.sqrt(1.0 + a * a * t2))


@pytest.mark.parametrize("alpha", [0.01, 0.09, 0.25, 1.0])
def test_ggx_d_and_g1_closed_forms(alpha):
    for c in np.linspace(0.05, 1.0, 40):
        assert scalar(2, c, alpha) == pytest.approx(ggx_d(c, alpha), rel=3e-4 if alpha > 0.05 else 2e-2)  # brdfs.rlsl:73-78
        assert scalar(3, c, alpha) == pytest.approx(smith_g1(c, alpha), rel=3e-4)   # brdfs.rlsl:88-93
        assert scalar(4, c, 0.7, alpha) == pytest.approx(smith_g1(c, alpha) * smith_g1(0.7, alpha), rel=5e-4)  # :95-98 (separable)


@pytest.mark.parametrize("alpha", [0.09, 0.25, 1.0])
def test_ggx_d_is_a_normalised_distribution(alpha):
    # integral of D(h) (n.h) over the hemisphere = 1; substitute x = cos(theta): 2 pi * int_0^1 D(x) x dx
    x = (np.arange(200000) + 0.5) / 200000
    d = np.array([scalar(2, xi, alpha) for xi in x[::200]])  # probe on a coarse grid, closed form checked above
    dense = alpha * alpha / (np.pi * (x * x * (alpha * alpha - 1.0) + 1.0) ** 2)
    assert np.allclose(d, dense[::200], rtol=3e-4)
    assert 2.0 * math.pi * float(np.mean(dense * x)) == pytest.approx(1.0, rel=1e-3)


def vndf_samples(v, alpha, n=256):
    L = lib()
    out = f3()
    vv = f3(*v)
    u = (np.arange(n) + 0.5) / n
    hs = np.empty((n, n, 3))
    for i, u1 in enumerate(u):
        for j, u2 in enumerate(u):
            L.ora_kat_sample_visible_ggx(vv, C.c_float(u1), C.c_float(u2), C.c_float(alpha), out)
            hs[i, j] = out[:]
    return hs.reshape(-1, 3)


def single_scatter_albedo(mu, alpha, f0=1.0, n=1024):
    """Directional albedo of the single-scattering GGX microfacet BRDF D F G2 / (4 mu_i mu_o) with the separable Smith G2
    the reference uses and Schlick Fresnel: int f cos dw_o, by quadrature in half-vector space with the density D(h)(n.h)
    (Walter 2007 eqs. 35-36), float64.  Independent of the oracle's visible-normal sampler."""
    u = (np.arange(n) + 0.5) / n
    u1, u2 = np.meshgrid(u, u, indexing="ij")
    cos2 = (1.0 - u1) / (1.0 + (alpha * alpha - 1.0) * u1)
    cos_h = np.sqrt(cos2)
    sin_h = np.sqrt(1.0 - cos2)
    phi = 2.0 * np.pi * u2
    h = np.stack([sin_h * np.cos(phi), sin_h * np.sin(phi), cos_h], axis=-1)  # z-up
    v = np.array([math.sqrt(1.0 - mu * mu), 0.0, mu])
    vh = h @ v
    l = 2.0 * vh[..., None] * h - v
    mu_l = l[..., 2]
    ok = (vh > 0) & (mu_l > 0)
    g1 = lambda c: 2.0 / (1.0 + np.sqrt(1.0 + alpha * alpha * (1.0 - c * c) / np.maximum(c * c, 1e-30)))
    fr = f0 + (1.0 - f0) * (1.0 - np.clip(vh, 0, 1)) ** 5
    # f cos_l dw_l with dw_l = 4 (v.h) dw_h and pdf(h) = D cos_h:  F G2 (v.h) / (mu cos_h)
    w = np.where(ok, fr * g1(mu) * g1(np.clip(mu_l, 1e-9, 1)) * vh / (mu * cos_h), 0.0)
    return float(w.mean())


@pytest.mark.parametrize("alpha,mu", [(0.25, 0.9), (0.25, 0.4), (1.0, 0.7), (0.04, 0.6)])
def test_visible_normal_sampling_reproduces_the_brdf_integral(alpha, mu):
    # utility.rlsl:109-139 sampleVisibleGGX + microfacet.rlsl:100-151: with visible-normal sampling the estimator of the
    # specular lobe is F G2 / G1 (F = 1 here); its mean must equal the lobe's directional albedo computed by an unrelated
    # quadrature (half-vector sampling with pdf D cos).  Y-up local space: v = (sin, cos, 0).
    v = (math.sqrt(1.0 - mu * mu), mu, 0.0)
    hs = vndf_samples(v, alpha, n=192)
    assert np.allclose(np.linalg.norm(hs, axis=1), 1.0, atol=1e-5)
    assert (hs[:, 1] >= -1e-6).all()  # microfacet normals in the upper hemisphere
    vv = np.array(v)
    vh = np.clip(hs @ vv, 0, 1)
    o = 2.0 * vh[:, None] * hs - vv
    mu_o = o[:, 1]
    g1 = lambda c: 2.0 / (1.0 + np.sqrt(1.0 + alpha * alpha * (1.0 - c * c) / np.maximum(c * c, 1e-30)))
    est = np.where(mu_o > 0, g1(np.clip(mu_o, 1e-9, 1)), 0.0)  # G2 / G1(v) = G1(o)
    assert float(est.mean()) == pytest.approx(single_scatter_albedo(mu, alpha), rel=4e-3)
    # and the samples follow the visible-normal density D_v(h) = G1(v) max(0, v.h) D(h) / mu (Heitz 2018 eq. 3):
    # compare the histogram of n.h with that density integrated over azimuth
    if alpha < 0.2:
        return  # a lobe this narrow needs a finer azimuth / polar grid than is worth it; the integral above covers it
    edges = np.linspace(0.0, 1.0, 11)
    hist, _ = np.histogram(hs[:, 1], bins=edges)
    m = 600
    ct = (np.arange(m) + 0.5) / m
    ph = (np.arange(m) + 0.5) / m * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    ST = np.sqrt(1 - CT * CT)
    vdoth = np.maximum(0.0, ST * np.cos(PH) * v[0] + CT * v[1])
    dens = g1(mu) * vdoth * (alpha * alpha / (np.pi * (CT * CT * (alpha * alpha - 1) + 1) ** 2)) / mu  # per solid angle; d(cos) dphi
    expect = np.array([dens[(ct >= a) & (ct < b)].sum() for a, b in zip(edges[:-1], edges[1:])]) * (1.0 / m) * (2 * np.pi / m)
    assert expect.sum() == pytest.approx(1.0, rel=2e-2)
    assert np.allclose(hist / hist.sum(), expect / expect.sum(), atol=6e-3)


def test_cosine_weighted_sample_density():
    # utility.rlsl:64-75: pdf = cos / pi about +y: E[y] = 2/3, E[y^2] = 1/2, E[x] = E[z] = 0
    L = lib()
    out = f3()
    n = 128
    u = (np.arange(n) + 0.5) / n
    d = np.empty((n, n, 3))
    for i, a in enumerate(u):
        for j, b in enumerate(u):
            L.ora_kat_cosine_sample(C.c_float(a), C.c_float(b), out)
            d[i, j] = out[:]
    d = d.reshape(-1, 3)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    assert d[:, 1].mean() == pytest.approx(2.0 / 3.0, rel=2e-3)
    assert (d[:, 1] ** 2).mean() == pytest.approx(0.5, rel=2e-3)
    assert abs(d[:, 0].mean()) < 2e-3 and abs(d[:, 2].mean()) < 2e-3


def test_orthonormal_frame():
    # utility.rlsl:43-60: a right- or left-handed orthonormal basis whose second column is N, for any N
    L = lib()
    out = (C.c_float * 9)()
    rng = np.random.default_rng(3)
    ns = rng.normal(size=(200, 3))
    ns = np.vstack([ns / np.linalg.norm(ns, axis=1, keepdims=True), [[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, 0, -1]]])
    for n in ns:
        L.ora_kat_frame(f3(*n), out)
        m = np.array(out[:]).reshape(3, 3)  # rows = columns c0 c1 c2
        assert np.allclose(m[1], n, atol=1e-6)
        assert np.allclose(m @ m.T, np.eye(3), atol=2e-5), (n, m)


def test_refract_obeys_snell():
    # GLSL refract(I, N, eta) as used by glass.rlsl:235 with the microfacet normal
    L = lib()
    out = f3()
    for n in (1.33, 1.5):
        for ang in (0.0, 0.3, 0.9, 1.3):
            i = np.array([math.sin(ang), -math.cos(ang), 0.0])
            L.ora_kat_refract(f3(*i), f3(0, 1, 0), C.c_float(1.0 / n), out)
            t = np.array(out[:])
            assert np.linalg.norm(t) == pytest.approx(1.0, abs=1e-5)
            assert math.sin(ang) == pytest.approx(n * abs(t[0]), abs=2e-5)  # n1 sin(i) = n2 sin(t)
            assert t[1] < 0 and t[0] * i[0] >= 0


# ------------------------------------------------------------------------------------------------ lobes
def test_pbr_lobe_probabilities():
    # physicallyBased.rlsl:206-228: Cdiff = base (1 - metal) (1 - ccScale), Cspec = mix(F0, base, metal) (1 - ccScale),
    # ccScale = Schlick(0.04, Ncc.V) * clearCoat; probabilities proportional to luminosity(Cdiff) : luminosity(Cspec) : ccScale
    # with luminosity = dot(c, (0.33, 0.59, 0.11)) (utility.rlsl:163-166) — and they sum to one.
    L = lib()
    out = (C.c_float * 10)()
    rng = np.random.default_rng(11)
    for _ in range(100):
        base = rng.uniform(0.05, 1.0, 3)
        metal, f0, cc, cos = rng.uniform(0, 1), rng.uniform(0, 0.08), rng.uniform(0, 0.2), rng.uniform(0.05, 1)
        L.ora_kat_pbr_lobes(f3(*base), C.c_float(metal), C.c_float(f0), C.c_float(cc), C.c_float(cos), out)
        o = np.array(out[:], dtype=np.float64)
        scale = (0.04 + 0.96 * (1 - cos) ** 5) * cc
        cdiff = base * (1 - metal) * (1 - scale)
        cspec = (f0 * (1 - metal) + base * metal) * (1 - scale)
        lum = lambda c: float(np.dot(c, [0.33, 0.59, 0.11]))
        tot = lum(cdiff) + lum(cspec) + scale
        assert np.allclose(o[0:3], cdiff, rtol=2e-5) and np.allclose(o[3:6], cspec, rtol=2e-5, atol=1e-8)
        assert o[6] == pytest.approx(scale, rel=2e-5)
        assert np.allclose(o[7:10], [lum(cdiff) / tot, lum(cspec) / tot, scale / tot], rtol=5e-5)
        assert o[7] + o[8] + o[9] == pytest.approx(1.0, abs=1e-6)
    # without clearcoat the third lobe is never selected
    L.ora_kat_pbr_lobes(f3(0.5, 0.5, 0.5), C.c_float(0.3), C.c_float(0.04), C.c_float(0.0), C.c_float(0.5), out)
    assert out[9] == 0.0 and out[6] == 0.0


# ------------------------------------------------------------------------------------------------ lights
def light_engine(rig, env_rgb=None):
    eng = oracle_lib.engine()
    eng.resize(8, 8)
    if env_rgb is not None:
        tid = eng.create_texture(np.asarray(env_rgb, dtype=F).reshape(1, 1, 3), wrap=ffi.HR_WRAP_REPEAT)
        rig.set_environment(tid, rig.env_exposure_compensation, rig.env_theta_rotation)
    eng.set_lights(rig.bake())
    return eng


def light_sample(eng, n, p, xi):
    oi, of = (C.c_int * 3)(), (C.c_float * 5)()
    lib().ora_kat_light_sample(eng._ctx, f3(*n), f3(*p), C.c_float(xi), oi, of)
    return dict(type=oi[0], kind=oi[1], index=oi[2], prob=of[0], max_dist=of[1], dir=np.array(of[2:5]))


def light_shader(eng, kind, index, direction, weight=(1, 1, 1), t=1.0, extra_t=0.0, clamp=1e30):
    out = f3()
    lib().ora_kat_light_shader(eng._ctx, kind, index, f3(*direction), f3(*weight), C.c_float(t), C.c_float(extra_t), C.c_float(clamp), out)
    return np.array(out[:])


def test_light_pick_probabilities():
    # lightSampling.rlsl:11-161: weight_i = saturate(N.L_i) luminosity(color_i) (spot: x cone gate), environment = 50 x exposure;
    # probabilities = weights / sum; the CDF is walked directional -> point -> spot -> environment.
    rig = host.LightRig()
    rig.add_directional(color=(1, 1, 1), illuminance=683.0, phi=0.0, theta=math.pi / 2)       # straight up: N.L = 1
    rig.add_directional(color=(1, 0, 0), illuminance=683.0 * 3, phi=0.3, theta=-0.5)          # below the horizon of N = +y: weight 0
    rig.add_point((0.0, 2.0, 0.0), color=(0, 1, 0), luminous_intensity=683.0 / (4 * math.pi))  # colour (0,1,0) W
    rig.env_exposure_compensation = -5.0                                                        # exposure 2^-5: env weight 50/32
    eng = light_engine(rig, env_rgb=(1, 1, 1))
    L = rig.bake()
    n, p = (0.0, 1.0, 0.0), (0.0, 0.0, 0.0)
    d0 = np.array(L.directional_directions[0][:])
    d1 = np.array(L.directional_directions[1][:])
    assert d0[1] == pytest.approx(1.0, abs=1e-6) and d1[1] < 0
    lum = lambda c: 0.33 * c[0] + 0.59 * c[1] + 0.11 * c[2]
    w = [max(0.0, d0[1]) * lum(L.directional_colors[0][:]), 0.0, 1.0 * lum(L.point_colors[0][:]), 50.0 * 2.0 ** -5]
    tot = sum(w)
    ps = [x / tot for x in w]
    s = light_sample(eng, n, p, 0.5 * ps[0])
    assert (s["type"], s["index"]) == (1, 0) and s["prob"] == pytest.approx(ps[0], rel=1e-5) and np.allclose(s["dir"], d0, atol=1e-6)
    s = light_sample(eng, n, p, ps[0] + 0.5 * ps[2])  # the zero-weight directional light 1 is skipped
    assert (s["type"], s["index"]) == (2, 0) and s["prob"] == pytest.approx(ps[2], rel=1e-5)
    assert s["max_dist"] == pytest.approx(2.0, rel=1e-6) and np.allclose(s["dir"], (0, 1, 0), atol=1e-6)
    s = light_sample(eng, n, p, ps[0] + ps[2] + 0.5 * ps[3])
    assert s["type"] == 4 and s["prob"] == pytest.approx(ps[3], rel=1e-5)
    assert ps[0] + ps[2] + ps[3] == pytest.approx(1.0)


def test_light_shaders_directional_point():
    rig = host.LightRig()
    rig.add_directional(color=(1.0, 0.5, 0.25), illuminance=683.0 * 2.0)
    rig.add_point((0, 0, 0), color=(0.2, 0.4, 0.6), luminous_intensity=683.0)
    eng = light_engine(rig)
    # directionalLight.rlsl:20-26 weight * colour, colour = rgb * illuminance / 683 (DirectionalLight.cpp:42-51)
    assert np.allclose(light_shader(eng, 2, 0, (0, 1, 0), weight=(0.5, 1, 2)), np.array([1.0, 0.5, 0.25]) * 2.0 * np.array([0.5, 1, 2]), rtol=1e-6)
    # pointLight.rlsl:20-29 weight * colour / (t + extraT)^2, colour = rgb * (I / 683) * 4 pi (PointLight.cpp:41-50)
    v = light_shader(eng, 3, 0, (0, 1, 0), t=2.0, extra_t=1.0)
    assert np.allclose(v, np.array([0.2, 0.4, 0.6]) * 4 * math.pi / 9.0, rtol=1e-6)
    # accumulator.rlsl:12-28: min(maxChannelValue, colour) per channel
    assert np.allclose(light_shader(eng, 2, 0, (0, 1, 0), weight=(10, 10, 1), clamp=math.pi), [math.pi, math.pi, 0.5], rtol=1e-6)


def test_spot_cone():
    # SpotLight.cpp:44-69: colour = rgb (I / 683) pi, angles stored as cosines (inner, outer), direction = where the light points.
    # spotLight.rlsl:20-36: value = weight colour / t^2 (1 - smoothstep(cosInner, cosOuter, cos)), nothing behind the light;
    # lightSampling.rlsl:52-70: pick weight additionally gated to zero outside the outer cone.
    inner, outer = math.radians(15.0), math.radians(35.0)
    rig = host.LightRig()
    rig.add_spot((0.0, 3.0, 0.0), color=(1, 1, 1), luminous_intensity=683.0, phi=0.0, theta=math.pi / 2, inner_angle=inner, outer_angle=outer)
    eng = light_engine(rig)
    L = rig.bake()
    sd = np.array(L.spot_directions[0][:])
    assert np.allclose(sd, (0, -1, 0), atol=1e-6)  # theta = pi/2: shining straight down
    assert np.allclose(L.spot_angles[0][:], (math.cos(inner), math.cos(outer)), rtol=1e-6)
    col = math.pi
    for ang in np.radians([0.0, 5.0, 14.0, 20.0, 25.0, 30.0, 34.0, 36.0, 60.0, 89.0]):
        # a surface point below the light, seen from the light under angle `ang` from its axis
        p = np.array([3.0 * math.tan(ang), 0.0, 0.0])
        to_light = np.array([0.0, 3.0, 0.0]) - p
        t = float(np.linalg.norm(to_light))
        to_light /= t
        c = math.cos(ang)
        x = min(max((c - math.cos(inner)) / (math.cos(outer) - math.cos(inner)), 0.0), 1.0)
        falloff = 1.0 - x * x * (3.0 - 2.0 * x)
        v = light_shader(eng, 4, 0, to_light, t=t)
        assert np.allclose(v, col / (t * t) * falloff, rtol=2e-4, atol=1e-7), (math.degrees(ang), v)
        if ang <= inner:
            assert np.allclose(v, col / (t * t), rtol=1e-5)   # inside the inner cone: full intensity
        if ang >= outer:
            assert (v == 0).all()                              # outside the outer cone: nothing
        s = light_sample(eng, (0, 1, 0), p, 0.5)
        if ang < outer:
            assert s["type"] == 3 and s["prob"] == pytest.approx(1.0, rel=1e-5) and s["max_dist"] == pytest.approx(t, rel=1e-5)
        else:
            assert s["type"] == 4 and s["prob"] == 0.0        # no light has weight: falls through to the (absent) environment
    # a ray arriving from behind the light (rayAngle < 0) gets nothing
    assert (light_shader(eng, 4, 0, (0, -1, 0), t=1.0) == 0).all()


# ------------------------------------------------------------------------------------------------ environment
def latlong_uv(d, rot=0.0):
    """environmentLight.rlsl:19-34 written out: where in the lat/long map (u to the right, t upwards, both in [0,1]) a
    direction lands.  Independent statement of the convention: -z is the map centre, +x a quarter turn to the right,
    +y the top row."""
    az = math.atan2(d[0], -d[2]) + rot
    if az > 2 * math.pi:
        az -= 2 * math.pi
    el = math.atan2(d[1], math.hypot(d[0], d[2]))
    return az / (2 * math.pi) + 0.5, el / math.pi + 0.5


def test_environment_latlong_orientation():
    # a 16 x 8 map whose texels are all different: look along the six axes and a few oblique directions
    w, h = 16, 8
    env = np.zeros((h, w, 3), dtype=F)
    env[..., 0] = np.arange(w)[None, :] + 1
    env[..., 1] = np.arange(h)[:, None] + 1
    env[..., 2] = 0.5
    rig = host.LightRig()
    eng = oracle_lib.engine()
    eng.resize(8, 8)
    tid = eng.create_texture(env, wrap=ffi.HR_WRAP_REPEAT, filter=ffi.HR_FILTER_NEAREST)
    for rot in (0.0, 0.7, 2.5):
        rig.set_environment(tid, 1.0, rot)  # exposure compensation 1 -> factor 2
        eng.set_lights(rig.bake())
        for d in [(0, 0, -1), (1, 0, 0), (0, 0, 1), (-1, 0, 0), (0.3, 0.8, -0.5), (0.3, -0.8, 0.5), (-0.7, 0.1, 0.7), (0.01, 0.999, 0.0),
                  (0.0, -0.999, 0.02)]:
            d = np.array(d, dtype=np.float64)
            d /= np.linalg.norm(d)
            u, t = latlong_uv(d, rot)
            col = int(math.floor((u % 1.0) * w)) % w
            row = min(h - 1, max(0, int(math.floor(t * h))))  # row 0 = bottom of the map = looking down
            v = light_shader(eng, 1, 0, d)
            assert np.allclose(v, env[row, col] * 2.0, rtol=1e-6), (rot, d, v, env[row, col])


def test_environment_seen_by_the_camera():
    # whole path, empty scene: a primary ray that misses runs the environment shader (perspective.rlsl:91 defaultPrimitive).
    # A camera at the origin looking along -z sees the map centre, yawed by phi it sees the column a turn of phi away.
    w, h = 64, 32
    env = np.zeros((h, w, 3), dtype=F)
    env[..., 0] = np.arange(w)[None, :] + 1
    env[..., 1] = np.arange(h)[:, None] + 1
    sc = scenes.Scene("sky", width=9, height=9, use_multiscatter_lut=False)
    sc.env_pixels = env
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 2, 1.0, host.FSTOP_DISABLED, 2000.0
    o.max_channel_value = 1e6  # accumulator.rlsl:12-28 would clamp the coded texel values at pi
    for phi, theta in [(0.0, 0.0), (math.pi / 2, 0.0), (-math.pi / 2, 0.0), (0.4, 0.6), (2.0, -0.9)]:
        o.view_matrix = host.orbit_view_matrix(0.0, phi, theta)
        fwd = -np.asarray(o.view_matrix, dtype=np.float64)[:3, 2]  # the camera looks along its -z axis
        img, _ = render(sc, 1)
        rgb = img[4, 4, :3] / img[4, 4, 3]
        u, t = latlong_uv(fwd)
        # bilinear filtering: the texel centre nearest to (u, t) dominates; compare with the bilinear value
        x, y = (u % 1.0) * w - 0.5, t * h - 0.5
        x0, y0 = math.floor(x), math.floor(y)
        fx, fy = x - x0, y - y0
        tx = lambda i: env[min(max(int(y0) + i[1], 0), h - 1), (int(x0) + i[0]) % w].astype(np.float64)
        expect = (tx((0, 0)) * (1 - fx) + tx((1, 0)) * fx) * (1 - fy) + (tx((0, 1)) * (1 - fx) + tx((1, 1)) * fx) * fy
        assert np.allclose(rgb[:2], expect[:2], atol=0.05), (phi, theta, rgb, expect)
    # looking along -z: centre column, middle row
    o.view_matrix = host.orbit_view_matrix(0.0, 0.0, 0.0)
    img, _ = render(sc, 1)
    assert img[4, 4, 0] / img[4, 4, 3] == pytest.approx(w / 2 + 0.5, abs=0.05)


# ------------------------------------------------------------------------------------------------ glass
def glass_slab_scene(base, density, thickness, ior, depth):
    """A slab of glass filling the view, seen at normal incidence in a uniform environment of radiance 1."""
    sc = scenes.Scene("slab", width=24, height=24, use_multiscatter_lut=False)
    s = 50.0
    front = scenes._quad((-s, -s, 0), (s, -s, 0), (s, s, 0), (-s, s, 0))                       # normal +z, towards the camera
    back = scenes._quad((-s, -s, -thickness), (-s, s, -thickness), (s, s, -thickness), (s, -s, -thickness))  # normal -z
    p, n, i = scenes._merge([front, back])
    sc.materials[0] = host.bake_glass(base_color=base, roughness=0.0, ior=ior, density=density)  # roughness clamps to 0.01
    sc.meshes.append(scenes.MeshData(p, n, i, material_id=0))
    sc.env_pixels = np.ones((1, 1, 3), dtype=F)
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = depth, 1.0, host.FSTOP_DISABLED, 1000.0
    o.view_matrix = host.orbit_view_matrix(5.0, 0.0, 0.0)
    o.max_render_passes = 256
    return sc


def test_glass_slab_reflectance_and_beer_lambert():
    # glass.rlsl:138-280.  At normal incidence on (almost) smooth glass:
    #   with probability F = ((n-1)/(n+1))^2 the path reflects: NEE towards the environment with weight baseColor (:47-81);
    #   otherwise it refracts in (weight x baseColor), is attenuated by exp(-(1 - baseColor) density d) inside (beersLaw :131-136),
    #   always refracts out (weight x baseColor again, :227-231) and sees the environment.
    n, d, rho = 1.5, 0.8, 1.7
    base = np.array([0.9, 0.6, 0.3])
    r0 = ((n - 1) / (n + 1)) ** 2
    # the reflect / refract choice reads one coordinate of a 256-point stratified sequence per pixel (one point per 1/256):
    # the reflected fraction of a pixel is k/256 with k = 10 or 11 around F = 0.04; 16 differently scrambled sequences average it
    passes = 256
    # depth 0: transmission is cut (:244 depth < maxRayDepth), only the reflection NEE contributes -> measures F
    img0, _ = render(glass_slab_scene(base, rho, d, n, depth=0), passes)
    rgb0 = (img0[..., :3] / img0[..., 3:4]).reshape(-1, 3).mean(axis=0)
    assert np.allclose(rgb0, r0 * base, rtol=0.04), (rgb0, r0 * base)
    # full depth: plus the transmitted path
    img, eng = render(glass_slab_scene(base, rho, d, n, depth=6), passes)
    rgb = (img[..., :3] / img[..., 3:4]).reshape(-1, 3).mean(axis=0)
    expect_t = base * np.exp(-(1 - base) * rho * d) * base
    refl = float((rgb0 / base).mean())  # the reflected fraction actually drawn
    assert np.allclose(rgb - rgb0, (1 - refl) * expect_t, rtol=3e-3), (rgb - rgb0, (1 - refl) * expect_t)
    # per sample the value is one of the two outcomes, never a blend: reflected samples carry baseColor, transmitted ones the
    # Beer-Lambert product (first pass only, a single sample per pixel)
    one, _ = render(glass_slab_scene(base, rho, d, n, depth=6), 1)
    v = one[..., 0].reshape(-1)
    t_val = float(base[0] * math.exp(-(1 - base[0]) * rho * d) * base[0])
    is_t, is_r = np.isclose(v, t_val, rtol=2e-3), np.isclose(v, base[0], rtol=2e-3)
    assert (is_t | is_r).all()
    assert is_r.mean() == pytest.approx(r0, abs=0.02)
    # white glass does not absorb: the two outcomes add up to the environment exactly (energy conservation of the split)
    imgw, _ = render(glass_slab_scene((1, 1, 1), rho, d, n, depth=6), 8)
    assert np.allclose(imgw[..., :3] / imgw[..., 3:4], 1.0, atol=2e-3)


# ------------------------------------------------------------------------------------------------ furnace
def plane_furnace(material, mu, use_lut, env=0.8, size=48):
    """A large plane seen under cos(view angle) = mu through a very long lens, uniform environment."""
    sc = scenes.Scene("plane_furnace", width=size, height=size, use_multiscatter_lut=use_lut)
    p, n, uv, i = scenes.plane_strip(2000, 2000)
    sc.materials[0] = material
    sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
    sc.env_pixels = np.full((1, 1, 3), env, dtype=F)
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 1, 1.0, host.FSTOP_DISABLED, 4000.0
    o.view_matrix = host.orbit_view_matrix(10.0, 0.3, math.asin(mu))  # elevation theta: N.V = sin(theta)
    o.max_render_passes = 64
    return sc


@pytest.mark.parametrize("roughness", [0.2, 0.5, 1.0])
def test_white_metal_furnace_and_multiscatter_lut(roughness):
    # microfacet.rlsl:100-151 + :17-23.  White metal (Cspec = 1, F = 1) in a uniform environment E: the only contribution is the
    # NEE ray towards the environment, an estimator of E x (directional albedo of the single-scatter GGX lobe); with the
    # multiscatter LUT (value (1 - A) / A, MultiScatterUtil.cpp:49-80, pinned by the reference's shipped TIFF) the factor
    # 1 + Cspec LUT = 1 / A restores the missing energy: the plane looks exactly like the environment.
    mu, env = 0.8, 0.8
    mat = lambda: host.bake_pbr(base_color=(1, 1, 1), roughness=roughness, metallic=1.0, specular_f0=0.5)
    alpha = max(roughness, 0.01) ** 2
    a_ss = single_scatter_albedo(mu, alpha)
    img, _ = render(plane_furnace(mat(), mu, use_lut=False, env=env), 32)
    got = float((img[..., :3] / img[..., 3:4]).mean()) / env
    assert got == pytest.approx(a_ss, rel=6e-3), (got, a_ss)
    assert got <= 1.0 + 1e-4                                   # single scattering never creates energy
    img, _ = render(plane_furnace(mat(), mu, use_lut=True, env=env), 32)
    got_ms = float((img[..., :3] / img[..., 3:4]).mean()) / env
    assert got_ms > got or roughness < 0.25                     # the LUT only ever adds energy
    assert got_ms == pytest.approx(1.0, abs=0.012), got_ms      # ... up to a white furnace (LUT resolution 128^2, 4096 samples)


def test_clearcoat_only_furnace():
    # physicallyBased.rlsl:206-273 with a black, non-metallic base: the clearcoat lobe is the only one (probability 1), its
    # "specular colour" is vec3(ccScale) with ccScale = Schlick(0.04, N.V) x 0.2 (PhysicallyBasedMaterial.cpp:133-145 clamps
    # clearCoat to 0.2), roughness = clearCoatRoughness.
    mu, env, ccr = 0.7, 0.8, 0.5
    mat = host.bake_pbr(base_color=(0, 0, 0), roughness=1.0, metallic=0.0, specular_f0=0.0, clear_coat=1.0, clear_coat_roughness=ccr)
    scale = (0.04 + 0.96 * (1 - mu) ** 5) * 0.2
    expect = single_scatter_albedo(mu, ccr * ccr, f0=scale)
    img, _ = render(plane_furnace(mat, mu, use_lut=False, env=env), 32)
    got = float((img[..., :3] / img[..., 3:4]).mean()) / env
    assert got == pytest.approx(expect, rel=8e-3), (got, expect)


def test_clearcoat_lobe_is_sampled_about_the_base_normal():
    # physicallyBased.rlsl:230-273 builds clearCoatFrame but passes `frame` (built from the base normal N) to the clearcoat
    # lobes: microfacet normals are generated around N while N.O is tested against clearCoatN.  With a clearcoat normal map
    # that tilts clearCoatN and a mirror-like coat, the reflection is therefore the mirror direction about the BASE normal.
    # The environment is black except one bright texel block around that direction.
    tilt = 0.35
    w, h = 64, 32
    theta_v = 0.9  # camera elevation
    sc = scenes.Scene("cc_frame", width=16, height=16, use_multiscatter_lut=False)
    p, n, uv, i = scenes.plane_strip(2000, 2000)
    tan = np.tile(np.array([1, 0, 0], dtype=F), (4, 1))
    bit = np.tile(np.array([0, 0, -1], dtype=F), (4, 1))
    # tangent-space normal (sin tilt, 0, cos tilt) -> world clearCoatN = T sin + N cos = tilted towards +x
    nm = np.array([math.sin(tilt), 0.0, math.cos(tilt)]) * 0.5 + 0.5
    sc.textures.append((np.asarray(nm, dtype=F).reshape(1, 1, 3), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST))
    sc.materials[0] = host.bake_pbr(base_color=(0, 0, 0), roughness=1.0, metallic=0.0, specular_f0=0.0, clear_coat=1.0,
                                    clear_coat_roughness=0.0, clear_coat_normalmap=0)
    sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, tangents=tan, bitangents=bit, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 1, 1.0, host.FSTOP_DISABLED, 4000.0
    o.view_matrix = host.orbit_view_matrix(10.0, 0.0, theta_v)
    view = np.asarray(o.view_matrix, dtype=np.float64)
    v = view[:3, 2] / np.linalg.norm(view[:3, 2])  # direction from the surface towards the camera
    base_n = np.array([0.0, 1.0, 0.0])
    cc_n = np.array([math.sin(tilt), math.cos(tilt), 0.0])
    mirror = lambda nn: 2.0 * np.dot(nn, v) * nn - v
    totals = {}
    for name, nn in (("base", base_n), ("coat", cc_n)):
        d = mirror(nn)
        u, t = latlong_uv(d)
        env = np.zeros((h, w, 3), dtype=F)
        cx, cy = int((u % 1.0) * w), int(t * h)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                env[min(max(cy + dy, 0), h - 1), (cx + dx) % w] = 100.0
        sc.env_pixels = env
        sc.lights = host.LightRig()
        img, _ = render(sc, 4)
        totals[name] = float(img[..., :3].sum())
    assert totals["base"] > 0.0
    assert totals["coat"] == 0.0, totals


# ------------------------------------------------------------------------------------------------ HR_ESTIMATOR_ENV_MIS
# The importance-sampled environment + one-sample MIS estimator (include/hrcore.h) is NOT in the reference (its shaders leave it as
# TODOs: lightSampling.rlsl:75-77, microfacet.rlsl:94-96), so it has no text to be checked against: its contract is (1) the same
# expectation as the reference estimator — checked against a float64 quadrature of the rendering integral written here, and (2) less
# variance under small bright sources.
def sun_map(w=256, h=128):
    env = np.full((h, w, 3), 0.05, dtype=F)
    env[100:103, 40:44] = 2000.0         # a 4 x 3 texel sun at ~52 degrees elevation, 40000 x the sky
    return env


def mis_plane(estimator, env, roughness, metallic, passes, size=8):
    sc = scenes.Scene("mis_plane", width=size, height=size, use_multiscatter_lut=False)
    p, n, uv, i = scenes.plane_strip(2000, 2000)
    sc.materials[0] = host.bake_pbr(base_color=(0.8, 0.8, 0.8), roughness=roughness, metallic=metallic, specular_f0=0.0 if metallic == 0 else 0.5)
    sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
    sc.env_pixels = env
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 1, 1.0, host.FSTOP_DISABLED, 4000.0
    o.max_channel_value = 1e9
    o.view_matrix = host.orbit_view_matrix(10.0, 0.3, 0.9)
    o.max_render_passes = passes
    o.estimator = estimator
    return sc


def radiance_by_quadrature(env, view, roughness, metallic, base=0.8):
    """Outgoing radiance of the plane towards the camera: integral of BRDF x cos x (bilinearly filtered lat/long map) over the
    hemisphere, midpoint rule in (elevation, azimuth), float64.  Lambert, or the single-scatter GGX lobe with separable Smith G."""
    h, w = env.shape[:2]
    e = env[..., 0].astype(np.float64)
    V = view[:3, 2] / np.linalg.norm(view[:3, 2])
    n_el, n_az = 2048, 4096
    el = (np.arange(n_el) + 0.5) / n_el * math.pi / 2
    az = (np.arange(n_az) + 0.5) / n_az * 2 * math.pi - math.pi
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    O = np.stack([np.cos(EL) * np.sin(AZ), np.sin(EL), -np.cos(EL) * np.cos(AZ)], -1)
    x, y = (AZ / (2 * math.pi) + 0.5) * w - 0.5, (EL / math.pi + 0.5) * h - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = x - x0, y - y0
    tx = lambda ix, iy: e[np.mod(iy, h), np.mod(ix, w)]
    L = (tx(x0, y0) * (1 - fx) + tx(x0 + 1, y0) * fx) * (1 - fy) + (tx(x0, y0 + 1) * (1 - fx) + tx(x0 + 1, y0 + 1) * fx) * fy
    d_omega = (math.pi / 2 / n_el) * (2 * math.pi / n_az) * np.cos(EL)
    n_o = O[..., 1]
    if metallic == 0:
        f_cos = base / math.pi * n_o
    else:
        a = max(roughness, 0.01) ** 2
        n_i = V[1]
        H = O + V
        H /= np.linalg.norm(H, axis=-1, keepdims=True)
        n_h, i_h = H[..., 1], (H * V).sum(-1)
        D = a * a / (np.pi * (n_h ** 2 * (a * a - 1) + 1) ** 2)
        g1 = lambda c: 2 * c / (np.sqrt(a * a + (1 - a * a) * c * c) + c)
        f_cos = D * (base + (1 - base) * (1 - i_h) ** 5) * g1(n_o) * g1(n_i) / (4 * n_i)
    return float((f_cos * L * d_omega).sum())


@pytest.mark.parametrize("roughness,metallic", [(1.0, 0.0), (0.6, 1.0)])
def test_env_mis_estimator_is_unbiased_and_quieter(roughness, metallic):
    env = sun_map()
    passes = 4096
    truth = radiance_by_quadrature(env, np.asarray(host.orbit_view_matrix(10.0, 0.3, 0.9), dtype=np.float64), roughness, metallic)
    out = {}
    for est in (ffi.HR_ESTIMATOR_REFERENCE, ffi.HR_ESTIMATOR_ENV_MIS):
        img, _ = render(mis_plane(est, env, roughness, metallic, passes), passes)
        rgb = img[..., 0] / img[..., 3]
        out[est] = (float(rgb.mean()), float(rgb.std() / rgb.mean()))
    ref_mean, ref_noise = out[ffi.HR_ESTIMATOR_REFERENCE]
    mis_mean, mis_noise = out[ffi.HR_ESTIMATOR_ENV_MIS]
    assert mis_mean == pytest.approx(truth, rel=0.01), (mis_mean, truth)        # same expectation as the rendering integral
    assert ref_mean == pytest.approx(truth, rel=0.12), (ref_mean, truth)        # ... and so is the reference's, within ITS noise
    assert mis_noise < 0.25 * ref_noise, (mis_noise, ref_noise)                 # pixel-to-pixel noise after 4096 passes: >= 4 x lower
    # a short render: the reference estimator has hardly seen the sun yet, the MIS one is already within a few per cent
    img, _ = render(mis_plane(ffi.HR_ESTIMATOR_ENV_MIS, env, roughness, metallic, 64), 64)
    assert float((img[..., 0] / img[..., 3]).mean()) == pytest.approx(truth, rel=0.05)


def glass_plane(estimator, env, passes, sun_light=None, size=8):
    sc = scenes.Scene("glass_plane", width=size, height=size, use_multiscatter_lut=False)
    p, n, uv, i = scenes.plane_strip(2000, 2000)
    sc.materials[0] = host.bake_glass(base_color=(0.9, 0.9, 0.9), roughness=0.5, ior=1.5, density=0.0)
    sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
    sc.env_pixels = env
    if sun_light is not None:
        sc.lights.directional.append(sun_light)
    o = sc.options
    # depth 0: the camera ray's vertex is all there is — its reflection branch sends the next-event ray, nothing is transmitted on
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 0, 1.0, host.FSTOP_DISABLED, 4000.0
    o.max_channel_value = 1e9
    o.view_matrix = host.orbit_view_matrix(10.0, 0.3, 0.9)
    o.max_render_passes = passes
    o.estimator = estimator
    return sc


@pytest.mark.parametrize("estimator", [ffi.HR_ESTIMATOR_ENV_MIS, ffi.HR_ESTIMATOR_ALL_LIGHTS])
def test_env_mis_on_glass_a_small_sun_in_the_map_lights_like_the_equivalent_directional_light(estimator):
    # Glass under the new estimators draws its environment next-event ray from the map's importance table too (it used to
    # lobe-sample the map).  Independent statement of what that must give: a small, very bright sun in an otherwise black map lights the
    # reflection lobe like a directional light of the same irradiance from the same direction, whose value glass.rlsl:104-109 gives in
    # closed form (D G2 / (4 N.I) x baseColor) and the reference-faithful estimator evaluates without any sampling of directions.
    w, h = 512, 256
    env = np.zeros((h, w, 3), dtype=F)
    j0, i0, L = 200, 300, 40000.0
    env[j0:j0 + 2, i0:i0 + 2] = L
    irr = 0.0
    for j in range(j0, j0 + 2):
        el = ((j + 0.5) / h - 0.5) * math.pi
        irr += 2 * L * (2 * math.pi / w) * (math.pi / h) * math.cos(el)       # radiance x solid angle of the row's two texels
    el_c, az_c = ((j0 + 1.0) / h - 0.5) * math.pi, ((i0 + 1.0) / w - 0.5) * 2 * math.pi
    to_sun = np.array([math.cos(el_c) * math.sin(az_c), math.sin(el_c), -math.cos(el_c) * math.cos(az_c)], dtype=F)
    # (only the Fresnel fraction of the samples, ~7 % at this angle, takes the reflection branch at all: many passes)
    passes = 16384
    img, _ = render(glass_plane(estimator, env, passes), passes)
    sun = float((img[..., 0] / img[..., 3]).mean())
    img, _ = render(glass_plane(ffi.HR_ESTIMATOR_REFERENCE, None, passes, sun_light=(to_sun, np.full(3, irr, dtype=F))), passes)
    lamp = float((img[..., 0] / img[..., 3]).mean())
    assert lamp > 0.0
    assert sun == pytest.approx(lamp, rel=0.05), (sun, lamp)
    # ... and the map sampler is what finds it: the reference estimator, lobe-sampling the same map, is far noisier per pixel
    img, _ = render(glass_plane(ffi.HR_ESTIMATOR_REFERENCE, env, passes), passes)
    ref = img[..., 0] / img[..., 3]
    img, _ = render(glass_plane(estimator, env, passes), passes)
    new = img[..., 0] / img[..., 3]
    assert float(new.std() / new.mean()) < 0.5 * float(ref.std() / max(ref.mean(), 1e-30))


def test_env_mis_estimator_white_furnace_and_fallbacks():
    # uniform environment: expectation E x albedo for both estimators (the reference one is exact per sample here, the MIS one in the mean)
    env = np.full((8, 16, 3), 0.8, dtype=F)
    img, _ = render(mis_plane(ffi.HR_ESTIMATOR_ENV_MIS, env, 1.0, 0.0, 1024), 1024)
    assert float((img[..., :3] / img[..., 3:4]).mean()) == pytest.approx(0.8 * 0.8, rel=3e-3)
    # without an environment light the flag changes nothing (analytic lights are sampled as in the reference)
    sc0, sc1 = scenes.cornell_box(24, 24, bounces=3), scenes.cornell_box(24, 24, bounces=3)
    sc1.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS
    assert render(sc0, 2)[0].tobytes() == render(sc1, 2)[0].tobytes()


def test_env_mis_table_density_and_light_pick():
    # the importance table is a probability density over directions: its samples' densities integrate to 1 in the Monte-Carlo sense
    # (E[1 / pdf] = 4 pi), bright texels are drawn in proportion to luminosity x solid angle, and the light pick weighs the map by
    # pi x its mean luminosity next to saturate(N.L) luminosity(colour) of the analytic lights (instead of lightSampling.rlsl:74-79's 50)
    L = lib()
    L.ora_kat_env_mean_luminosity.restype = C.c_float
    env = sun_map()
    rig = host.LightRig()
    rig.add_directional(color=(1, 1, 1), illuminance=683.0 * 2.0, phi=0.0, theta=math.pi / 2)   # straight up, colour 2
    eng = oracle_lib.engine()
    eng.resize(8, 8)
    tid = eng.create_texture(env, wrap=ffi.HR_WRAP_REPEAT)
    rig.set_environment(tid, 0.0, 0.3)
    eng.set_lights(rig.bake())
    out = (C.c_float * 4)()
    n = 128
    u = (np.arange(n) + 0.5) / n
    inv_pdf, in_sun = [], 0
    h, w = env.shape[:2]
    for a in u:
        for b in u:
            L.ora_kat_env_sample(eng._ctx, C.c_float(a), C.c_float(b), out)
            d = np.array(out[:3])
            assert np.linalg.norm(d) == pytest.approx(1.0, abs=1e-5)
            inv_pdf.append(1.0 / out[3])
            uu, tt = latlong_uv(d, 0.3)
            i, j = int((uu % 1.0) * w), int(tt * h)
            in_sun += 39 <= i <= 44 and 99 <= j <= 103          # the sun block and its bilinear halo
    # the density integrates to one over the sphere (midpoint rule on the map's own grid, through the direction -> texel lookup)
    L.ora_kat_env_pdf.restype = C.c_float
    total = 0.0
    for j in range(0, h):
        el = ((j + 0.5) / h - 0.5) * math.pi
        for i in range(0, w, 1):
            az = ((i + 0.5) / w - 0.5) * 2 * math.pi - 0.3
            d = (math.cos(el) * math.sin(az), math.sin(el), -math.cos(el) * math.cos(az))
            total += float(L.ora_kat_env_pdf(eng._ctx, f3(*d))) * (2 * math.pi / w) * (math.pi / h) * math.cos(el)
    assert total == pytest.approx(1.0, rel=2e-3)
    # expected share of the (dilated) sun: its weight over the total, from the map itself
    lum = env[..., 0].astype(np.float64) * 1.03
    dil = np.maximum.reduce([np.roll(np.roll(np.pad(lum, ((1, 1), (0, 0)), mode="edge"), dx, 1), dy, 0)[1:-1] for dx in (-1, 0, 1) for dy in (-1, 0, 1)])
    cos_row = np.cos(((np.arange(h) + 0.5) / h - 0.5) * math.pi)[:, None]
    wt = (dil + lum.max() / 65536.0) * cos_row
    share = wt[99:104, 39:45].sum() / wt.sum()
    assert in_sun / (n * n) == pytest.approx(share, abs=0.02)
    mean_lum = float(L.ora_kat_env_mean_luminosity(eng._ctx))
    assert mean_lum == pytest.approx(float(wt.sum() / (w * cos_row.sum())), rel=2e-3)
    oi, of = (C.c_int * 3)(), (C.c_float * 5)()
    L.ora_kat_light_sample_mis(eng._ctx, f3(0, 1, 0), f3(0, 0, 0), C.c_float(0.999), oi, of)
    w_dir, w_env = 1.0 * (0.33 + 0.59 + 0.11) * 2.0, mean_lum * math.pi
    assert oi[0] == 4 and of[0] == pytest.approx(w_env / (w_dir + w_env), rel=1e-4)      # the environment, with its power-based probability
    L.ora_kat_light_sample_mis(eng._ctx, f3(0, 1, 0), f3(0, 0, 0), C.c_float(0.001), oi, of)
    assert oi[0] == 1 and of[0] == pytest.approx(w_dir / (w_dir + w_env), rel=1e-4)
    s = light_sample(eng, (0, 1, 0), (0, 0, 0), 0.999)                                   # the reference estimator keeps the constant 50
    assert s["type"] == 4 and s["prob"] == pytest.approx(50.0 / (50.0 + w_dir), rel=1e-4)


# ------------------------------------------------------------------------------------------------ texture LOD (mip chain + ray cones)
# HR_TEXTURE_LOD_CONE (include/hrcore.h) is NOT in the reference-faithful path (OpenRL's level selection in ray shaders is closed).
# Its oracle contract is pinned here against independent numpy restatements: the box-filtered chain, the trilinear blend, the
# footprint-to-level formula, and two scene-level consequences (a distant fine checker turns into its mean; a near one is unchanged).
def lod_lib():
    L = oracle_lib.load()
    L.ora_kat_texture_lod.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    L.ora_kat_texture_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.ora_kat_texture_level.restype = C.c_int
    L.ora_kat_footprint.argtypes = [C.c_void_p, C.c_int, f3, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, f3]
    return L


def numpy_chain(px):
    """2x2 box filter with source coordinates clamped to the level below (odd sizes), float32, pairwise sums."""
    levels = [px.astype(F)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        s = levels[-1]
        sh, sw = s.shape[:2]
        dh, dw = max(1, sh // 2), max(1, sw // 2)
        ys0, ys1 = np.minimum(2 * np.arange(dh), sh - 1), np.minimum(2 * np.arange(dh) + 1, sh - 1)
        xs0, xs1 = np.minimum(2 * np.arange(dw), sw - 1), np.minimum(2 * np.arange(dw) + 1, sw - 1)
        a, b, c, d = s[ys0][:, xs0], s[ys0][:, xs1], s[ys1][:, xs0], s[ys1][:, xs1]
        levels.append((((a + b).astype(F) + (c + d).astype(F)).astype(F) * F(0.25)).astype(F))
    return levels


def numpy_bilinear(level, u, v):
    h, w = level.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    x0, y0 = math.floor(x), math.floor(y)
    fx, fy = x - x0, y - y0
    g = lambda xx, yy: level[yy % h, xx % w].astype(np.float64)
    return (g(x0, y0) * (1 - fx) + g(x0 + 1, y0) * fx) * (1 - fy) + (g(x0, y0 + 1) * (1 - fx) + g(x0 + 1, y0 + 1) * fx) * fy


@pytest.mark.parametrize("shape,dtype", [((16, 16, 3), F), ((12, 7, 4), F), ((9, 32, 1), np.uint8), ((5, 5, 3), np.uint8)])
def test_mip_chain_and_trilinear_lookup(shape, dtype):
    rng = np.random.default_rng(3)
    px = (rng.random(shape) * 255).astype(np.uint8) if dtype == np.uint8 else rng.random(shape).astype(F)
    eng = oracle_lib.engine()
    tid = eng.create_texture(px, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR)
    L = lod_lib()
    as_float = (px.astype(F) / F(255.0)).astype(F) if dtype == np.uint8 else px
    want = numpy_chain(as_float)
    n = L.ora_kat_texture_level(eng._ctx, tid, 0, None)
    assert n == len(want) == 1 + int(math.floor(math.log2(max(shape[0], shape[1]))))
    for lvl in range(1, n):
        got = np.zeros(want[lvl].shape, dtype=F)
        L.ora_kat_texture_level(eng._ctx, tid, lvl, got.ctypes.data_as(C.POINTER(C.c_float)))
        assert got.tobytes() == want[lvl].tobytes(), lvl              # same float32 operations: bit-identical
        assert abs(float(got.mean()) - float(as_float.mean())) < 0.12  # a box filter keeps the mean (up to the odd-size clamping)
    out = (C.c_float * 4)()
    for lam in [-1.0, 0.0, 0.3, 1.0, 1.75, n - 1.0, n + 3.0]:
        for (u, v) in [(0.13, 0.71), (0.5, 0.5), (0.99, 0.02)]:
            L.ora_kat_texture_lod(eng._ctx, tid, u, v, lam, out)
            l = min(max(lam, 0.0), n - 1.0)
            l0 = int(math.floor(l))
            f = l - l0
            a = numpy_bilinear(want[l0], u, v)
            b = numpy_bilinear(want[min(l0 + 1, n - 1)], u, v)
            ref = a * (1 - f) + b * f
            c = shape[2]
            got = np.array(out[:c if c > 1 else 1])
            assert np.allclose(got, ref[:len(got)], atol=3e-6), (lam, u, v, got, ref)
    tn = eng.create_texture(px, ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_NEAREST)      # nearest-filtered textures have no chain
    assert L.ora_kat_texture_level(eng._ctx, tn, 0, None) == 1


def checker_plane(distance, lod, size=24, checks=256, vis=True):
    """The reference's ground plane (uv in [-1, 1]^2 over 8 x 8 world units) with a `checks` x `checks` one-texel checkerboard, seen
    from straight above at `distance`; the base-colour visualiser shows the texture lookup itself."""
    sc = scenes.Scene("checker", width=size, height=size, use_multiscatter_lut=False)
    p, n, uv, i = scenes.plane_strip(8, 8)
    chk = (np.add.outer(np.arange(checks), np.arange(checks)) % 2).astype(F)
    sc.textures.append((np.stack([chk] * 3, axis=-1), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR))
    sc.materials[0] = host.bake_pbr(base_color=(1, 1, 1), roughness=1.0, metallic=0.0, specular_f0=0.0, base_color_texture=0)
    sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
    o = sc.options
    o.max_ray_depth, o.aspect_ratio, o.fstop, o.focal_length = 1, 1.0, host.FSTOP_DISABLED, 50.0
    o.view_matrix = host.orbit_view_matrix(distance, 0.0, math.pi / 2 - 1e-3)
    o.texture_lod = lod
    if vis:
        o.visualizer_mode = ffi.HR_VIS_BASE_COLOR
    return sc


def test_footprint_level_formula():
    # lambda = 0.5 log2(uv area / world area) + 0.5 log2(W H) + log2(cone width / |cos|): texels across the footprint, as a level
    sc = checker_plane(10.0, ffi.HR_TEXTURE_LOD_CONE)
    eng = oracle_lib.engine()
    sc.apply(eng)
    L = lod_lib()
    out = f3()
    for (cw, cg, t, cos_i) in [(0.0, 0.002, 10.0, 1.0), (0.01, 0.001, 3.0, 0.5), (0.0, 0.25, 1.0, 0.05)]:
        d = f3(math.sqrt(1 - cos_i * cos_i), -cos_i, 0.0)
        L.ora_kat_footprint(eng._ctx, 0, d, cw, cg, t, 0.3, 0.3, out)
        w = cw + cg * t
        density = 0.5 * math.log2((2.0 * 2.0) / (8.0 * 8.0))                 # uv spans 2 x 2 over 8 x 8 world units (per triangle: halves cancel)
        want = density + math.log2(w / max(cos_i, 0.1))
        assert abs(out[1] - w) < 1e-6 * max(1.0, w)
        assert abs(out[2] - density) < 1e-5
        assert abs(out[0] - want) < 2e-5, (out[0], want)


def test_distant_checker_becomes_its_mean_and_a_near_one_is_untouched():
    far_base, _ = render(checker_plane(40.0, ffi.HR_TEXTURE_LOD_BASE), 1)
    far_cone, _ = render(checker_plane(40.0, ffi.HR_TEXTURE_LOD_CONE), 1)
    base = (far_base[..., 0] / far_base[..., 3])[9:15, 9:15]         # the plane fills the central 40 % of the frame
    cone = (far_cone[..., 0] / far_cone[..., 3])[9:15, 9:15]
    # pixel angle 2 tan(fov/2) / H = 0.02 rad -> 0.8 world units = 25.6 texels at distance 40: level 4.7, where the checker is uniform 0.5
    assert base.std() > 0.05                                      # level 0: aliased noise
    assert np.ptp(cone) < 1e-7 and abs(float(cone.mean()) - float(base.mean())) < 0.02   # one value: the checker's mean as the visualiser shows it
    near = checker_plane(0.05, ffi.HR_TEXTURE_LOD_CONE, checks=8)  # footprint 0.001 world units = 0.001 texels: level 0
    near_b = checker_plane(0.05, ffi.HR_TEXTURE_LOD_BASE, checks=8)
    assert render(near, 1)[0].tobytes() == render(near_b, 1)[0].tobytes()


def test_cone_lod_leaves_the_expectation_of_a_smooth_texture_alone_and_widens_after_a_bounce():
    # a constant texture is the same at every level: the two modes agree to the bit on a full path-traced render
    def scene(lod):
        sc = scenes.multi_material(48, 27, bounces=4, textured=True)
        sc.textures[0] = (np.full((32, 32, 3), 0.6, dtype=F), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR)
        sc.textures[1] = (np.full((32, 32, 3), 128, dtype=np.uint8), ffi.HR_WRAP_REPEAT, ffi.HR_FILTER_LINEAR)
        sc.options.texture_lod = lod
        return sc
    a, _ = render(scene(ffi.HR_TEXTURE_LOD_BASE), 3)
    b, _ = render(scene(ffi.HR_TEXTURE_LOD_CONE), 3)
    assert np.allclose(a, b, rtol=2e-6, atol=1e-7)               # (trilinear blends of equal values round differently in the last bit)
    # the cone after a diffuse bounce: spread 0.25 rad on top of the pixel's angle (Ray.coneG travels as a bf16-truncated float)
    sc = checker_plane(10.0, ffi.HR_TEXTURE_LOD_CONE)
    eng = oracle_lib.engine()
    sc.apply(eng)
    out = f3()
    lod_lib().ora_kat_footprint(eng._ctx, 0, f3(0, -1, 0), 0.2, 0.25 + 0.02, 2.0, 0.3, 0.3, out)
    assert abs(out[1] - (0.2 + 0.27 * 2.0)) < 1e-6


# ------------------------------------------------------------------------------------------------ all-lights estimator
# HR_ESTIMATOR_ALL_LIGHTS (include/hrcore.h): every PBR vertex samples the environment (MIS) AND one analytic light, instead of ONE
# light picked at random.  Same expectation as the reference estimator, without the light-pick variance.
def sun_and_sky_plane(estimator, passes, size=8, roughness=1.0, metallic=0.0, depth=1):
    sc = mis_plane(estimator, np.full((16, 32, 3), 0.4, dtype=F), roughness, metallic, passes, size=size)
    sc.lights.add_directional(color=(1.0, 0.95, 0.9), illuminance=683.0 * 1.5, phi=0.3, theta=1.1)
    sc.lights.add_point((1.0, 2.0, -0.5), luminous_intensity=683.0 * 0.8)
    sc.options.max_ray_depth = depth
    return sc


@pytest.mark.parametrize("roughness,metallic", [(1.0, 0.0), (0.5, 1.0)])
def test_all_lights_estimator_is_unbiased_and_removes_the_light_pick_noise(roughness, metallic):
    n = 4096
    ref, _ = render(sun_and_sky_plane(ffi.HR_ESTIMATOR_REFERENCE, n, roughness=roughness, metallic=metallic), n)
    mis, _ = render(sun_and_sky_plane(ffi.HR_ESTIMATOR_ENV_MIS, n, roughness=roughness, metallic=metallic), n)
    both, eng = render(sun_and_sky_plane(ffi.HR_ESTIMATOR_ALL_LIGHTS, n, roughness=roughness, metallic=metallic), n)
    m = lambda img: float((img[..., :3] / img[..., 3:4]).mean())
    assert abs(m(both) - m(ref)) < 0.02 * m(ref), (m(both), m(ref))            # the same integral ...
    assert abs(m(both) - m(mis)) < 0.02 * m(mis)
    st = eng.stats()
    assert st.rays_any > 1.5 * st.paths                                          # ... with two occlusion rays at (nearly) every vertex
    # short renders, several pixels: the pixel-to-pixel spread is the estimator's noise (the plane is uniformly lit)
    k = 16
    spread = {}
    for est in (ffi.HR_ESTIMATOR_REFERENCE, ffi.HR_ESTIMATOR_ENV_MIS, ffi.HR_ESTIMATOR_ALL_LIGHTS):
        img, _ = render(sun_and_sky_plane(est, k, size=16, roughness=roughness, metallic=metallic), k)
        v = (img[..., :3] / img[..., 3:4]).mean(axis=-1)
        spread[est] = float(v.std() / v.mean())
    assert spread[ffi.HR_ESTIMATOR_ALL_LIGHTS] < 0.5 * spread[ffi.HR_ESTIMATOR_ENV_MIS], spread
    assert spread[ffi.HR_ESTIMATOR_ALL_LIGHTS] < 0.75 * spread[ffi.HR_ESTIMATOR_REFERENCE], spread


def test_all_lights_estimator_degenerate_rigs():
    # no analytic light: the ENV_MIS estimator with the map's sampler used 7 times out of 8 and three samples at the first hit (same mean); no environment: one analytic
    # ray per vertex, every vertex
    env = sun_map()
    a, _ = render(mis_plane(ffi.HR_ESTIMATOR_ENV_MIS, env, 1.0, 0.0, 2048), 2048)
    b, eb = render(mis_plane(ffi.HR_ESTIMATOR_ALL_LIGHTS, env, 1.0, 0.0, 2048), 2048)
    ma, mb = float((a[..., :3] / a[..., 3:4]).mean()), float((b[..., :3] / b[..., 3:4]).mean())
    assert abs(ma - mb) < 0.02 * ma, (ma, mb)
    assert eb.stats().rays_any <= eb.stats().paths * 4              # three environment rays at the camera ray's hit, one at the bounce: no analytic light
    def sun_only(est):
        sc = scenes.Scene("sun_only", width=8, height=8, use_multiscatter_lut=False)
        p, n, uv, i = scenes.plane_strip(100, 100)
        sc.materials[0] = host.bake_pbr(base_color=(0.5, 0.5, 0.5), roughness=1.0, metallic=0.0, specular_f0=0.0)
        sc.meshes.append(scenes.MeshData(p, n, i, uvs=uv, mode=ffi.HR_TRIANGLE_STRIP, material_id=0))
        sc.lights.add_directional(color=(1, 1, 1), illuminance=683.0, phi=0.2, theta=0.8)
        o = sc.options
        o.max_ray_depth, o.aspect_ratio, o.fstop = 2, 1.0, host.FSTOP_DISABLED
        o.view_matrix = host.orbit_view_matrix(5.0, 0.0, 0.6)
        o.estimator = est
        return sc
    r, _ = render(sun_only(ffi.HR_ESTIMATOR_REFERENCE), 4)
    b2, _ = render(sun_only(ffi.HR_ESTIMATOR_ALL_LIGHTS), 4)
    # one light: the reference picks it with probability 1, so both estimators compute the same radiance (different summation order)
    assert np.allclose(r[..., :3] / r[..., 3:4], b2[..., :3] / b2[..., 3:4], rtol=2e-6)
