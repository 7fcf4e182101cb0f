import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib
from heatray_amd import _ffi as ffi, core, host, scenes
import test_gpu_fuzz as fz
golden = np.load("tests/golden/ref_vectors.npz")

def run(sc, passes=3):
    g, o = core.create_engine(), oracle_lib.engine()
    for eng in (g, o):
        sc.apply(eng, lut=golden["multiscatter_lut"])
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
    a, b = g.readback(), o.readback()
    return int((a != b).any(axis=-1).sum()), a, b

for seed in [int(x) for x in sys.argv[1:]]:
    sc = fz.random_scene(1000 + seed)
    n, a, b = run(sc)
    print(f"seed {seed}: {n} differ; depth {sc.options.max_ray_depth} fstop {sc.options.fstop} mode {sc.options.sample_mode} lights d{sc.lights.n_directional if hasattr(sc.lights,'n_directional') else '?'}")
    d = np.argwhere((a != b).any(axis=-1))[:5]
    for (y, x) in d:
        print("   px", x, y, a[y, x], b[y, x])
    # ablations
    def variant(name, f):
        s2 = fz.random_scene(1000 + seed); f(s2); n2, _, _ = run(s2); print(f"   {name}: {n2}")
    variant("depth0", lambda s: setattr(s.options, "max_ray_depth", 0))
    variant("depth1", lambda s: setattr(s.options, "max_ray_depth", 1))
    variant("no dof", lambda s: setattr(s.options, "fstop", host.FSTOP_DISABLED))
    def strip_tex(s):
        for m in s.materials.values():
            m.flags &= ~(ffi.HR_MF_HAS_BASE_COLOR_TEXTURE | ffi.HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE | ffi.HR_MF_HAS_EMISSIVE_TEXTURE | ffi.HR_MF_HAS_NORMALMAP | ffi.HR_MF_HAS_CLEARCOAT_TEXTURE | ffi.HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE | ffi.HR_MF_HAS_CLEARCOAT_NORMALMAP)
    variant("no textures", strip_tex)
    def strip_nm(s):
        for m in s.materials.values():
            m.flags &= ~(ffi.HR_MF_HAS_NORMALMAP | ffi.HR_MF_HAS_CLEARCOAT_NORMALMAP)
    variant("no normalmaps", strip_nm)
    def strip_alpha(s):
        for m in s.materials.values():
            m.flags &= ~ffi.HR_MF_ALPHA_MASK; m.flags |= ffi.HR_MF_DOUBLE_SIDED
    variant("no alpha/single-sided", strip_alpha)
    def no_glass(s):
        for k, m in list(s.materials.items()):
            if m.type == ffi.HR_MAT_GLASS: s.materials[k] = host.bake_pbr()
    variant("no glass", no_glass)
    def no_env(s): s.env_pixels = None
    variant("no env", no_env)
    def no_vc(s):
        for m in s.materials.values(): m.flags &= ~ffi.HR_MF_VERTEX_COLORS
    variant("no vertex colors", no_vc)
    def all_occ(s):
        for me in s.meshes: me.is_occluder = True
    variant("all occluders", all_occ)
