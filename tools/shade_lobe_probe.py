"""tools/shade_lobe_probe.py — what sorting hits by sampled lobe could buy k_shade_hit: the c3 soup with its 16 materials as they are (a quarter
metals among dielectrics, per triangle: a wave shades both kinds), all dielectric, all metal; shading time per shaded hit of each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from heatray_amd import _ffi as ffi, core, host, scenes

def run(label, metal):
    sc = scenes.triangle_soup(1_000_000, 1920, 1080, bounces=8, passes=64, env=True)
    if metal is not None:
        for i, m in list(sc.materials.items()):
            sc.materials[i] = host.bake_pbr(base_color=tuple(m.base_color), roughness=float(m.roughness), metallic=float(metal))
    e = core.create_engine(time_kernels=True, collect_stats=True)
    sc.apply(e)
    for rep in range(2):
        e.clear()
        for p in range(32):
            e.render_pass(sc.options.pass_params(p))
        e.synchronize()
    kt, st = e.kernel_times(), e.stats()
    hits = st.shaded_hits
    print(f"{label:14s} shade {kt['shade'][0]:7.3f} ms over {kt['shade'][1]} launches, {hits} shaded hits -> {kt['shade'][0] * 1e6 / max(hits, 1):6.3f} ns per hit;  trace {kt['trace'][0]:.2f} ms", flush=True)

run("c3 as it is", None)
run("all dielectric", 0.0)
run("all metal", 1.0)
