#!/usr/bin/env python3
"""Experiment (GPU box): Mrays/s on the coherent 1 M-triangle terrain mesh (scenes.terrain) — a non-uniform counterpart of the
benchmark's triangle fog, to compare tree builders (HRCORE_LIB=build_variants/libhrcore_<variant>.so)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from heatray_amd import core, scenes  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = scenes.terrain(1000, 500, 1920, 1080, bounces=8, passes=max(32, passes + 8), env=True)
eng = core.create_engine(collect_stats=True)
sc.apply(eng)
for i in range(8):
    eng.render_pass(sc.options.pass_params(i))
eng.flush()
eng.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(passes):
    eng.render_pass(sc.options.pass_params(8 + i))
eng.flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = eng.stats()
rays = st.rays_closest + st.rays_any
print("terrain %d triangles, %d nodes, %d levels: %.1f Mrays/s, %.3f ms/pass, node visits per ray %.1f, triangle tests per ray %.2f" % (
    sc.n_triangles, eng.scene_info().n_nodes, eng.scene_info().bvh_levels, rays / dt / 1e6, dt / passes * 1e3,
    st.node_visits / max(rays, 1), st.tri_tests / max(rays, 1)))
