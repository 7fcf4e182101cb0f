import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib
from heatray_amd import core, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
t0 = time.perf_counter(); sc = scenes.triangle_soup(n, width=64, height=64, bounces=3, passes=4, env=True); print("scene", time.perf_counter() - t0)
lut = np.load(os.path.join(os.getcwd(), 'tests', 'golden', 'ref_vectors.npz'))['multiscatter_lut']
g = core.create_engine(); t0 = time.perf_counter(); sc.apply(g, lut=lut); print("gpu apply", time.perf_counter() - t0, "build_ms", g.scene_info().build_ms, "nodes", g.scene_info().n_nodes)
o = oracle_lib.engine(); t0 = time.perf_counter(); sc.apply(o, lut=lut); print("oracle apply", time.perf_counter() - t0)
rng = np.random.default_rng(1)
m = 20000
org = rng.uniform(-1.1, 1.1, (m, 3)).astype(np.float32)
d = rng.normal(size=(m, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
hg, ho = g.debug_trace(org, d), o.debug_trace(org, d)
print("hits", (ho["prim"] >= 0).mean(), "equal", hg.tobytes() == ho.tobytes())
for s in range(2):
    g.render_pass(sc.options.pass_params(s)); o.render_pass(sc.options.pass_params(s))
print("render equal", g.readback().tobytes() == o.readback().tobytes())
