import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from heatray_amd import core, scenes
sc = bench.build_scene("c3", 0, 0, 32)
eng = core.create_engine()
t0 = time.perf_counter(); sc.apply(eng); t1 = time.perf_counter()
print("first apply (incl. host->ctx copies): %.1f ms; build_ms %.2f" % ((t1 - t0) * 1e3, eng.scene_info().build_ms))
for k in range(4):
    eng.set_transform(0, scenes._translate(0.01 * (k + 1), 0.0, 0.0))
    t0 = time.perf_counter(); eng.commit(); t1 = time.perf_counter()
    print("recommit after a transform: wall %.2f ms, build_ms %.2f" % ((t1 - t0) * 1e3, eng.scene_info().build_ms))
eng.render_pass(sc.options.pass_params(0)); eng.flush(); eng.synchronize()
print("ok", eng.stats().paths)
