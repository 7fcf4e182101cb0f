#!/usr/bin/env python3
"""Experiment (GPU box): where k_trace's wave-instruction slots go.  Needs build_variants/libhrcore_lp.so
(tools/build_variant.sh lp "-DHR_LANEPROF").   HRCORE_LIB=build_variants/libhrcore_lp.so python tools/laneprof.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HRCORE_LIB", os.path.join(ROOT, "build_variants", "libhrcore_lp.so"))
import bench  # noqa: E402
from heatray_amd import core  # noqa: E402

sc = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "c3", 0, 0, 32)
eng = core.create_engine(collect_stats=True)
sc.apply(eng)
lib = core.load_library()
buf = (C.c_ulonglong * 16)()
for i in range(2):
    eng.render_pass(sc.options.pass_params(i))
eng.flush()
lib.hr_debug_laneprof(buf, 1)
for i in range(2, 10):
    eng.render_pass(sc.options.pass_params(i))
st = eng.stats()
lib.hr_debug_laneprof(buf, 0)
v = list(buf)
rays = 8 * (st.rays_closest + st.rays_any) / 10.0
print("node-step slots %d  avg active lanes %.1f   (deep-stack path taken in %.1f%% of slots)" % (v[0], v[1] / max(v[0], 1), 100.0 * v[8] / max(v[0], 1)))
print("lanes holding a ray during node-step slots: %.1f of 64   (slots incl. empty %d)" % (v[10] / max(v[11], 1), v[11]))
print("tri phases %d  avg lanes %.1f" % (v[2], v[3] / max(v[2], 1)))
print("refills %d  avg idle lanes at refill %.1f" % (v[5], v[6] / max(v[5], 1)))
print("retire executions %d avg lanes %.1f" % (v[9], v[12] / max(v[9], 1)))
print("outer rounds %d;  node-step slots per round %.2f" % (v[7], v[0] / max(v[7], 1)))
print("per ray: node-step lane-slots %.1f, wave-slots*64 %.1f" % (v[1] / rays, v[0] * 64 / rays))
