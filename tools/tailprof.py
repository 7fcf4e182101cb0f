#!/usr/bin/env python3
"""Experiment (GPU box): per k_trace launch, when the work queue runs dry and when the launch ends, and the longest ray.
Needs build_variants/libhrcore_tail.so (tools/build_variant.sh tail "-DHR_TAILPROF")."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HRCORE_LIB", os.path.join(ROOT, "build_variants", "libhrcore_tail.so"))
os.environ.setdefault("HR_TUNE", "groups=1,batch=1")  # one trace launch per pass
import bench  # noqa: E402
from heatray_amd import core  # noqa: E402

sc = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "c3", 0, 0, 32)
eng = core.create_engine()
sc.apply(eng)
lib = core.load_library()
buf = (C.c_ulonglong * 24)()
for i in range(12):                      # fill the pipeline
    eng.render_pass(sc.options.pass_params(i))
eng.synchronize()
lib.hr_debug_tailprof(buf, 1)
rows = []
for i in range(12, 20):                  # steady state: one trace launch per pass (the call below syncs the device after each)
    eng.render_pass(sc.options.pass_params(i))
    lib.hr_debug_tailprof(buf, 1)
    v = list(buf)
    if v[5]:
        rows.append(v)
        # wall_clock64 ticks at 100 MHz
        print("launch: %8d rays  total %.3f ms  queue dry after %.3f ms  tail %.3f ms   steps/ray mean %.1f max %d" % (
            v[5], (v[2] - v[0]) / 1e5, (v[1] - v[0]) / 1e5, (v[2] - v[1]) / 1e5, v[4] / v[5], v[3]))
        print("        subtrees handed over %d; waves by drain time (0.05 ms buckets, first 8): %s" % (v[6], " ".join(str(x) for x in v[8:16])))
        if v[19]:
            print("        shader clock during the launch: %.0f MHz" % (v[18] / (v[19] / 100.0)))
        if v[20] and v[22]:
            print("        node-step rounds %d with %.1f lanes of 64; triangle phases %d with %.1f lanes (%.2f phases and %.1f rounds per ray)" % (
                v[22], v[23] / v[22], v[20], v[21] / v[20], v[20] / v[5] * 64, v[22] / v[5] * 64))
        print("        drain rounds of a wave: max %d, mean %.1f; lanes busy in a drain round: %.1f of 64" % (v[7], v[17] / 5120.0, v[16] / max(v[17], 1)))
