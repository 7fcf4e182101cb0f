#!/usr/bin/env python3
"""PCIe-inclusive rates on workload c3 (not the bench `value`): one host snapshot of the RGBA32F buffer per pass, complete
(hr_readback: drains the pipeline) vs progressive (hr_readback_progressive), and RGBA8 display snapshots (hr_display_readback)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from heatray_amd import _ffi as ffi, core  # noqa: E402

sc = bench.build_scene("c3", 0, 0, 160)
eng = core.create_engine()
sc.apply(eng)
P = ffi.display_params(tonemapping_enabled=True)


def run(name, snap, n=96):
    eng.clear()
    for i in range(8):
        eng.render_pass(sc.options.pass_params(i))
        snap()
    eng.readback()
    st0 = eng.stats()
    r0 = st0.rays_closest + st0.rays_any
    t0 = time.perf_counter()
    for i in range(8, 8 + n):
        eng.render_pass(sc.options.pass_params(i))
        snap()
    eng.readback()
    el = time.perf_counter() - t0
    st = eng.stats()
    rays = st.rays_closest + st.rays_any - r0
    print(f"{name:48s} {el / n * 1e3:7.3f} ms per pass  {rays / el / 1e6:8.1f} Mrays/s")


run("no snapshot (pipelined)", lambda: None)
run("hr_readback_progressive every pass (33 MB)", lambda: eng.readback_progressive(copy=False))
run("hr_readback every pass (33 MB, drains)", lambda: eng.readback(copy=False))
run("hr_display_readback RGBA8 every pass (8 MB, drains)", lambda: eng.display(P, ffi.HR_DISPLAY_RGBA8))
run("hr_display_readback RGBA8 progressive (8 MB)", lambda: eng.display(P, ffi.HR_DISPLAY_RGBA8 | ffi.HR_DISPLAY_PROGRESSIVE))
