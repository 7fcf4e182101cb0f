#!/usr/bin/env python3
"""Render a scene with the HIP core and write what the viewer would show (hr_display, RGBA8) as a PNG — a sanity check for
human eyes.   python tools/render_image.py <cornell|multi|soup|terrain> out.png [passes] [width height]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heatray_amd import _ffi as ffi  # noqa: E402
from heatray_amd import core, scenes  # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    w = int(sys.argv[4]) if len(sys.argv) > 5 else 480
    h = int(sys.argv[5]) if len(sys.argv) > 5 else 360
    if name == "cornell":
        sc = scenes.cornell_box(w, h, bounces=5, passes=passes)
    elif name == "multi":
        sc = scenes.multi_material(w, h, bounces=6, passes=passes, textured=True)
    elif name == "terrain":
        sc = scenes.terrain(200, 100, w, h, bounces=4, passes=passes, env=True)
    else:
        sc = scenes.triangle_soup(20000, w, h, bounces=4, passes=passes, env=True)
    eng = core.create_engine()
    sc.apply(eng)
    for s in range(passes):
        eng.render_pass(sc.options.pass_params(s))
    img = eng.display(ffi.display_params(tonemapping_enabled=True), ffi.HR_DISPLAY_RGBA8)[::-1]   # row 0 = bottom -> top first
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img[..., :3])).save(out)
    print(out, img.shape, "mean", img[..., :3].mean(axis=(0, 1)))


if __name__ == "__main__":
    main()
