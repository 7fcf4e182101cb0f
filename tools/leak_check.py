"""Creates and destroys contexts with every kind of state (meshes, textures, env table, refit, cache, slots) and watches the free
device memory: a leak shows as a steady decline.   python tools/leak_check.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from heatray_amd import _ffi as ffi
from heatray_amd import core, scenes

sc = scenes.triangle_soup(200_000, 640, 360, bounces=6, passes=16, env=True, passthrough_fraction=0.3)
sc.env_pixels = scenes.synthetic_hdri(512, 256)
path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "t.hrbvh")
free = []
for it in range(12):
    eng = core.create_engine()
    eng.set_scene_cache(path if it % 2 else None)
    sc.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS if it % 3 == 0 else ffi.HR_ESTIMATOR_REFERENCE
    sc.apply(eng)
    for s in range(10):
        eng.render_pass(sc.options.pass_params(s))
    eng.set_transform(0, scenes._translate(0.01, 0, 0))
    eng.commit()
    gid = eng.add_mesh(sc.meshes[0].positions[:300], sc.meshes[0].normals[:300], list(range(300)), material_id=1)
    eng.commit()
    eng.remove_mesh(gid)
    eng.commit()
    eng.render_pass(sc.options.pass_params(11))
    eng.readback()
    eng.resize(320, 200)
    eng.close()
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0] >> 20)
print("free MiB after each round:", free)
assert max(free[2:]) - min(free[2:]) < 64, "device memory keeps shrinking: leak"
print("ok")
