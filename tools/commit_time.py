"""Commit timings on the GPU box (DESIGN.md §Dynamic updates): first commit from host arrays, re-commit after transform edits
(refit), re-commit with refit disabled (full rebuild of resident geometry).   python tools/commit_time.py [n_tris]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from heatray_amd import core, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = scenes.triangle_soup(n, 1920, 1080, bounces=8, passes=32, env=True)
out = {"triangles": n}
for label, tune in (("refit", ""), ("rebuild", "refit=0")):
    os.environ["HR_TUNE"] = tune
    eng = core.create_engine()
    eng.resize(sc.width, sc.height)
    for mid, mat in sc.materials.items():
        eng.set_material(mid, mat)
    eng.synchronize()
    t0 = time.perf_counter()
    for me in sc.meshes:
        eng.add_mesh(me.positions, me.normals, me.indices, material_id=me.material_id)
    t1 = time.perf_counter()
    eng.commit()
    t2 = time.perf_counter()
    out[f"{label}_first_ingest_ms"] = (t1 - t0) * 1e3
    out[f"{label}_first_commit_ms"] = (t2 - t1) * 1e3
    out[f"{label}_first_build_ms_device"] = eng.scene_info().build_ms
    walls, devs = [], []
    for k in range(6):
        m = scenes._translate(0.01 * (k + 1), 0.0, 0.0)
        for gid in range(len(sc.meshes)):
            eng.set_transform(gid, m)
        t0 = time.perf_counter()
        eng.commit()
        walls.append((time.perf_counter() - t0) * 1e3)
        info = eng.scene_info()
        devs.append(info.build_ms)
        assert bool(info.refitted) == (label == "refit")
    out[f"{label}_recommit_wall_ms"] = walls
    out[f"{label}_recommit_device_ms"] = devs
    eng.close()
# tree cache (hr_scene_cache): first commit writes the file, a second context reads it
import tempfile
path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "scene.hrbvh")
os.environ["HR_TUNE"] = ""
for label in ("cache_miss_and_write", "cache_hit"):
    eng = core.create_engine()
    eng.resize(sc.width, sc.height)
    eng.set_scene_cache(path)
    for me in sc.meshes:
        eng.add_mesh(me.positions, me.normals, me.indices, material_id=me.material_id)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.commit()
    out[f"{label}_commit_wall_ms"] = (time.perf_counter() - t0) * 1e3
    out[f"{label}_refitted_flag"] = int(eng.scene_info().refitted)
    eng.close()
out["cache_file_mb"] = os.path.getsize(path) / 1e6
import json
print(json.dumps(out))
