#!/usr/bin/env python3
"""GPU box: is a two-level BVH needed for per-object motion?

An instanced scene — 16 rigid, non-overlapping objects of 65 k triangles each on a grid — in which ONE object travels through the
others along the grid's diagonal in 100 edit steps (hr_geom_set_transform + hr_scene_commit, the reference's Scene::applyTransform,
Scene.cpp:38-49).  At checkpoints the traversal rate of the tree that has been REFITTED ever since the first build is compared with a
FRESH build of the same pose; then the object is flung far outside the scene to find where the tree-quality guard (area / diagonal^2
against its value at build time, default 4 x) should cut in.  Decision rule of the verdict: build a TLAS only if the refitted tree loses
more than 10 %."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from heatray_amd import core, scenes  # noqa: E402

PASSES = 48


def rate(eng, sc):
    for i in range(9):
        eng.render_pass(sc.options.pass_params(i))
    eng.flush()
    eng.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(PASSES):
        eng.render_pass(sc.options.pass_params(i % sc.options.max_render_passes))
    eng.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.stats()
    rays = st.rays_closest + st.rays_any
    return rays / dt / 1e6, st.node_visits / max(rays, 1) if st.node_visits else 0.0


def pose(sc, t):
    """object 0 at parameter t in [0, 1] of its way along the grid's diagonal (through objects 5, 10 and 15)"""
    a = sc.meshes[0].world[:3, 3].copy()
    b = sc.meshes[15].world[:3, 3].copy()
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = a + (b - a) * np.float32(t)
    return m


def main():
    sc = scenes.instanced()
    print(f"scene: {sc.n_triangles} triangles, {len(sc.meshes)} submeshes")
    os.environ["HR_TUNE"] = "guard=100000"      # never fall back to a rebuild: this is the measurement of what the fallback is worth
    eng = core.create_engine(collect_stats=True)
    sc.apply(eng)
    r0, v0 = rate(eng, sc)
    print(f"step   0 (built)      : {r0:8.1f} Mrays/s  {v0:5.1f} node visits per ray")
    steps = 100
    commit_ms = []
    for k in range(1, steps + 1):
        eng.set_transform(0, pose(sc, k / steps))
        t0 = time.perf_counter()
        eng.commit()
        commit_ms.append((time.perf_counter() - t0) * 1e3)
        assert eng.scene_info().refitted == 1
        if k in (10, 25, 33, 50, 66, 75, 100):
            rr, vr = rate(eng, sc)
            q = eng.scene_info().box_area_ratio
            # a fresh context with object 0 at this pose: full build
            sc_pose = scenes.instanced()
            sc_pose.meshes[0].world = pose(sc, k / steps)
            f2 = core.create_engine(collect_stats=True)
            sc_pose.apply(f2)
            rf, vf = rate(f2, sc_pose)
            f2.close()
            print(f"step {k:3d} (refitted x{k:3d}): {rr:8.1f} Mrays/s  {vr:5.1f} visits | fresh build {rf:8.1f} Mrays/s  {vf:5.1f} visits | refit / fresh = {rr / rf:.3f} | box area ratio {q:.3f}")
    print(f"commit (refit) wall time: median {np.median(commit_ms):.3f} ms, max {np.max(commit_ms):.3f} ms over {steps} edits")
    # ---- the guard: fling object 0 away by d scene diagonals and compare the refitted tree with a fresh build
    info = eng.scene_info()
    diag = float(np.linalg.norm(np.array(info.aabb_max) - np.array(info.aabb_min)))
    base = sc.meshes[15].world[:3, 3].copy()
    print("fling: object 0 moved d x the scene diagonal along +x (the scene's bounds and ray epsilon grow with it)")
    for d in (0.25, 0.5, 1.0, 2.0, 4.0):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = base + np.array([d * diag, 0, 0], np.float32)
        eng.set_transform(0, m)
        eng.commit()
        assert eng.scene_info().refitted == 1
        rr, vr = rate(eng, sc)
        q = eng.scene_info().box_area_ratio
        sc_pose = scenes.instanced()
        sc_pose.meshes[0].world = m
        sc_pose.options = sc.options
        f2 = core.create_engine(collect_stats=True)
        sc_pose.apply(f2)
        rf, vf = rate(f2, sc_pose)
        f2.close()
        print(f"  d = {d:4.2f}: refitted {rr:8.1f} Mrays/s {vr:5.1f} visits | fresh {rf:8.1f} Mrays/s {vf:5.1f} visits | refit / fresh = {rr / rf:.3f} | box area ratio {q:.3f}")
    eng.close()
    # ---- the default guard (box area > 1.25 x the built tree's) on the same walk and the same flings
    os.environ["HR_TUNE"] = ""
    g = core.create_engine(collect_stats=True)
    sc2 = scenes.instanced()
    sc2.apply(g)
    rebuilds, ms = [], []
    for k in range(1, steps + 1):
        g.set_transform(0, pose(sc, k / steps))
        g.commit()
        ms.append(g.scene_info().build_ms)
        if g.scene_info().refitted != 1:
            rebuilds.append(k)
        if k in (50, 100):
            rr, vr = rate(g, sc2)
            print(f"default guard, step {k:3d}: {rr:8.1f} Mrays/s {vr:5.1f} visits, box area ratio {g.scene_info().box_area_ratio:.3f}")
    print(f"default guard (1.25 x): {len(rebuilds)} of {steps} commits rebuilt (at steps {rebuilds}); commit device time mean {np.mean(ms):.3f} ms, max {np.max(ms):.3f} ms")
    for d in (0.25, 0.5, 1.0, 2.0, 4.0):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = base + np.array([d * diag, 0, 0], np.float32)
        g.set_transform(0, m)
        g.commit()
        print(f"  default guard, fling d = {d:4.2f}: {'refit' if g.scene_info().refitted == 1 else 'REBUILD'}  commit {g.scene_info().build_ms:.2f} ms")
    g.close()


if __name__ == "__main__":
    main()
